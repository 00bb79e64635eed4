#!/usr/bin/env python
"""Where a HunyuanVideo block's non-attention time goes at C3 size (bench.py's HunyuanBlock): HIP-event time of every op of a
double-stream and of a single-stream block on 119 056 rows.  python tools/block_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda:0")
N, HID, FFN, H = 119056, 3072, 12288, 24


def t(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / reps


with torch.no_grad():
    x = torch.randn(N, HID, device=dev, dtype=torch.bfloat16)
    o = torch.randn(1, H, N, 128, device=dev, dtype=torch.bfloat16)
    for kind in ("double", "single"):
        blk = bench.HunyuanBlock(kind, dev, HID, FFN, H)
        h = blk.pre(x)
        print(f"{kind}: pre {t(lambda: blk.pre(x)):.2f} ms, post {t(lambda: blk.post(x, h, o)):.2f} ms")
        m = blk.mod
        print(f"   layer_norm {t(lambda: torch.nn.functional.layer_norm(x, (HID,))):.2f}  addcmul(shift, xn, 1+scale) {t(lambda: torch.addcmul(m[0], x, 1 + m[1])):.2f}")
        if kind == "double":
            print(f"   qkv gemm {t(lambda: torch.addmm(blk.qkv.bias, x, blk.qkv.weight.t())):.2f}  proj gemm {t(lambda: torch.addmm(blk.proj.bias, x, blk.proj.weight.t())):.2f}")
            hh = torch.addmm(blk.qkv.bias, x, blk.qkv.weight.t())
            print(f"   fc1+gelu {t(lambda: torch._addmm_activation(blk.fc1.bias, x, blk.fc1.weight.t(), use_gelu=True)):.2f}")
            g = torch._addmm_activation(blk.fc1.bias, x, blk.fc1.weight.t(), use_gelu=True)
            print(f"   fc2 {t(lambda: torch.addmm(blk.fc2.bias, g, blk.fc2.weight.t())):.2f}")
        else:
            print(f"   linear1 gemm {t(lambda: torch.addmm(blk.lin1.bias, x, blk.lin1.weight.t())):.2f}")
            hh = torch.addmm(blk.lin1.bias, x, blk.lin1.weight.t())
            blk.post(x, hh, o)
            print(f"   gelu.out slice->cat {t(lambda: torch.ops.aten.gelu.out(hh[:, 3 * HID:], approximate='tanh', out=blk.cat[:, HID:])):.2f}  "
                  f"linear2 gemm {t(lambda: torch.addmm(blk.lin2.bias, blk.cat, blk.lin2.weight.t())):.2f}")
        qk = hh[:, :2 * HID].view(-1, 2 * H, HID // H)
        print(f"   q/k rms_norm {t(lambda: torch.nn.functional.rms_norm(qk, (HID // H,))):.2f}  tokens_first {t(lambda: blk._tokens_first(o)):.2f}  "
              f"gated residual {t(lambda: torch.addcmul(x, m[2], x)):.2f}")

    # q/k RMSNorm variants (the strided-view call above is what bench.py first used: 4.8 ms for 1.46 GB of q, k)
    hh = torch.randn(N, 3 * HID + FFN, device=dev, dtype=torch.bfloat16)
    v3 = hh.view(N, (3 * HID + FFN) // 128, 128)[:, :2 * H]
    print(f"rms_norm on [N, 48, 128] strided view: {t(lambda: torch.nn.functional.rms_norm(v3, (128,))):.2f} ms")
    c3 = v3.contiguous()
    print(f"   contiguous copy {t(lambda: v3.contiguous()):.2f} + rms_norm contiguous {t(lambda: torch.nn.functional.rms_norm(c3, (128,))):.2f}")
    w = torch.ones(128, device=dev, dtype=torch.bfloat16)
    print(f"   rms_norm contiguous with weight {t(lambda: torch.nn.functional.rms_norm(c3, (128,), w, 1e-6)):.2f}")
    def manual(x):
        xf = x.float()
        return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6)).to(x.dtype)
    print(f"   manual fp32 chain {t(lambda: manual(v3)):.2f}")
    try:
        cm = torch.compile(manual)
        cm(v3)
        print(f"   torch.compile'd chain {t(lambda: cm(v3)):.2f}")
    except Exception as e:
        print("   torch.compile unavailable:", str(e)[:100])
    print(f"layer_norm contiguous [N, 3072]: {t(lambda: torch.nn.functional.layer_norm(x, (HID,))):.2f}")
