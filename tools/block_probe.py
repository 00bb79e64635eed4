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
        g = blk.pre(x)
        print(f"{kind}: pre {t(lambda: blk.pre(x)):.2f} ms, post {t(lambda: blk.post(x, g, o)):.2f} ms")
    hq = torch.randn(N, 3 * HID, device=dev, dtype=torch.bfloat16)
    print(f"qkv_split_norm (split + q/k RMSNorm + rotary + head-major, 2.2 GB): {t(lambda: blk._qk_norm(hq)):.2f} ms")
    w2 = blk.lin2.weight
    a2 = torch.randn(N, HID, device=dev, dtype=torch.bfloat16)
    g2 = torch.randn(N, FFN, device=dev, dtype=torch.bfloat16)
    print(f"linear2 halves: K=3072 {t(lambda: torch.addmm(blk.lin2.bias, a2, w2[:, :HID].t())):.2f}  K=12288 {t(lambda: torch.addmm(a2, g2, w2[:, HID:].t())):.2f}")
    w1 = blk.lin1.weight
    print(f"linear1 halves: qkv {t(lambda: torch.addmm(blk.lin1.bias[:3 * HID], x, w1[:3 * HID].t())):.2f}  mlp+gelu {t(lambda: torch._addmm_activation(blk.lin1.bias[3 * HID:], x, w1[3 * HID:].t(), use_gelu=True)):.2f}")
