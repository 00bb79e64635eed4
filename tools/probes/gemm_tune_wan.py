"""tools/probes/gemm_tune.py at the Wan2.1 1.3B block's GEMM shapes (32 768 rows, dim 1536, ffn 8960): is there a faster hipBLASLt / rocBLAS
solution than the one torch picks?  (14.8 % of the Wan line's kernel time is these GEMMs.)   usage (GPU box): python tools/probes/gemm_tune_wan.py"""
import os, sys, time
import torch
dev = torch.device("cuda:0")
R, HID, FFN, TXT = 32768, 1536, 8960, 512
g = torch.Generator(device=dev).manual_seed(0)
def rnd(*s):
    return (torch.randn(*s, device=dev, dtype=torch.bfloat16, generator=g) * 0.05)
x, h, ctx = rnd(R, HID), rnd(R, FFN), rnd(TXT, HID)
w_qkv, b_qkv = rnd(3 * HID, HID), rnd(3 * HID)
w_o, b_o = rnd(HID, HID), rnd(HID)
w_kv, b_kv = rnd(2 * HID, HID), rnd(2 * HID)
w_fc2, b_fc2 = rnd(HID, FFN), rnd(HID)
w_fc1, b_fc1 = rnd(FFN, HID), rnd(FFN)
cases = {
    "qkv  addmm [R,1536]x[1536,4608]": (lambda: torch.addmm(b_qkv, x, w_qkv.t()), 2.0 * R * HID * 3 * HID),
    "o/cq/co addmm [R,1536]x[1536,1536]": (lambda: torch.addmm(b_o, x, w_o.t()), 2.0 * R * HID * HID),
    "ckv  addmm [512,1536]x[1536,3072]": (lambda: torch.addmm(b_kv, ctx, w_kv.t()), 2.0 * TXT * HID * 2 * HID),
    "fc1  addmm+gelu [R,1536]x[1536,8960] (dense comparator)": (lambda: torch._addmm_activation(b_fc1, x, w_fc1.t(), use_gelu=True), 2.0 * R * HID * FFN),
    "fc2  addmm [R,8960]x[8960,1536] (full steps)": (lambda: torch.addmm(b_fc2, h, w_fc2.t()), 2.0 * R * HID * FFN),
}
def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
base = {k: bench(f) for k, (f, _) in cases.items()}
for k, t in base.items():
    print(f"default  {k:58s} {t * 1e3:8.1f} us  {cases[k][1] / t / 1e9:7.1f} TFLOP/s", flush=True)
import torch.cuda.tunable as tun
out = os.environ.get("GT_OUT", "gpurun_out/tunableop_wan.csv")
os.makedirs(os.path.dirname(out), exist_ok=True)
tun.enable(True)
tun.tuning_enable(True)
tun.set_filename(out)
tun.set_max_tuning_duration(int(os.environ.get("GT_MS", "40")))
tun.set_max_tuning_iterations(int(os.environ.get("GT_IT", "8")))
for k, (f, _) in cases.items():
    t0 = time.time()
    f()
    torch.cuda.synchronize()
    print(f"tuned {k} in {time.time() - t0:.1f}s", flush=True)
tun.tuning_enable(False)
if hasattr(tun, "write_file"):
    tun.write_file(out)
for k, (f, fl) in cases.items():
    t = bench(f)
    print(f"tuned    {k:58s} {t * 1e3:8.1f} us  {fl / t / 1e9:7.1f} TFLOP/s   x{base[k] / t:.3f}", flush=True)
if os.path.exists(out):
    print(open(out).read())
