"""Which hipBLASLt / rocBLAS solution does torch pick for the block's GEMM shapes, and is there a faster one?

torch's TunableOp (torch.cuda.tunable) times every candidate solution for each (shape, layout, epilogue) it meets and writes
the winners to a CSV; with tuning off and the CSV loaded it just dispatches to the recorded solution.  This probe times the
HunyuanVideo block's GEMMs (119 056 rows) before and after tuning and writes the CSV under gpurun_out/."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
dev = torch.device("cuda:0")
ROWS = int(os.environ.get("GT_ROWS", "119056"))
HID, MLP = 3072, 12288
g = torch.Generator(device=dev).manual_seed(0)
def rnd(*s):
    return (torch.randn(*s, device=dev, dtype=torch.bfloat16, generator=g) * 0.05)
x = rnd(ROWS, HID)
h = rnd(ROWS, MLP)
w_qkv, b_qkv = rnd(3 * HID, HID), rnd(3 * HID)
w_proj, b_proj = rnd(HID, HID), rnd(HID)
w_fc1, b_fc1 = rnd(MLP, HID), rnd(MLP)
w_fc2, b_fc2 = rnd(HID, MLP), rnd(HID)
w_l2 = rnd(HID, HID + MLP)
y = rnd(ROWS, HID)
cases = {
    "qkv   addmm        [R,3072]x[3072,9216]": lambda: torch.addmm(b_qkv, x, w_qkv.t()),
    "proj  addmm        [R,3072]x[3072,3072]": lambda: torch.addmm(b_proj, x, w_proj.t()),
    "fc1   addmm+gelu   [R,3072]x[3072,12288]": lambda: torch._addmm_activation(b_fc1, x, w_fc1.t(), use_gelu=True),
    "fc2   addmm        [R,12288]x[12288,3072]": lambda: torch.addmm(b_fc2, h, w_fc2.t()),
    "lin2a addmm view   [R,3072]x[3072,3072] ld 15360": lambda: torch.addmm(b_proj, x, w_l2[:, :HID].t()),
    "lin2b addmm(y) view[R,12288]x[12288,3072] ld 15360": lambda: torch.addmm(y, h, w_l2[:, HID:].t()),
}
flops = {k: 2.0 * ROWS * (3072 * 9216 if "qkv" in k else 3072 * 3072 if ("proj" in k or "lin2a" in k) else 3072 * 12288) for k in cases}
def bench(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
base = {k: bench(f) for k, f in cases.items()}
for k, t in base.items():
    print(f"default  {k:52s} {t:7.3f} ms  {flops[k] / t / 1e9:7.1f} TFLOP/s", flush=True)
import torch.cuda.tunable as tun
out = os.environ.get("GT_OUT", "gpurun_out/tunableop_results.csv")
os.makedirs(os.path.dirname(out), exist_ok=True)
tun.enable(True)
tun.tuning_enable(True)
tun.set_filename(out)
tun.set_max_tuning_duration(int(os.environ.get("GT_MS", "60")))
tun.set_max_tuning_iterations(int(os.environ.get("GT_IT", "6")))
for k, f in cases.items():
    t0 = time.time()
    f()
    torch.cuda.synchronize()
    print(f"tuned {k} in {time.time() - t0:.1f}s", flush=True)
tun.tuning_enable(False)
if hasattr(tun, "write_file"):
    tun.write_file(out)
for k, f in cases.items():
    t = bench(f)
    print(f"tuned    {k:52s} {t:7.3f} ms  {flops[k] / t / 1e9:7.1f} TFLOP/s   x{base[k] / t:.3f}", flush=True)
if os.path.exists(out):
    print(open(out).read())
# measured on MI355X (round 3): the default solution is the tuned one or within 1 % of it for all six shapes (1.35 - 1.47 PFLOP/s)
