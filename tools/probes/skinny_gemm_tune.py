"""The sparse MLP's probe GEMM, fc1(block_mean(x)): [34, 3072] x [3072, 12288] + bias (reference modules/mlp.py:62) -- what torch dispatches to, what
TunableOp finds, and the transposed formulation (out^T = W x^T), each over 8 rotating weight sets (fc1 comes from HBM in the loop).
usage: python tools/probes/skinny_gemm_tune.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
dev = torch.device("cuda:0")
M, K, N, L = 34, 3072, 12288, 8
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, K, device=dev, dtype=torch.bfloat16, generator=g)
ws = [(torch.randn(N, K, device=dev, dtype=torch.bfloat16, generator=g) * 0.02) for _ in range(L)]
b = torch.randn(N, device=dev, dtype=torch.bfloat16, generator=g)
st = {"i": 0}
def nxt():
    st["i"] = (st["i"] + 1) % L
    return ws[st["i"]]
cases = {"F.linear(x, W, b)            ": lambda: torch.nn.functional.linear(x, nxt(), b),
         "(W @ x^T + b[:, None])^T      ": lambda: torch.addmm(b[:, None], nxt(), x.t()).t(),
         "mv-like: W @ x^T, no bias     ": lambda: torch.mm(nxt(), x.t())}
def bench(fn, n=40):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for k, f in cases.items():
    t = bench(f)
    print(f"default  {k} {t:7.1f} us  {N * K * 2 / t / 1e6:6.2f} TB/s of weights", flush=True)
import torch.cuda.tunable as tun
tun.enable(True); tun.tuning_enable(True); tun.set_filename("gpurun_out/tunableop_skinny.csv")
tun.set_max_tuning_duration(200); tun.set_max_tuning_iterations(20)
for k, f in cases.items():
    f(); torch.cuda.synchronize()
tun.tuning_enable(False)
for k, f in cases.items():
    t = bench(f)
    print(f"tuned    {k} {t:7.1f} us  {N * K * 2 / t / 1e6:6.2f} TB/s of weights", flush=True)
