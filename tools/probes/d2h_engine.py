#!/usr/bin/env python
"""Which engine runs a 100 MB device-to-host copy issued on a side stream while the compute stream is busy -- an SDMA engine or a blit kernel
(__amd_rocclr_copyBuffer) on a compute queue?  Run under `rocprofv3 --kernel-trace --stats`: the blit kernel shows up in the kernel stats,
an SDMA copy does not.  Prints the copy's own duration and the slowdown of a concurrent GEMM loop.
usage (GPU box): python tools/probes/d2h_engine.py [d2h|h2d] [idle|busy]"""
import sys
import torch

what = sys.argv[1] if len(sys.argv) > 1 else "d2h"
load = sys.argv[2] if len(sys.argv) > 2 else "busy"
dev = torch.device("cuda:0")
x = torch.randn(1, 12, 32760, 128, device=dev).to(torch.bfloat16)
buf = torch.empty(x.numel(), dtype=torch.bfloat16, device="cpu", pin_memory=True)
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
side = torch.cuda.Stream()
mb = x.numel() * 2 / 1e6


def gemms(n):
    for _ in range(n):
        a @ a


def copy():
    with torch.cuda.stream(side):
        if what == "d2h":
            buf.copy_(x.reshape(-1), non_blocking=True)
        else:
            x.reshape(-1).copy_(buf, non_blocking=True)


gemms(20); copy(); torch.cuda.synchronize()
# GEMM loop alone
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); gemms(40); e1.record(); torch.cuda.synchronize()
alone = e0.elapsed_time(e1)
res = []
for rep in range(5):
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    if load == "busy":
        gemms(10)
    side.wait_stream(torch.cuda.current_stream()) if load == "idle" else None
    with torch.cuda.stream(side):
        s0.record()
    copy()
    with torch.cuda.stream(side):
        s1.record()
    if load == "busy":
        gemms(30)
    e1.record()
    torch.cuda.synchronize()
    res.append((s0.elapsed_time(s1), e0.elapsed_time(e1)))
c = sorted(r[0] for r in res)[2]
t = sorted(r[1] for r in res)[2]
print(f"{what} {load}: copy {c:.2f} ms = {mb / c:.1f} GB/s; 40 GEMMs alone {alone:.2f} ms, with the copy beside them {t:.2f} ms")
