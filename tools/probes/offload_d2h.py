#!/usr/bin/env python
"""Why does the device-to-host copy of MaybeOffloadedTensor.offload run at 10 GB/s in the Wan2.1 loop (rocprofv3 --memory-copy-trace) when a plain
pinned copy on the same box runs at 56 GB/s (tools/probes/h2d_bw.py)?  Times the class's own offload() and variations of its copy statement.
usage (GPU box): python tools/probes/offload_d2h.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import chipmunk_amd  # noqa: F401
from chipmunk_amd.util import config as cfg
from chipmunk_amd.util.storage import offloaded_tensor as ot

dev = torch.device("cuda:0")
cfg.reset_to_base()
G = cfg.GLOBAL_CONFIG
G["offloading"]["global_disable_offloading"] = False
G["offloading"]["attn.out_cache"] = True
G["offloading"]["keep_resident_if_fits"] = False
shape = (1, 12, 32760, 128)
x = torch.randn(shape, device=dev, dtype=torch.bfloat16)
xt = torch.randn(1, 32760, 12, 128, device=dev, dtype=torch.bfloat16).permute(0, 2, 1, 3)   # token-major storage viewed [B, H, N, D]
mb = x.numel() * 2 / 1e6


def timed(tag, fn, stream):
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(stream):
            s.record()
        fn()
        with torch.cuda.stream(stream):
            e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e))
    print(f"{tag:70s} {best:7.2f} ms = {mb / best:6.1f} GB/s", flush=True)


t = ot.MaybeOffloadedTensor("attn.out_cache", 0, torch.bfloat16, dev)
timed("MaybeOffloadedTensor.offload(contiguous [1,12,N,128])", lambda: t.offload(x), ot.offload_stream())
t2 = ot.MaybeOffloadedTensor("attn.out_cache", 1, torch.bfloat16, dev)
timed("MaybeOffloadedTensor.offload(token-major view)", lambda: t2.offload(xt), ot.offload_stream())
side = ot.offload_stream()
buf = torch.empty(x.numel(), dtype=torch.bfloat16, device="cpu", pin_memory=True)


def plain():
    with torch.cuda.stream(side):
        buf.copy_(x.reshape(-1), non_blocking=True)


def strided():
    with torch.cuda.stream(side):
        buf[: x.numel()].as_strided(x.shape, x.stride()).copy_(x, non_blocking=True)


def strided_tm():
    with torch.cuda.stream(side):
        buf[: xt.numel()].as_strided(xt.shape, xt.stride()).copy_(xt, non_blocking=True)


timed("flat pinned.copy_(x.reshape(-1))", plain, side)
timed("pinned.as_strided(shape, stride).copy_(x)   [4-d, contiguous]", strided, side)
timed("pinned.as_strided(shape, stride).copy_(xt)  [4-d, token-major]", strided_tm, side)
print("pinned?", buf.is_pinned(), buf[: x.numel()].as_strided(x.shape, x.stride()).is_pinned())

# many pinned buffers, as the Wan loop holds them (30 blocks x 2 invocations x (100.6 + 8.4 + 1.6 MB)): does the copy rate depend on which?
bufs = [torch.empty(x.numel(), dtype=torch.bfloat16, device="cpu", pin_memory=True) for _ in range(int(os.environ.get("NBUF", "120")))]
for i in (0, len(bufs) // 2, len(bufs) - 1):
    b = bufs[i]

    def d2h(b=b):
        with torch.cuda.stream(side):
            b.copy_(x.reshape(-1), non_blocking=True)

    def h2d(b=b):
        with torch.cuda.stream(side):
            x.reshape(-1).copy_(b, non_blocking=True)
    timed(f"D2H into pinned buffer #{i} of {len(bufs)}", d2h, side)
    timed(f"H2D from pinned buffer #{i} of {len(bufs)}", h2d, side)
# first copy into a never-used buffer vs the second
fresh = torch.empty(x.numel(), dtype=torch.bfloat16, device="cpu", pin_memory=True)
for rep in range(3):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        s.record(); fresh.copy_(x.reshape(-1), non_blocking=True); e.record()
    torch.cuda.synchronize()
    print(f"D2H into a fresh pinned buffer, copy #{rep}: {s.elapsed_time(e):7.2f} ms = {mb / s.elapsed_time(e):6.1f} GB/s")
