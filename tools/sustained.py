#!/usr/bin/env python
"""Does a kernel slow down under sustained load (clock / power management)?  Times consecutive blocks of 50 launches of
GEMM1+scatter and GEMM2 alternating (the bench loop's duty), with 8 rotating layer sets, for a few seconds."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import chipmunk_amd  # noqa: E402,F401

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
M, K, F, keep, L = 4352, 3072, 12288, 4096, 8
a = torch.randn(M, K, device=dev, dtype=torch.bfloat16, generator=g)
bias = torch.zeros(F, device=dev, dtype=torch.bfloat16)
sets = []
for _ in range(L):
    w1 = (torch.randn(F, K, device=dev, generator=g) * 0.02).to(torch.bfloat16)
    cache = torch.randn(F, M, device=dev, dtype=torch.bfloat16, generator=g)
    packed = torch.randn(M, F, device=dev, dtype=torch.bfloat16, generator=g) * 0.1
    w2t = (torch.randn(F, K, device=dev, generator=g) * 0.02).to(torch.bfloat16)
    out = torch.zeros(M, K, device=dev, dtype=torch.bfloat16)
    sets.append((w1, cache, packed, w2t, out))
G = M // 128
inds = torch.stack([torch.randperm(F, device=dev, generator=g)[:keep].sort().values for _ in range(G)])
inds = torch.nn.functional.pad(inds, (0, F - keep)).to(torch.int32).contiguous()
counts = torch.full((G,), keep, dtype=torch.int32, device=dev)
for blk in range(40):
    s1, e1, e2 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t1 = t2 = 0.0
    for i in range(50):
        w1, cache, packed, w2t, out = sets[i % L]
        s1.record()
        torch.ops.chipmunk.csp_mlp_mm1_scatter(a, w1, packed, bias, cache, inds, counts)
        e1.record()
        torch.ops.chipmunk.csp_mlp_mm2(packed, w2t, inds, counts, out)
        e2.record()
        e2.synchronize()
        t1 += s1.elapsed_time(e1)
        t2 += e1.elapsed_time(e2)
    if blk % 4 == 0:
        print(f"block {blk:2d}: mm1+scatter {t1 / 50 * 1e3:6.1f} us   mm2 {t2 / 50 * 1e3:6.1f} us")
