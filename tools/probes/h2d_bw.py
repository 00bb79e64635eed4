#!/usr/bin/env python
"""Pinned-host -> HBM copy rate on this box (what bounds the Wan2.1 workload when its attention caches live in host memory, as the
reference ships it): one hipMemcpyAsync of 100 MB (a layer's attention output cache) and of 1 GB, alone and under a running kernel.
usage (GPU box): python tools/probes/h2d_bw.py"""
import torch

dev = torch.device("cuda:0")
side = torch.cuda.Stream()
for mb in (100, 1000):
    n = mb * (1 << 20) // 2
    host = torch.empty(n, dtype=torch.bfloat16, pin_memory=True).normal_()
    gpu = torch.empty(n, dtype=torch.bfloat16, device=dev)
    for busy in (False, True):
        a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if busy:
                for _ in range(20):
                    a @ a
            with torch.cuda.stream(side):
                s.record()
                gpu.copy_(host, non_blocking=True)
                e.record()
            torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e))
        print(f"{mb:5d} MB pinned host -> HBM {'under a GEMM' if busy else 'alone       '}: {best:7.2f} ms = {mb * 1.048576 / best:6.1f} GB/s")
    h2 = torch.empty(n, dtype=torch.bfloat16, pin_memory=True)
    for busy in (False, True):
        best = 1e9
        for _ in range(5):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if busy:
                for _ in range(20):
                    a @ a
            with torch.cuda.stream(side):
                s.record()
                h2.copy_(gpu, non_blocking=True)
                e.record()
            torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e))
        print(f"{mb:5d} MB HBM -> pinned host {'under a GEMM' if busy else 'alone       '}: {best:7.2f} ms = {mb * 1.048576 / best:6.1f} GB/s")
