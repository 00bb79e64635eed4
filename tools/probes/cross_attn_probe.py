"""Wan2.1 cross-attention shape (12 heads x 32 768 queries x 512 text keys) through chipmunk.dense_attn: the general kernel (default at
Nk < 1024) against the one-wave-per-SIMD kernel of attn64.hip (option attn_dense64 = 1).  usage: python tools/probes/cross_attn_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import chipmunk_amd  # noqa: E402,F401
from chipmunk_amd import _native  # noqa: E402
from tools.kbench import timeit  # noqa: E402

dev = torch.device("cuda:0")
H, NQ = 12, 32768
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn(1, H, NQ, 128, device=dev, dtype=torch.bfloat16, generator=g)
for NK in (512, 1024):
    k, v = [torch.randn(1, H, NK, 128, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(2)]
    flops = 4.0 * H * NQ * NK * 128
    outs = []
    for opt in (2, 1):
        _native.set_option("attn_dense64", opt)
        o, l = torch.ops.chipmunk.dense_attn(q, k, v)
        outs.append(o)
        ms = timeit(lambda: torch.ops.chipmunk.dense_attn(q, k, v), reps=10)
        print(f"Nk={NK:5d} attn_dense64={opt}: {ms * 1e3:8.1f} us  {flops / ms / 1e9:7.1f} TFLOP/s")
    _native.set_option("attn_dense64", 0)
    print("   max |diff| between the two kernels:", float((outs[0].float() - outs[1].float()).abs().max()))
    ms = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v), reps=5)
    print(f"Nk={NK:5d} torch SDPA          : {ms * 1e3:8.1f} us  {flops / ms / 1e9:7.1f} TFLOP/s")
