"""does the K-only pass glitch depend on which dense kernel ran before it?"""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import chipmunk_amd
from chipmunk_amd import _native
dev = torch.device("cuda:0")
N, H = 119056, 2
g = torch.Generator(device=dev).manual_seed(7)
q, k, v = [torch.randn(1, H, N, 128, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(3)]
_, l = torch.ops.chipmunk.dense_attn(q, k, v)
def run(**opts):
    for o, val in opts.items():
        _native.set_option(o, val)
    try:
        return torch.ops.chipmunk.dense_colsum_attn(q, k, v, l)[1].float()
    finally:
        for o in opts:
            _native.set_option(o, 0)
def nbad(a, b):
    return int(((a - b).abs() > 1e-5 + 2e-2 * b.abs()).sum())
f = run()
for name, opts in (("attn64 dense + colsum64", dict(attn_fused_colsum=2)),
                   ("general dense + colsum64", dict(attn_fused_colsum=2, attn_dense64=2, attn_colsum64=1)),
                   ("general dense + CSONLY", dict(attn_fused_colsum=2, attn_dense64=2, attn_colsum64=2)),
                   ("attn64 dense, running max + colsum64", dict(attn_fused_colsum=2, attn_nomax=2))):
    res = [nbad(run(**opts), f) for _ in range(6)]
    print(f"{name:40s} elements off vs one-pass over 6 runs: {res}")
