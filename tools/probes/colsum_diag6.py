"""the K-only passes ALONE (probe option attn_fused_colsum = 5: no dense pass in front), synchronised before and after"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import chipmunk_amd
from chipmunk_amd import _native
dev = torch.device("cuda:0")
N, H = 119056, 2
g = torch.Generator(device=dev).manual_seed(7)
q, k, v = [torch.randn(1, H, N, 128, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(3)]
_, l = torch.ops.chipmunk.dense_attn(q, k, v)
f = torch.ops.chipmunk.dense_colsum_attn(q, k, v, l)[1].float()
def nbad(a):
    return int(((a - f).abs() > 1e-5 + 2e-2 * f.abs()).sum())
for name, extra in (("colsum64 alone", {}), ("general K-only pass alone", {"attn_colsum64": 2})):
    res = []
    for i in range(8):
        _native.set_option("attn_fused_colsum", 5)
        for o, val in extra.items():
            _native.set_option(o, val)
        torch.cuda.synchronize()
        cs = torch.ops.chipmunk.dense_colsum_attn(q, k, v, l)[1]
        torch.cuda.synchronize()
        _native.set_option("attn_fused_colsum", 0)
        for o in extra:
            _native.set_option(o, 0)
        a = cs.float()
        res.append(nbad(a))
        bad = ((a - f).abs() > 1e-5 + 2e-2 * f.abs()).nonzero().tolist()
        ev = sorted({(h, gi, j // 64, (j % 64) // 32) for _, h, gi, j in bad})
        if ev:
            print("    events (head, group, 64-key tile, half):", ev)
    print(f"{name:28s} elements off vs one-pass over 8 launches: {res}")
