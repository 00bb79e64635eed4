#!/usr/bin/env python
"""Error of the gathered kernel against an fp64 reference at C3 size for inputs of growing score magnitude (q x scale): what folding
log2(e)/sqrt(D) into the bf16 Q fragments costs in accuracy.  Run once per library (tools/ab_lib.sh swaps them)."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import chipmunk_amd  # noqa: E402,F401
from chipmunk_amd import _native  # noqa: E402

dev = torch.device("cuda:0")
N, C = 119056, 9088
G = math.ceil(N / 192)
g = torch.Generator(device=dev).manual_seed(5)
q0, k, v = [torch.randn(1, 1, N, 128, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(3)]
inds = torch.zeros(1, 1, G, N, dtype=torch.int32, device=dev)
inds[0, 0, :, :C] = torch.rand(G, N, device=dev, generator=g).topk(C, dim=-1).indices.sort(dim=-1).values.to(torch.int32)
counts = torch.full((1, 1, G), C, dtype=torch.int32, device=dev)
_native.set_option("attn_csp96", 1)
for scale in (1.0, 2.0, 4.0, 6.0):
    q = (q0.float() * scale).to(torch.bfloat16)
    o = torch.ops.chipmunk.csp_128_attn(q, k, v, inds, counts)
    errs, rel = [], []
    for gi in (0, 7, 100, 333, 500, 619):
        rows = slice(gi * 192, (gi + 1) * 192)
        idx = inds[0, 0, gi, :C].long()
        s = (q[0, 0, rows].double() @ k[0, 0, idx].double().T) / math.sqrt(128)
        ref = torch.softmax(s, -1) @ v[0, 0, idx].double()
        d = (o[0, 0, rows].double() - ref).abs()
        errs.append(d.max().item())
        rel.append((d / (0.02 + 0.02 * ref.abs())).max().item())
    print(f"q x {scale}: max abs err {max(errs):.4f}  worst err / tolerance {max(rel):.3f}")
