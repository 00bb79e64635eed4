"""run-to-run determinism and agreement of the three column-sum routes at HunyuanVideo size (2 heads): one pass (attn64
MODE 3), dense + colsum64_kernel (forced: at 2 heads the size rule would pick the general pass), dense + the general
kernel's K-only pass.  This comparison found the missing lgkmcnt(0) in front of the general pass's barrier."""
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
for name, opts in (("dense + colsum64_kernel", dict(attn_fused_colsum=2, attn_colsum64=1)),
                   ("dense + general K-only pass", dict(attn_fused_colsum=2, attn_colsum64=2))):
    outs = [run(**opts) for _ in range(6)]
    print(f"{name:30s} run-to-run identical {all(torch.equal(outs[0], o) for o in outs[1:])}; elements off vs one pass: {[nbad(o, f) for o in outs]}")
