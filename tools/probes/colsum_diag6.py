"""the K-only passes ALONE (probe option attn_fused_colsum = 5: no dense pass in front), synchronised before and after.
NOTE the selection by size: at 2 heads `use_colsum64` is false and the DEFAULT K-only pass is the general kernel's
(attn.hip); attn_colsum64 = 1 forces colsum64_kernel, = 2 forces the general pass."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import chipmunk_amd
from chipmunk_amd import _native
dev = torch.device("cuda:0")
N = 119056
for H in (2, 6):
    g = torch.Generator(device=dev).manual_seed(7)
    q, k, v = [torch.randn(1, H, N, 128, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(3)]
    _, l = torch.ops.chipmunk.dense_attn(q, k, v)
    f = torch.ops.chipmunk.dense_colsum_attn(q, k, v, l)[1].float()
    for name, which in (("colsum64_kernel (forced)", 1), ("general K-only pass (forced)", 2)):
        res = []
        for i in range(8):
            _native.set_option("attn_fused_colsum", 5)
            _native.set_option("attn_colsum64", which)
            torch.cuda.synchronize()
            cs = torch.ops.chipmunk.dense_colsum_attn(q, k, v, l)[1].float()
            torch.cuda.synchronize()
            _native.set_option("attn_fused_colsum", 0)
            _native.set_option("attn_colsum64", 0)
            res.append(int(((cs - f).abs() > 1e-5 + 2e-2 * f.abs()).sum()))
        print(f"H={H} {name:30s} elements off vs one-pass over 8 launches: {res}")
    del q, k, v, l, f
