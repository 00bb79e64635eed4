"""run-to-run identity of the attention kernels at HunyuanVideo size (6 heads): dense (attn64 / general), gathered
(attn96 / general), one-pass column sums"""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import chipmunk_amd
from chipmunk_amd import _native
dev = torch.device("cuda:0")
N, H, count = 119056, int(os.environ.get('DET_HEADS', '6')), 9088
G = (N + 191) // 192
g = torch.Generator(device=dev).manual_seed(3)
q, k, v = [torch.randn(1, H, N, 128, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(3)]
inds = torch.empty(1, H, G, count, dtype=torch.int32, device=dev)
for h in range(H):
    for g0 in range(0, G, 64):
        r = torch.rand(min(64, G - g0), N, device=dev, generator=g)
        inds[0, h, g0:g0 + r.shape[0]] = r.topk(count, dim=-1).indices.sort(-1).values.to(torch.int32)
counts = torch.full((1, H, G), count, dtype=torch.int32, device=dev)
def rep(name, fn, n=4, **opts):
    n = max(n, int(os.environ.get('DET_N', '0')))
    for o, val in opts.items():
        _native.set_option(o, val)
    try:
        outs = [fn() for _ in range(n)]
    finally:
        for o in opts:
            _native.set_option(o, 0)
    torch.cuda.synchronize()
    same = all(all(torch.equal(a, b) for a, b in zip(outs[0], o)) for o in outs[1:])
    worst = max(float((a.float() - b.float()).abs().max()) for o in outs[1:] for a, b in zip(outs[0], o))
    print(f"{name:36s} run-to-run identical: {same}   (max abs difference {worst:.3g})")
rep("dense_attn, attn64", lambda: torch.ops.chipmunk.dense_attn(q, k, v))
rep("dense_attn, general kernel", lambda: torch.ops.chipmunk.dense_attn(q, k, v), n=3, attn_dense64=2)
rep("csp_128_attn, attn96", lambda: (torch.ops.chipmunk.csp_128_attn(q, k, v, inds, counts),))
rep("csp_128_attn, general kernel", lambda: (torch.ops.chipmunk.csp_128_attn(q, k, v, inds, counts),), attn_csp96=2)
_, l = torch.ops.chipmunk.dense_attn(q, k, v)
rep("dense_colsum_attn, one pass", lambda: torch.ops.chipmunk.dense_colsum_attn(q, k, v, l))
rep("csp_128_attn, attn96, running-maximum loop", lambda: (torch.ops.chipmunk.csp_128_attn(q, k, v, inds, counts),), n=6, attn_nomax=2)
rep("csp_128_attn, attn96 (soak)", lambda: (torch.ops.chipmunk.csp_128_attn(q, k, v, inds, counts),), n=int(os.environ.get("DET_SOAK", "8")))
from chipmunk_amd import ops
st = (torch.rand(1, H, G, N, device=dev, generator=g) < 0.002)
gr = torch.ones(1, H, G, 1, dtype=torch.bool, device=dev)
rep("dense_colsum_topk_mask (no cs)", lambda: ops.dense_colsum_topk_mask(q, k, v, l, 5888, 0.0, gr, st), n=3)
qk = torch.randn(N, 3 * H * 128, device=dev, dtype=torch.bfloat16, generator=g)
rep("qkv_split_norm", lambda: tuple(ops.qkv_split_norm(qk, None, None, H)), n=3)
