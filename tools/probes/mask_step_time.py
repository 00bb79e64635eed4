"""time of the mask step's fused operator (dense attention + column sums + top-k mask) at HunyuanVideo size, per kernel, from HIP
events around repeated calls: KB_HEADS heads (default 6)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import chipmunk_amd
from chipmunk_amd import ops
dev = torch.device("cuda:0")
H, N = int(os.environ.get("KB_HEADS", "6")), int(os.environ.get("KB_N", "119056"))
G = (N + 191) // 192
g = torch.Generator(device=dev).manual_seed(3)
q, k, v = [torch.randn(1, H, N, 128, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(3)]
_, l = torch.ops.chipmunk.dense_attn(q, k, v)
st = torch.rand(1, H, G, N, device=dev, generator=g) < 0.002
gr = torch.ones(1, H, G, 1, dtype=torch.bool, device=dev)
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ops.manual_seed(1)
full = t(lambda: ops.dense_colsum_topk_mask(q, k, v, l, int(os.environ.get("KB_TOPK", "5888")), 0.01, gr, st))
dense = t(lambda: torch.ops.chipmunk.dense_attn(q, k, v))
o, cs, _ = torch.ops.chipmunk.dense_colsum_attn(q, k, v, l)
tk = t(lambda: ops.topk_mask(cs[..., :G, :N], int(os.environ.get("KB_TOPK", "5888")), 0.01, gr, st), 5)
print(f"H={H}: dense_colsum_topk_mask {full:.2f} ms; dense_attn {dense:.2f} ms; ratio {full / dense:.3f}; topk_mask on cs {tk:.3f} ms")
