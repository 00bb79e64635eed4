"""hunt for run-to-run differences of attn96.hip at 24 heads: repeated launches, where the outputs differ"""
import math, sys, os
sys.path.insert(0, os.environ.get("PKG_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import chipmunk_amd
from chipmunk_amd import _native
dev = torch.device("cuda:0")
N, H, count = int(os.environ.get("DET_N_TOKENS", "119056")), int(os.environ.get("DET_HEADS", "24")), int(os.environ.get("DET_COUNT", "9088"))
G = (N + 191) // 192
g = torch.Generator(device=dev).manual_seed(3)
q, k, v = [torch.randn(1, H, N, 128, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(3)]
inds = torch.empty(1, H, G, count, dtype=torch.int32, device=dev)
for h in range(H):
    for g0 in range(0, G, 64):
        r = torch.rand(min(64, G - g0), N, device=dev, generator=g)
        inds[0, h, g0:g0 + r.shape[0]] = r.topk(count, dim=-1).indices.sort(-1).values.to(torch.int32)
counts = torch.full((1, H, G), count, dtype=torch.int32, device=dev)
for name, val in [a.split("=") for a in sys.argv[1:]]:
    _native.set_option(name, int(val))
ref = torch.ops.chipmunk.csp_128_attn(q, k, v, inds, counts)
torch.cuda.synchronize()
# DET_STRESS=1: a second stream copies 4 GB tensors around the launch (the loop's index loads are consumed under a counted vmcnt with
# LDS-DMAs behind them: DESIGN.md 4.1, "a third step was a race").  Weak as a latency stress: the gathered kernel owns every VGPR of the
# CUs it runs on, so the copy kernels mostly run before / after it, not beside it.
stress = os.environ.get("DET_STRESS") == "1"
if stress:
    side = torch.cuda.Stream()
    big_a = torch.empty(1 << 31, dtype=torch.bfloat16, device=dev)
    big_b = torch.empty_like(big_a)
bad = 0
for i in range(int(os.environ.get("RUNS", "40"))):
    if stress:
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(4):
                big_b.copy_(big_a)
    o = torch.ops.chipmunk.csp_128_attn(q, k, v, inds, counts)
    torch.cuda.synchronize()
    if not torch.equal(o, ref):
        bad += 1
        d = (o.float() - ref.float())
        nz = (d != 0) | torch.isnan(d)
        idx = nz.nonzero()
        hs, rows = idx[:, 1], idx[:, 2]
        items = sorted({(int(a), int(b) // 192) for a, b in zip(hs.tolist(), rows.tolist())})
        print(f"run {i}: {int(nz.sum())} elements differ, max |d| {float(d.abs().nan_to_num(1e9).max()):.4g}, nan {int(torch.isnan(o.float()).sum())}, "
              f"items (head, group) {items[:6]}{'...' if len(items) > 6 else ''}, rows in group {sorted({int(r) % 192 for r in rows.tolist()})[:12]}, "
              f"cols {sorted({int(c) for c in idx[:, 3].tolist()})[:8]}")
print("differing runs:", bad)
