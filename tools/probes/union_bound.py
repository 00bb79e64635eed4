"""how many kept keys do ADJACENT query groups of one head share on the bench's C3 launches, and what would a
two-groups-against-the-union work item buy?  Builds the masks exactly as the bench's mask step does (dense_colsum_topk_mask on the
synthetic q, k, v: 5 % top keys + 1 % random + the 256 text columns) for KB_HEADS heads at the C3 size and counts, per adjacent pair
(g, g + 1):  shared = |A & B| / mean(|A|, |B|),  union = |A | B| / (|A| + |B|).
A union item gathers |A | B| rows once for two groups (gather bytes per useful flop x 2 * union) and runs |A | B| keys through both
groups' MFMAs (MFMA work x 2 * union, the masked share wasted).  Upper bound of what ANY gather saving can buy: the same launch with
every group of a head gathering THE SAME keys (KB_SAME_INDICES=1: all gathers hit L2) -- measured here beside the plain launch.
usage: KB_HEADS=4 python tools/probes/union_bound.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import chipmunk_amd  # noqa: E402,F401
import chipmunk_amd.ops as ops  # noqa: E402
from tools.kbench import timeit  # noqa: E402

dev = torch.device("cuda:0")
H = int(os.environ.get("KB_HEADS", "4"))
N, TXT = 119056, 256
g = torch.Generator(device=dev).manual_seed(1234)
q, k, v = [torch.randn(1, H, N, 128, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(3)]
_, l = torch.ops.chipmunk.dense_attn(q, k, v)
G = (N + 191) // 192
gt = (N - TXT) // 192                              # first query group that holds text rows: those keep every key (static rows)
groups = torch.ones(1, H, G, 1, dtype=torch.bool, device=dev)
groups[:, :, gt:] = False
static = torch.zeros(1, H, G, N, dtype=torch.bool, device=dev)
static[..., N - TXT:] = True                      # the text columns are kept by every group (local_voxels = 0 in configs/hunyuan_c3.yml)
static[:, :, gt:] = True
tk = int(128 * round(0.05 * N / 128))
o, mask, _ = ops.dense_colsum_topk_mask(q, k, v, l, tk, 0.01, groups, static, False)
mask = mask.view(H, G, -1)[:, :, :N]
cnt = mask.sum(-1).float()
a, b = mask[:, :-3], mask[:, 1:-2]                # (the last groups hold the text rows and keep every key: left out)
inter = (a & b).sum(-1).float()
mean_ab = 0.5 * (cnt[:, :-3] + cnt[:, 1:-2])
shared = (inter / mean_ab)
union = (cnt[:, :-3] + cnt[:, 1:-2] - inter) / (cnt[:, :-3] + cnt[:, 1:-2])
print(f"C3 masks of the bench's mask step, {H} heads x {G} groups, N = {N}: kept keys per group mean {cnt[:, :-2].mean().item():.0f}")
print(f"adjacent groups share {100 * shared.mean().item():.1f} % of their kept keys (min {100 * shared.min().item():.1f}, max {100 * shared.max().item():.1f}); "
      f"of which the {TXT} text columns alone are {100 * TXT / mean_ab.mean().item():.1f} %")
u = union.mean().item()
print(f"|A u B| / (|A| + |B|) = {u:.3f}: a two-group work item against the union gathers {u:.3f} of today's K/V rows "
      f"and issues {2 * u:.2f}x today's MFMAs ({100 * (1 - 1 / (2 * u)):.0f} % of them on masked scores)")
# what the saved gather can buy at most: profiles/r04d_key_sharing_bound.txt -- every group of a head gathering THE SAME keys (all gathers
# L2 hits, fetch 31.9 -> 3.7 GB per launch) runs 9.38 -> 8.96 ms = -4.5 %: the kernel is MFMA-issue bound, not gather bound
ceiling = 0.045
gain = ceiling * (1.0 - u) / (1.0 - 3.67 / 31.86)       # share of the all-hit saving a (1 - u) cut of the gathered rows can reach, at best
loss = 2 * u - 1.0                                       # extra MFMA + softmax work on the masked half of the union
print(f"bound on the launch: at most -{100 * gain:.1f} % from the gather side, at least +{100 * loss:.0f} % matrix / softmax work "
      f"=> {'worth building' if gain - loss >= 0.08 else 'NOT worth building (VERDICT threshold: >= 8 % on the launch)'}")
