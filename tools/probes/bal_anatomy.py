"""Timing anatomy of the work-balanced gathered launch (attn.hip, BAL): a few shapes, balanced vs plain, accumulate vs plain output.
usage (the balanced form lives in the probe-forms library, tools/probes/mm1_forms/build.sh):
  LD_LIBRARY_PATH=tools/bin/forms CHIPMUNK_HIP_LIB=$PWD/tools/bin/forms/libchipmunk_hip.so python tools/probes/bal_anatomy.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import chipmunk_amd  # noqa: E402,F401
from chipmunk_amd import _native  # noqa: E402
from tools.kbench import timeit, sorted_random_indices  # noqa: E402

dev = torch.device("cuda:0")


def run(H, N, count, form):
    g = torch.Generator(device=dev).manual_seed(0)
    q, k, v = [torch.randn(1, H, N, 128, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(3)]
    G = (N + 191) // 192
    inds = sorted_random_indices(H, G, N, count, N, g)
    counts = torch.full((1, H, G), count, dtype=torch.int32, device=dev)
    o = torch.zeros_like(q)
    out = []
    for bal, rs in ((2, 2), (1, 2), (2, 0)):          # plain, balanced, row-split tail (by shape)
        _native.set_option("attn_balanced", bal)
        _native.set_option("attn_row_split", rs)
        if form == "inplace":
            ms = timeit(lambda: torch.ops.chipmunk.csp_attn(q, k, v, o, inds, counts, 1), reps=10)
        else:
            ms = timeit(lambda: torch.ops.chipmunk.csp_128_attn(q, k, v, inds, counts), reps=10)
        out.append(ms * 1e3)
    _native.set_option("attn_balanced", 0)
    _native.set_option("attn_row_split", 0)
    print(f"H={H:3d} N={N:5d} items={H*G:4d} keys={count:5d} {form:8s}: plain {out[0]:7.1f} us   balanced {out[1]:7.1f} us   row-split tail {out[2]:7.1f} us")


for form in ("inplace", "plain"):
    run(16, 6144, 672, form)    # 512 items = one whole item per workgroup: no cuts
    run(24, 4352, 672, form)    # FLUX: 552 items
    run(24, 4352, 1344, form)
    run(8, 6144, 672, form)     # 256 items: half the slots
    run(24, 4352, 2688, form)   # 84 tiles per item
    run(28, 4352, 672, form)    # 644 items: 132 in the second round (the thirds would not fit: no row split by shape)
