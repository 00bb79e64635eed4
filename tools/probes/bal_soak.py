"""Hand-off soak of the work-balanced gathered launch (attn.hip, BAL) under UNEVEN load: a second stream keeps the CUs busy with GEMMs of
varying size while the balanced launch runs; every launch must give the plain launch's bits (a continued item is the uncut item's own
arithmetic).  Also soaks the row-split tail (run-to-run identity).  usage (probe-forms library, tools/probes/mm1_forms/build.sh):
  LD_LIBRARY_PATH=tools/bin/forms CHIPMUNK_HIP_LIB=$PWD/tools/bin/forms/libchipmunk_hip.so python tools/probes/bal_soak.py [launches]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import chipmunk_amd  # noqa: E402,F401
from chipmunk_amd import _native  # noqa: E402
from tools.kbench import sorted_random_indices  # noqa: E402

dev = torch.device("cuda:0")
N_LAUNCH = int(sys.argv[1]) if len(sys.argv) > 1 else 300
H, N, count = 24, 4352, 672
g = torch.Generator(device=dev).manual_seed(0)
q, k, v = [torch.randn(1, H, N, 128, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(3)]
G = (N + 191) // 192
inds = sorted_random_indices(H, G, N, count, N, g)
counts = torch.full((1, H, G), count, dtype=torch.int32, device=dev)
counts[0, 3, 5] = 96
counts[0, 7, 1] = 0
counts[0, 11, 20] = 1344
base = torch.randn(1, H, N, 128, device=dev, dtype=torch.bfloat16, generator=g)

_native.set_option("attn_row_split", 2)
_native.set_option("attn_balanced", 2)
plain = torch.ops.chipmunk.csp_attn_out(q, k, v, base, inds, counts, 1)
side = torch.cuda.Stream()
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
bad = {"balanced": 0, "row_split": 0}
ref_rs = None
for it in range(N_LAUNCH):
    with torch.cuda.stream(side):                      # uneven background load: GEMMs of three sizes, a few per launch
        for _ in range(1 + it % 3):
            n = (1024, 4096, 8192)[it % 3]
            a[:n, :n] @ a[:n, :n]
    _native.set_option("attn_balanced", 1)
    out = torch.ops.chipmunk.csp_attn_out(q, k, v, base, inds, counts, 1)
    _native.set_option("attn_balanced", 2)
    if not torch.equal(out, plain):
        bad["balanced"] += 1
    _native.set_option("attn_row_split", 1)
    rs = torch.ops.chipmunk.csp_attn_out(q, k, v, base, inds, counts, 1)
    _native.set_option("attn_row_split", 2)
    if ref_rs is None:
        ref_rs = rs
    elif not torch.equal(rs, ref_rs):
        bad["row_split"] += 1
torch.cuda.synchronize()
_native.set_option("attn_row_split", 0)
_native.set_option("attn_balanced", 0)
print(f"{N_LAUNCH} launches each under background GEMMs: balanced launches that differ from the plain launch's bits: {bad['balanced']}; "
      f"row-split launches that differ from the first one: {bad['row_split']}")
print("max |row-split - plain| =", float((ref_rs.float() - plain.float()).abs().max()))
