"""The kept-column counts the FLUX bench's SparseDiffMlp modules really produce (per 128-row group), after a few steps of the bench's own
loop: mean, spread and max / mean per layer -- GEMM2 is ONE round of tiles whose k loops are as long as their group's count.
usage: python tools/probes/flux_mlp_counts.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda:0")
timer = bench.KernelTimer()
step, dense_step, desc, core = bench.build_flux(dev, 8, timer, whole_block=False)
for i in range(14):
    step(i)
torch.cuda.synchronize()
import gc  # noqa: E402
from chipmunk_amd.modules import SparseDiffMlp  # noqa: E402
mods = [o for o in gc.get_objects() if isinstance(o, SparseDiffMlp)]
for m in mods[:8]:
    c = m.storage.get_counts()[0].flatten().float().cpu()
    print(f"groups {c.numel():3d}  mean {c.mean().item():7.1f}  min {int(c.min())}  max {int(c.max())}  max/mean {c.max().item() / c.mean().item():.3f}  std/mean {c.std().item() / c.mean().item():.3f}   sorted: {sorted(int(x) for x in c)[::4]}")
