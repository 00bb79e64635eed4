#!/usr/bin/env python
"""Which engine runs MaybeOffloadedTensor.offload()'s device-to-host copies for the three tensors the Wan2.1 run offloads per layer (token-major
output cache 100 MB, bit-packed mask 8.4 MB, counts 8 KB), next to a busy compute stream?  Run under `rocprofv3 --kernel-trace --stats`:
blit kernels (__amd_rocclr_copyBuffer) appear in the kernel stats, SDMA copies do not.
usage (GPU box): python tools/probes/d2h_offload_engine.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import chipmunk_amd  # noqa: F401
from chipmunk_amd.util import config as cfg
from chipmunk_amd.util.storage import offloaded_tensor as ot

dev = torch.device("cuda:0")
cfg.reset_to_base()
G = cfg.GLOBAL_CONFIG
G["offloading"].update({"global_disable_offloading": False, "attn.out_cache": True, "attn.indices": True, "attn.counts": True,
                        "keep_resident_if_fits": False})
o_tm = torch.randn(1, 32760, 12, 128, device=dev).to(torch.bfloat16).permute(0, 2, 1, 3)      # token-major storage, [B, H, N, D] view
packed = torch.randint(0, 255, (1, 12, 171, 4095), device=dev, dtype=torch.uint8)
counts = torch.randint(0, 3000, (1, 12, 171), device=dev, dtype=torch.int32)
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
holders = [ot.MaybeOffloadedTensor("attn.out_cache", 0, torch.bfloat16, dev), ot.MaybeOffloadedTensor("attn.indices", 0, torch.uint8, dev),
           ot.MaybeOffloadedTensor("attn.counts", 0, torch.int32, dev)]
for rep in range(4):
    for _ in range(10):
        a @ a
    for h, t in zip(holders, (o_tm, packed, counts)):
        h.offload(t)
    for _ in range(10):
        a @ a
    for h in holders:
        h.load_async()
        h.load_async_wait()
    torch.cuda.synchronize()
print("done: 4 rounds x (3 offloads + 3 loads)")
