"""is the K-only pass glitch tied to the key INDEX (keys 32-63 of a 64-key window) or to the ADDRESS of the K rows?
K is a view that starts 32 rows (8 KiB) into a larger buffer: index and address bit 13 now disagree."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import chipmunk_amd
from chipmunk_amd import _native
dev = torch.device("cuda:0")
N, H = 119056, 2
g = torch.Generator(device=dev).manual_seed(7)
q, v = [torch.randn(1, H, N, 128, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(2)]
for shift in (0, 32, 16):
    kbig = torch.randn(1, H, N + 64, 128, device=dev, dtype=torch.bfloat16, generator=g)
    k = kbig[:, :, shift:shift + N]
    _, l = torch.ops.chipmunk.dense_attn(q, k, v)
    f = torch.ops.chipmunk.dense_colsum_attn(q, k, v, l)[1].float()
    halves = {0: 0, 1: 0}
    for i in range(12):
        _native.set_option("attn_fused_colsum", 5)
        torch.cuda.synchronize()
        cs = torch.ops.chipmunk.dense_colsum_attn(q, k, v, l)[1].float()
        torch.cuda.synchronize()
        _native.set_option("attn_fused_colsum", 0)
        bad = ((cs - f).abs() > 1e-5 + 2e-2 * f.abs()).nonzero().tolist()
        for ev in {(h, gi, j // 64, (j % 64) // 32) for _, h, gi, j in bad}:
            halves[ev[3]] += 1
    print(f"K view starting at row {shift} (data_ptr % 16384 = {k.data_ptr() % 16384}): events in keys 0-31: {halves[0]}, in keys 32-63: {halves[1]}")
