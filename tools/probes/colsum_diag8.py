"""with the -DCS64_DEBUG library: do the events coincide with a change of a wave's K fragments (accumulator registers
a[192:255]) between the start of a tile and the start of its last pass?"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import chipmunk_amd
from chipmunk_amd import _native
L = _native.lib()
dev = torch.device("cuda:0")
N, H = 119056, 2
G = (N + 191) // 192
G4 = (G + 3) // 4
g = torch.Generator(device=dev).manual_seed(7)
q, k, v = [torch.randn(1, H, N, 128, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(3)]
_, l = torch.ops.chipmunk.dense_attn(q, k, v)
f = torch.ops.chipmunk.dense_colsum_attn(q, k, v, l)[1].float()
buf = (ctypes.c_uint * (4 + 4 * 256))()
L.chipmunk_cs64_debug_read(buf, 1)
for i in range(10):
    _native.set_option("attn_fused_colsum", 5)
    torch.cuda.synchronize()
    cs = torch.ops.chipmunk.dense_colsum_attn(q, k, v, l)[1].float()
    torch.cuda.synchronize()
    _native.set_option("attn_fused_colsum", 0)
    L.chipmunk_cs64_debug_read(buf, 1)
    bad = ((cs - f).abs() > 1e-5 + 2e-2 * f.abs()).nonzero().tolist()
    ev = sorted({(h, gi, j // 64, (j % 64) // 32) for _, h, gi, j in bad})
    n = buf[0]
    regs = []
    for e in range(min(n, 8)):
        wid, t, meta = buf[4 + 4 * e], buf[5 + 4 * e], buf[6 + 4 * e]
        bh, wg = wid // G4, wid % G4
        regs.append((bh, wg * 4 + (meta & 0xff), t, (meta >> 8) & 0xff, (meta >> 16) & 0xff))
    n2 = buf[1]
    cross = []
    for e in range(min(n2, 6)):
        wid, t, meta = buf[4 + 4 * (128 + e)], buf[5 + 4 * (128 + e)], buf[6 + 4 * (128 + e)]
        cross.append((wid // G4, (wid % G4) * 4 + (meta & 0xff), t, (meta >> 8) & 0xff, (meta >> 16) & 0xff))
    print(f"          waves whose fragments differ from wave 0's at tile start: {n2} {cross}")
    print(f"launch {i}: wrong sums at (head, group, tile, half) {ev}; fragment changes seen {n}: (head, group, tile, lanes block0, lanes block1) {regs}")
