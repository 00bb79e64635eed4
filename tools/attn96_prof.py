#!/usr/bin/env python
"""Cycle anatomy of the attn96.hip main loop: builds the library with -DATTN96_PROF into tools/bin/libchipmunk_a96prof.so
and prints cycles per 32-key tile for both waves of workgroup 700: the counted wait + barrier at the top of a tile | its 48
MFMA slots (finer splits cost more than they tell: every s_memtime read drains the LDS queue)."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "bin", "libchipmunk_a96prof.so")
import glob
SRC = sorted(glob.glob(os.path.join(ROOT, "chipmunk_amd", "csrc", "*.hip")))   # (not via chipmunk_amd.build: importing the package would load the product library beside this one)
if "--build-only" in sys.argv or not os.path.exists(LIB):
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DATTN96_PROF", "-o", LIB] + SRC)
    if "--build-only" in sys.argv:
        sys.exit(0)
import torch

lib = ctypes.CDLL(LIB)
dev = torch.device("cuda:0")
H, N, C = 2, 119056, 7296
g = torch.Generator(device=dev).manual_seed(0)
q, k, v = [torch.randn(1, H, N, 128, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(3)]
o = torch.empty_like(q)
G = (N + 191) // 192
inds = torch.zeros(1, H, G, N, dtype=torch.int32, device=dev)
for h in range(H):
    inds[0, h, :, :C] = torch.rand(G, N, device=dev, generator=g).topk(C, dim=-1).indices.sort(dim=-1).values.to(torch.int32)
counts = torch.full((1, H, G), C, dtype=torch.int32, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
assert lib.chipmunk_set_option(b"attn_csp96", 1) == 0
for _ in range(3):
    rc = lib.chipmunk_csp_128_attn(P(q), P(k), P(v), P(o), P(inds), P(counts), 1, H, N, N, N, None)
    assert rc == 0, ctypes.c_char_p(lib.chipmunk_last_error()).value
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
assert lib.chipmunk_attn96_prof_read(buf) == 0
names = ["wait+bar", "48 slots"]
for w in range(2):
    n = buf[w * 8 + 7]
    per = [buf[w * 8 + i] / max(n, 1) for i in range(2)]
    print(f"wave {w}: tiles {n}  " + "  ".join(f"{nm} {x:7.1f}" for nm, x in zip(names, per)) + f"   total {sum(per):7.1f}"
          f"   | prologue (entry -> first tile) {buf[w * 8 + 2]} cycles = {buf[w * 8 + 2] / max(sum(per), 1):.1f} tiles, "
          f"epilogue (loop end -> stores landed) {buf[w * 8 + 3]} cycles = {buf[w * 8 + 3] / max(sum(per), 1):.1f} tiles")
