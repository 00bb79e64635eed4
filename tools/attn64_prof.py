#!/usr/bin/env python
"""Cycle anatomy of the attn64.hip main loop: builds the library with -DATTN64_PROF into tools/bin/libchipmunk_a64prof.so
(s_memtime at the segment boundaries of every tile, every wave of one mid-grid workgroup) and prints cycles per tile per
segment: wait+barrier | phase A (32 QK MFMAs + finish) | V-fragment wait | phase B gaps 0-11 | reference check |
phase B gaps 12-31 | rescale + K-read wait.  `--build-only` here, run on the GPU box.
`--colsum` profiles the fused dense + column-sum launch (MODE 3); `--mx` builds it with -DA64_CSUM_VALU=0 (column sums over the
matrix pipe, a measured negative result of round 4) into a second library for the same-box comparison."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MX = "--mx" in sys.argv
LIB = os.path.join(ROOT, "tools", "bin", "libchipmunk_a64prof_mx.so" if MX else "libchipmunk_a64prof.so")
SRC = [os.path.join(ROOT, "chipmunk_amd", "csrc", f) for f in ("attn.hip", "attn64.hip", "attn96.hip", "mlp.hip", "indexed_io.hip", "rowwise.hip", "capi.hip")]

if "--build-only" in sys.argv or not os.path.exists(LIB):
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DATTN64_PROF"] + (["-DA64_CSUM_VALU=0"] if MX else []) +
                          ["-o", LIB] + SRC)
    if "--build-only" in sys.argv:
        sys.exit(0)
import torch

lib = ctypes.CDLL(LIB)
dev = torch.device("cuda:0")
H, N = 6, 32768
g = torch.Generator(device=dev).manual_seed(0)
q, k, v = [torch.randn(1, H, N, 128, device=dev, dtype=torch.bfloat16, generator=g) for _ in range(3)]
o = torch.empty_like(q)
l = torch.empty(1, H, N, 1, device=dev, dtype=torch.float32)
st = (ctypes.c_int64 * 3)(H * N * 128, N * 128, 128)
P = lambda t: ctypes.c_void_p(t.data_ptr())
assert lib.chipmunk_set_option(b"attn_dense64", 1) == 0
for _ in range(3):
    rc = lib.chipmunk_dense_attn(P(q), P(k), P(v), st, st, st, P(o), P(l), 1, H, N, N, None)
    assert rc == 0, ctypes.c_char_p(lib.chipmunk_last_error()).value
if "--colsum" in sys.argv:
    G = (N + 191) // 192
    cs = torch.empty(1, H, G, N, device=dev, dtype=torch.bfloat16)
    l2 = torch.empty_like(l)
    for _ in range(3):
        rc = lib.chipmunk_dense_colsum_attn(P(q), P(k), P(v), st, st, st, P(l), P(o), P(cs), P(l2), 1, H, N, N, N, None)
        assert rc == 0, ctypes.c_char_p(lib.chipmunk_last_error()).value
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 32)()
assert lib.chipmunk_attn64_prof_read(buf) == 0
names = ["wait+bar", "phaseA", "vf wait", "B 0-11", "check", "B 12-31", "tail"]
for w in range(4):
    n = buf[w * 8 + 7]
    per = [buf[w * 8 + i] / max(n, 1) for i in range(7)]
    print(f"wave {w}: tiles {n}  " + "  ".join(f"{nm} {x:7.1f}" for nm, x in zip(names, per)) + f"   total {sum(per):7.1f}")
