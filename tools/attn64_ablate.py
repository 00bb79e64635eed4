#!/usr/bin/env python
"""Prices the parts of the attn64.hip main loop: builds tools/bin/libchipmunk_a64_<mask>.so with -DA64_ABL=<mask> (parts of
the loop compiled out; results wrong, clock meaningful).  `build` here, then on the GPU box `run` swaps each library
in turn under chipmunk_amd/lib and times the HunyuanVideo dense launch (6 heads) with tools/kbench.py."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MASKS = [0, 32, 1, 2, 3, 4, 8, 12, 15, 16, 31]
SRC = [os.path.join(ROOT, "chipmunk_amd", "csrc", f) for f in ("attn.hip", "attn64.hip", "attn96.hip", "mlp.hip", "indexed_io.hip", "capi.hip")]
LIB = os.path.join(ROOT, "chipmunk_amd", "lib", "libchipmunk_hip.so")


def path(mask):
    return os.path.join(ROOT, "tools", "bin", f"libchipmunk_a64_{mask}.so")


if sys.argv[1] == "build":
    os.makedirs(os.path.join(ROOT, "tools", "bin"), exist_ok=True)
    # an argument is either an ablation mask (+ 4096 * DMA position) or a tag NAME:-DFOO=1:-DBAR=2 (free-form defines)
    procs = []
    for a in sys.argv[2:] or [str(m) for m in MASKS]:
        if ":" in a:
            tag, *defs = a.split(":")
        else:
            tag, defs = a, [f"-DA64_ABL={int(a) & 0xfff}", f"-DA64_DMA_POS={int(a) >> 12}"]
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                                       *defs, "-o", path(tag)] + SRC))
    assert all(p.wait() == 0 for p in procs)
else:
    keep = LIB + ".keep"
    shutil.copy(LIB, keep)
    try:
        for m in ([int(a) for a in sys.argv[2:]] or MASKS):
            if not os.path.exists(path(m)):
                continue
            shutil.copy(path(m), LIB)
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kbench.py"), "dense_hunyuan"], capture_output=True, text=True).stdout
            print(f"mask {m:2d}: {out.strip()}")
    finally:
        shutil.copy(keep, LIB)
        os.remove(keep)
