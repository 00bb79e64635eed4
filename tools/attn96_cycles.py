#!/usr/bin/env python
"""Cycles per 32-key tile and wave of the attn96.hip kernel from SQ counters, for the product library or for ablation builds
(-DA96_ABL=<mask>: parts of the loop compiled out).  `build <mask>...` here; on the GPU box `python tools/attn96_cycles.py
<mask>...` (0 = product).  HunyuanVideo gathered launch, 2 heads, 7 296 keys per group."""
import collections, csv, glob, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "chipmunk_amd", "lib", "libchipmunk_hip.so")
sys.path.insert(0, ROOT)
from chipmunk_amd.build import HIP_SOURCES
SRC = [os.path.join(ROOT, "chipmunk_amd", "csrc", f) for f in HIP_SOURCES]
path = lambda m: os.path.join(ROOT, "tools", "bin", f"libchipmunk_a96_{m}.so")
if sys.argv[1:2] == ["build"]:
    os.makedirs(os.path.join(ROOT, "tools", "bin"), exist_ok=True)
    procs = [subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", f"-DA96_ABL={m}", "-o", path(m)] + SRC)
             for m in sys.argv[2:]]
    assert all(p.wait() == 0 for p in procs)
    sys.exit(0)
CTRS = ["SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"]
H, N, C = 2, 119056, 7296
WAVES, TILES = H * ((N + 191) // 192) * 2, (((C + 31) // 32 + 3) // 4) * 4 + 1
keep = LIB + ".keep"
shutil.copy(LIB, keep)
try:
    for m in sys.argv[1:] or ["0"]:
        if m != "0":
            shutil.copy(path(m), LIB)
        out = os.path.join(ROOT, "gpurun_out", "a96cyc")
        subprocess.run(["rm", "-rf", out])
        env = dict(os.environ, TMPDIR="/tmp", KB_HEADS=str(H), CHIPMUNK_AMD_OPTIONS="attn_csp96=1")
        r = subprocess.run(["rocprofv3", "--pmc"] + CTRS + ["--output-format", "csv", "-d", out, "--", sys.executable,
                            os.path.join(ROOT, "tools", "kbench.py"), "csp_hunyuan"], cwd="/tmp", env=env, capture_output=True, text=True)
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if "csp96" in row["Kernel_Name"]:
                    per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
        vals = {c: sum(per[c].values()) / max(len(per[c]), 1) / WAVES / TILES * 4 for c in CTRS}
        print(f"{m:>4}: " + "  ".join(f"{c[3:]} {vals[c]:7.1f}" for c in CTRS), flush=True)
        shutil.copy(keep, LIB)
finally:
    shutil.copy(keep, LIB)
    os.remove(keep)
