#!/usr/bin/env python
"""Cycles per 64-key tile of the attn64.hip kernel from SQ counters (clock-independent, unlike wall time: ablations that
change the data change the power draw and with it the clock).  For every library tools/bin/libchipmunk_a64_<mask>.so
given (default: the product library): one rocprofv3 --pmc pass over tools/kbench.py dense_hunyuan with 2 heads.
usage (GPU box): python tools/attn64_cycles.py [mask ...]"""
import collections
import csv
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "chipmunk_amd", "lib", "libchipmunk_hip.so")
CTRS = ["SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"]
H, N = 2, 119056
WGR = int(os.environ.get("A64_WGROWS", "256"))   # 192 for the loader-wave experiment builds
WAVES = H * ((N + WGR - 1) // WGR) * 4
TILES = ((N + 63) // 64 + 3) // 4 * 4 + 1


def measure(tag):
    out = os.path.join(ROOT, "gpurun_out", "a64cyc")
    subprocess.run(["rm", "-rf", out])
    env = dict(os.environ, TMPDIR="/tmp", KB_HEADS=str(H))
    r = subprocess.run(["rocprofv3", "--pmc"] + CTRS + ["--output-format", "csv", "-d", out, "--", sys.executable,
                        os.path.join(ROOT, "tools", "kbench.py"), "dense_hunyuan"], cwd="/tmp", env=env, capture_output=True, text=True)
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "attn64_kernel<0>" in row["Kernel_Name"]:
                per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    tf = [l for l in r.stdout.splitlines() if "TFLOP" in l]
    vals = {c: sum(per[c].values()) / max(len(per[c]), 1) / WAVES / TILES * 4 for c in CTRS}   # quad-cycles -> cycles per wave per tile
    print(f"{tag:>8}: " + "  ".join(f"{c[3:]} {vals[c]:7.1f}" for c in CTRS) + "   " + (tf[0].split("us")[1].strip() if tf else ""))


masks = sys.argv[1:]
if not masks:
    measure("product")
else:
    keep = LIB + ".keep"
    shutil.copy(LIB, keep)
    try:
        for m in masks:
            shutil.copy(os.path.join(ROOT, "tools", "bin", f"libchipmunk_a64_{m}.so"), LIB)
            measure(m)
    finally:
        shutil.copy(keep, LIB)
        os.remove(keep)
