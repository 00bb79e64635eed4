#!/usr/bin/env python
"""SQ / LDS counters of one kernel under tools/kbench.py, one `rocprofv3 --pmc` pass per counter group (no trace
domains).  usage (GPU box): python tools/pmc_counters.py <kernel-substring> <kbench case> [ENV=VAL ...]"""
import collections
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = [
    ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVES", "GRBM_GUI_ACTIVE"],
    ["SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_SALU", "SQ_INSTS_LDS"],
    ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_ANY"],
    ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INST_CYCLES_VMEM", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS"],
    ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_UNALIGNED_STALL"],
    ["SQ_WAIT_ANY", "SQ_INSTS_VMEM", "SQ_ACTIVE_INST_MISC", "SQ_INST_LEVEL_LDS"],
]
pat, case = sys.argv[1], sys.argv[2]
env = dict(os.environ, TMPDIR="/tmp")
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    env[k] = v
for gi, grp in enumerate(GROUPS):
    out = os.path.join(ROOT, "gpurun_out", f"pmcg_{gi}")
    subprocess.run(["rm", "-rf", out])
    r = subprocess.run(["rocprofv3", "--pmc"] + grp + ["--output-format", "csv", "-d", out, "--", sys.executable,
                        os.path.join(ROOT, "tools", "kbench.py"), case], cwd="/tmp", env=env, capture_output=True, text=True)
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if pat in row["Kernel_Name"]:
                per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    if not per:
        print("group", gi, "no data", r.stderr[-300:])
    for c in grp:
        if c in per:
            v = per[c]
            print(f"{c:28s} {sum(v.values()) / len(v):16.0f}   ({len(v)} dispatches)")
