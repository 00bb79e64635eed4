#!/usr/bin/env python
"""HBM-side traffic per launch of the attention kernels ON THE BENCH'S OWN LAUNCHES (24 heads, the ragged key counts the module's
mask pipeline produces), as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE `rocprofv3 --pmc` passes (no
trace domains) over `bench.py --steps 2 --warmup 3 --no-legs --dense-steps 0` (step 0 dense, step 1 mask recompute, steps 2-4
sparse); FETCH_SIZE doubled (gfx950 reports half of a wide streaming read), WRITE_SIZE as reported; both are in KiB.
Writes profiles/<tag>_pmc_traffic.json (and gpurun_out/, which is what comes back from the GPU box); bench.py reads it for
`roofline.traffic` and stamps the file name into `roofline.traffic_source`.

usage (on the GPU box, in the same gpurun call as the final bench): python tools/collect_bench_traffic.py r03"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OPS = {  # json key -> kernel-name substrings whose per-launch bytes add up to the op
    "csp_128_attn_c3": ["csp96_kernel<true>"],
    "dense_attn_c3": ["attn64_kernel<0>"],
    "dense_colsum_topk_mask_c3": ["attn64_kernel<3>", "topk_mask_kernel<120, true, true>"],
    "dense_colsum_kernel_c3": ["attn64_kernel<3>"],
}


def one_pass(counter, cmd):
    out = os.path.join(ROOT, "gpurun_out", f"pmcb_{counter}")
    subprocess.run(["rm", "-rf", out])
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable] + cmd,
                   cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                per[r["Kernel_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: (sum(v.values()) / len(v), len(v)) for k, v in per.items()}


def main(tag):
    cmd = [os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "3", "--no-legs", "--dense-steps", "0", "--no-cpu-baseline"]
    fetch, write = one_pass("FETCH_SIZE", cmd), one_pass("WRITE_SIZE", cmd)
    res = {}
    for key, pats in OPS.items():
        tot, parts = 0.0, {}
        for pat in pats:
            f = [v for k, v in fetch.items() if pat in k]
            w = [v for k, v in write.items() if pat in k]
            if not f or not w:
                tot = None
                break
            b = (2.0 * f[0][0] + w[0][0]) * 1024.0
            parts[pat] = {"FETCH_SIZE_KB_raw": f[0][0], "WRITE_SIZE_KB_raw": w[0][0], "launches_profiled": f[0][1], "hbm_bytes_per_launch": b}
            tot += b
        if tot is not None:
            res[key] = {"hbm_bytes_per_launch": tot, "kernels": parts,
                        "note": "bench.py's own launches (hunyuan_c3: 24 heads x 119 056 tokens, ragged module-generated key counts); "
                                "FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md HBM section), WRITE_SIZE uncorrected; "
                                "separate --pmc passes"}
    for d in ("profiles", "gpurun_out"):
        json.dump(res, open(os.path.join(ROOT, d, f"{tag}_pmc_traffic.json"), "w"), indent=1)
    for k, v in res.items():
        print(f"{k:28s} {v['hbm_bytes_per_launch'] / 1e9:9.2f} GB per launch")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r03")
