#!/usr/bin/env python
"""HBM-side traffic per launch of the hot kernels, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in
SEPARATE `rocprofv3 --pmc` passes (no trace domains) over tools/kbench.py at the BASELINE C2 single-block shapes;
FETCH_SIZE doubled (gfx950 reports half of a wide streaming read), WRITE_SIZE as reported; both are in KiB.
Writes profiles/<tag>_pmc_traffic.json, which bench.py reads for the `roofline.traffic` field.

usage (on the GPU box): python tools/collect_pmc_traffic.py r01"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {  # json key -> substring of the kernel name
    "mm1": "mm1_kernel", "mm2": "mm2_kernel", "scatter_add": "scatter_add_kernel",
    "csp_attn": "attn_kernel<true, true", "dense_attn": "attn_kernel<false, false, true, false>",
}


def one_pass(counter, what, extra_env=None):
    out = os.path.join(ROOT, "gpurun_out", f"pmc_{counter}_{'_'.join(what)}")
    subprocess.run(["rm", "-rf", out])
    env = dict(os.environ, TMPDIR="/tmp", **(extra_env or {}))
    subprocess.run(["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", out, "--",
                    sys.executable, os.path.join(ROOT, "tools", "kbench.py")] + what,
                   cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                per[r["Kernel_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: sum(v.values()) / len(v) for k, v in per.items()}


C3_KERNELS = {  # HunyuanVideo shapes (bench.py workload hunyuan_c3): kbench cases with all 24 heads
    "csp_128_attn_c3": "csp96_kernel<true>", "dense_attn_c3": "attn64_kernel<0>",   # long launches: attn96.hip / attn64.hip
    "dense_colsum_kernel_c3": "attn64_kernel<3>", "colsum_combine_c3": "cs_combine_kernel",   # dense_colsum_attn: fused pass + combine
}


def main(tag):
    res = {}
    c3_env = {"KB_HEADS": "24", "KB_COUNT_C3": "9088"}
    fetch, write = one_pass("FETCH_SIZE", ["csp_hunyuan", "colsum_hunyuan"], c3_env), one_pass("WRITE_SIZE", ["csp_hunyuan", "colsum_hunyuan"], c3_env)
    for key, pat in C3_KERNELS.items():
        f = [v for k, v in fetch.items() if pat in k]
        w = [v for k, v in write.items() if pat in k]
        if f and w:
            res[key] = {"FETCH_SIZE_KB_raw": f[0], "WRITE_SIZE_KB_raw": w[0], "hbm_bytes_per_launch": (2.0 * f[0] + w[0]) * 1024.0,
                        "note": "24 heads x 119 056 tokens; sparse: 9 088 sorted random keys per 192-query group (the bench's mean "
                                "count), in-place accumulate form; FETCH_SIZE doubled (gfx950 correction), WRITE_SIZE uncorrected"}
    if "dense_colsum_kernel_c3" in res and "colsum_combine_c3" in res:
        res["dense_colsum_attn_c3"] = {"hbm_bytes_per_launch": res["dense_colsum_kernel_c3"]["hbm_bytes_per_launch"] + res["colsum_combine_c3"]["hbm_bytes_per_launch"],
                                       "note": "dense pass with the column sums folded in (fp32 partial sums per 64-row wave block) + the combine of the partials"}
    for what, keys, fused in ((["mm1", "mm2", "scatter", "csp_flux", "dense_flux"], list(KERNELS), False),
                              (["mm1s"], ["mm1"], True)):
        fetch, write = one_pass("FETCH_SIZE", what), one_pass("WRITE_SIZE", what)
        for key in keys:
            pat = KERNELS[key]
            f = [v for k, v in fetch.items() if pat in k]
            w = [v for k, v in write.items() if pat in k]
            if not f or not w:
                continue
            name = "mm1+scatter_add" if fused else key
            res[name] = {"FETCH_SIZE_KB_raw": f[0], "WRITE_SIZE_KB_raw": w[0],
                         "hbm_bytes_per_launch": (2.0 * f[0] + w[0]) * 1024.0,
                         "note": "FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md HBM section); WRITE_SIZE "
                                 "uncorrected; separate --pmc passes over tools/kbench.py at the C2 single-block shapes"}
    # gpurun only brings gpurun_out/ back: write there too, then copy into profiles/ on the dev box
    for d in ("profiles", "gpurun_out"):
        json.dump(res, open(os.path.join(ROOT, d, f"{tag}_pmc_traffic.json"), "w"), indent=1)
    for k, v in res.items():
        print(f"{k:18s} {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB per launch")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
