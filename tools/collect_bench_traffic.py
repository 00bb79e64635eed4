#!/usr/bin/env python
"""HBM-side traffic per launch of the attention kernels ON THE BENCH'S OWN LAUNCHES (24 heads, the ragged key counts the module's
mask pipeline produces), as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE `rocprofv3 --pmc` passes (no
trace domains) over `bench.py --steps 2 --warmup 3 --no-legs --dense-steps 0` (step 0 dense, step 1 mask recompute, steps 2-4
sparse); FETCH_SIZE doubled (gfx950 reports half of a wide streaming read), WRITE_SIZE as reported; both are in KiB.
Writes profiles/<tag>_pmc_traffic.json (and gpurun_out/, which is what comes back from the GPU box); bench.py reads it for
`roofline.traffic` and stamps the file name into `roofline.traffic_source`.

usage (on the GPU box, in the same gpurun call as the final bench): python tools/collect_bench_traffic.py r04 [hunyuan_c3|flux_c2|wan_c5 ...]
(all workloads' entries go into ONE profiles/<tag>_pmc_traffic.json; flux / wan keys are the ones bench.py's pmc_traffic() looks up)"""
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


WORKLOADS = {
    # workload -> (bench.py arguments, {json key: [kernel-name substrings]}, note)
    "hunyuan_c3": (["--steps", "2", "--warmup", "3", "--no-legs", "--dense-steps", "0", "--no-cpu-baseline", "--no-step-caching"], OPS,
                   "bench.py's own launches (hunyuan_c3: 24 heads x 119 056 tokens, ragged module-generated key counts)"),
    "flux_c2": (["--workload", "flux_c2", "--steps", "4", "--warmup", "12", "--dense-steps", "0", "--no-cpu-baseline"],
                {"mm1+scatter_add": ["mm1_kernel<128, 64, 2, 2, false"], "mm2": ["mm2_kernel"], "csp_attn": ["attn_kernel<true, true, false, false"],
                 "topk_delta_indices": ["topk_indices_kernel"], "block_mean": ["block_mean_kernel"]},
                "bench.py's own launches (flux_c2: 24 heads x 4 352 tokens, 672 kept keys; MLP 34 / 30 groups, module-generated index lists)"),
    "wan_c5": (["--workload", "wan_c5", "--steps", "2", "--warmup", "12", "--dense-steps", "0", "--no-cpu-baseline"],
               {"mm1_fp8_wan": ["mm1_kernel<128, 64, 2, 2, true"], "mm2_wan": ["mm2_kernel"], "csp_128_attn_c3_wan": ["csp96_kernel"],
                "dense_attn_c3_wan": ["attn64_kernel<0>"], "dense_colsum_topk_mask_c3_wan": ["attn64_kernel<3>", "topk_mask_kernel"]},
               "bench.py's own launches (wan_c5: Wan2.1 1.3B shapes, 12 heads x 32 760 tokens, fp8 GEMM1 M 32 768 / K 1 536 / F 8 960)"),
}


# rocprofv3 --pmc segfaults on the wan_c5 bench itself (ROCm 7.2; the workload's pinned-host copies on side streams): its two GEMM
# kernels are profiled on tools/kbench.py's launches at the same shapes instead, and the entry says so
WORKLOADS["wan_c5_kbench"] = (["@kbench", "fp8_wan", "mm2_wan", "csp_hunyuan"],   # (the gathered kernel at 12 heads x 32 760 tokens, 8 832 keys per group: the bench's mean)
                              {"mm1_fp8_wan": ["mm1_kernel<128, 64, 2, 2, true"], "mm2_wan": ["mm2_kernel"], "csp_128_attn_c3_wan": ["csp96_kernel"]},
                              "tools/kbench.py launches at the wan_c5 shapes (M 32 768, K 1 536, F 8 960, keep 0.3; same buffers every launch), NOT "
                              "the bench's own launches: rocprofv3 --pmc crashes on that workload")


# round 5: the bench's OWN launches of the Wan2.1 line with the caches resident (no pinned-host copies on side streams -- what crashed the
# counter passes) and eight of the thirty blocks: the kernels and their operands are the full run's
WORKLOADS["wan_c5_resident"] = (["--workload", "wan_c5", "--layers", "8", "--steps", "2", "--warmup", "12", "--dense-steps", "0", "--no-cpu-baseline", "--no-legs"],
                                {"mm1_fp8_wan": ["mm1_kernel<128, 64, 2, 2, true"], "mm2_wan": ["mm2_kernel"], "csp_128_attn_c3_wan": ["csp96_kernel"],
                                 "dense_attn_c3_wan": ["attn64_kernel<0>"], "dense_colsum_topk_mask_c3_wan": ["attn64_kernel<3>", "topk_mask_kernel"]},
                                "bench.py's own launches (wan_c5 with WAN_RESIDENT=1 and 8 of the 30 blocks: 12 heads x 32 760 tokens, fp8 GEMM1 M 32 768 / K 1 536 / "
                                "F 8 960; the offloaded run's pinned-host copies crash rocprofv3 --pmc)")


def one_pass(counter, cmd):
    out = os.path.join(ROOT, "gpurun_out", f"pmcb_{counter}")
    subprocess.run(["rm", "-rf", out])
    env = dict(os.environ, TMPDIR="/tmp")
    if "--layers" in cmd and "wan_c5" in cmd:
        env["WAN_RESIDENT"] = "1"
    if "kbench.py" in cmd[0]:
        env.update(KB_HEADS="12", KB_N="32760", KB_COUNT_C3="8832")     # only the csp_hunyuan case reads these: the Wan2.1 sequence
    subprocess.run(["timeout", "1200", "rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable] + cmd,
                   cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                per[r["Kernel_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: (sum(v.values()) / len(v), len(v)) for k, v in per.items()}


def main(tag, workloads):
    res = {}
    path = os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json")
    if os.path.exists(path):
        res = json.load(open(path))
    for wl in workloads:
        args, ops, what = WORKLOADS[wl]
        cmd = [os.path.join(ROOT, "tools", "kbench.py")] + args[1:] if args[0] == "@kbench" else [os.path.join(ROOT, "bench.py")] + args
        fetch, write = one_pass("FETCH_SIZE", cmd), one_pass("WRITE_SIZE", cmd)
        for key, pats in ops.items():
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
                            "note": what + "; FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md HBM section), WRITE_SIZE uncorrected; "
                                    "separate --pmc passes"}
    for d in ("profiles", "gpurun_out"):
        json.dump(res, open(os.path.join(ROOT, d, f"{tag}_pmc_traffic.json"), "w"), indent=1)
    for k, v in res.items():
        print(f"{k:32s} {v['hbm_bytes_per_launch'] / 1e9:9.3f} GB per launch")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r04", sys.argv[2:] or ["hunyuan_c3"])
