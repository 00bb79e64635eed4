#!/usr/bin/env python
"""Per-step GPU time breakdown of the last `steps` bench steps from a rocprofv3 --kernel-trace result database:
kernel time by name, idle gaps between kernels, busy fraction.  usage: step_breakdown.py <dir-or-db> [steps] [mm1-per-step]"""
import collections
import glob
import sqlite3
import sys


def main(path, steps=20, per_step=55):
    dbp = path if path.endswith(".db") else glob.glob(path + "/**/*_results.db", recursive=True)[0]
    cur = sqlite3.connect(dbp).cursor()
    rows = list(cur.execute("select name,start,end from kernels order by start"))
    mm1 = [i for i, r in enumerate(rows) if "mm1_kernel" in r[0]]
    first = mm1[-steps * per_step]
    # step boundary: walk back from the first mm1 of the window to the start of that step (first kernel after the
    # previous step's last mm2/scatter) -- approximate by starting at the mm1 itself and ending one step-length later
    # the timed region ends with the GEMM2 that follows the last GEMM1 (bench.py's probes / work sums come after it)
    last = next(i for i in range(mm1[-1], len(rows)) if "mm2_kernel" in rows[i][0])
    t0, t1 = rows[first][1], rows[last][2]
    agg, cnt = collections.Counter(), collections.Counter()
    busy = 0
    for name, s, e in rows[first:last + 1]:
        agg[name[:100]] += e - s
        cnt[name[:100]] += 1
        busy += e - s
    gaps = [rows[i + 1][1] - rows[i][2] for i in range(first, last)]
    print(f"window {(t1-t0)/1e6:.2f} ms = {(t1-t0)/1e6/steps:.3f} ms/step; busy {busy/1e6/steps:.3f} ms/step "
          f"({busy/(t1-t0):.1%}); idle gaps {sum(g for g in gaps if g > 0)/1e6/steps:.3f} ms/step over {len(gaps)/steps:.0f} launches/step")
    for k, v in agg.most_common(40):
        print(f"{v/1e6/steps:8.3f} ms/step {cnt[k]/steps:7.1f}/step {v/cnt[k]/1e3:9.1f} us  {k}")


if __name__ == "__main__":
    main(sys.argv[1], *(int(a) for a in sys.argv[2:]))
