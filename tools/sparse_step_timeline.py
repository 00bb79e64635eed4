#!/usr/bin/env python
"""GPU timeline of the HunyuanVideo sparse steps in a rocprofv3 --kernel-trace database of bench.py: a step's window runs from
its layer-0 dense attention launch to the next step's; inside it, kernel time by name, idle gaps and the busy fraction.
usage: sparse_step_timeline.py <dir-or-db> [steps-from-the-end]"""
import collections
import glob
import sqlite3
import sys


def main(path, steps=3):
    dbp = path if path.endswith(".db") else glob.glob(path + "/**/*_results.db", recursive=True)[0]
    cur = sqlite3.connect(dbp).cursor()
    rows = list(cur.execute("select name,start,end from kernels order by start"))
    # sparse steps: 2 dense launches (layers 0, 1), then 58 csp96 launches
    dense = [i for i, r in enumerate(rows) if "attn64_kernel<0>" in r[0]]
    csp = [i for i, r in enumerate(rows) if "csp96_kernel" in r[0]]
    starts = []
    for a, b in zip(dense, dense[1:]):
        n_between = sum(1 for i in csp if a < i < b)
        if n_between == 0 and starts and starts[-1] == a:
            continue
        if n_between == 0:      # a = layer 0, b = layer 1 of the same step
            starts.append(a)
    # keep the steps whose next start exists and which hold exactly 58 csp launches
    wins = []
    for a, b in zip(starts, starts[1:]):
        if sum(1 for i in csp if a < i < b) == 58:
            wins.append((a, b))
    wins = wins[-steps:]
    print(f"{len(wins)} sparse step windows")
    agg, cnt = collections.Counter(), collections.Counter()
    busy = wall = idle = nl = 0
    biggaps = collections.Counter()
    for a, b in wins:
        wall += rows[b][1] - rows[a][1]
        for i in range(a, b):
            name, s, e = rows[i]
            agg[name[:110]] += e - s
            cnt[name[:110]] += 1
            busy += e - s
            g = rows[i + 1][1] - e
            if g > 0:
                idle += g
                biggaps[(name[:50], rows[i + 1][0][:50])] += g
            nl += 1
    n = len(wins)
    print(f"step {wall/1e6/n:.2f} ms; kernels {busy/1e6/n:.2f} ms ({busy/wall:.1%}); idle gaps {idle/1e6/n:.2f} ms over {nl/n:.0f} launches")
    for k, v in agg.most_common(30):
        print(f"{v/1e6/n:9.3f} ms/step {cnt[k]/n:7.1f}/step {v/cnt[k]/1e3:9.1f} us  {k}")
    print("largest idle gaps (after -> before):")
    for (a, b), v in biggaps.most_common(12):
        print(f"{v/1e6/n:9.3f} ms/step  {a} -> {b}")


if __name__ == "__main__":
    main(sys.argv[1], *(int(a) for a in sys.argv[2:]))
