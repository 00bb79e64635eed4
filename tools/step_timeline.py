#!/usr/bin/env python
"""GPU busy fraction and idle gaps of the steady state in a rocprofv3 --kernel-trace database of bench.py, any workload: the window
spans the last `steps` x `per_step` launches of the kernel whose name contains `marker` (e.g. FLUX: "attn_kernel<true, true" 57 per
step); inside it: union of kernel intervals / wall, the gaps by size class, kernel time by name.
usage: step_timeline.py <dir-or-db> <marker> <per_step> [steps]"""
import collections
import glob
import sqlite3
import sys


def main(path, marker, per_step, steps=10):
    dbp = path if path.endswith(".db") else glob.glob(path + "/**/*_results.db", recursive=True)[0]
    cur = sqlite3.connect(dbp).cursor()
    rows = list(cur.execute("select name,start,end from kernels order by start"))
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    need = steps * per_step + 1
    if len(marks) < need:
        raise SystemExit(f"only {len(marks)} launches of '{marker}'")
    a, b = marks[-need], marks[-1]
    wall = rows[b][1] - rows[a][1]
    busy, last_end, gaps = 0, rows[a][1], collections.Counter()
    agg, cnt = collections.Counter(), collections.Counter()
    pairs, pcnt = collections.Counter(), collections.Counter()
    for i in range(a, b):
        name, s, e = rows[i]
        agg[name[:100]] += e - s
        cnt[name[:100]] += 1
        if s > last_end:
            g = s - last_end
            if g >= 10000:
                pairs[(rows[i - 1][0][:48], name[:48])] += g
                pcnt[(rows[i - 1][0][:48], name[:48])] += 1
            gaps["<2us" if g < 2000 else "<5us" if g < 5000 else "<10us" if g < 10000 else "<50us" if g < 50000 else ">=50us"] += g
        busy += max(0, e - max(s, last_end))
        last_end = max(last_end, e)
    print(f"{steps} steps, {b - a} launches, wall {wall / 1e6:.2f} ms = {wall / 1e6 / steps:.3f} ms/step, busy {busy / wall * 100:.1f} %, {(b - a) / steps:.0f} launches/step")
    print("idle time by gap size: " + ", ".join(f"{k} {v / 1e6:.2f} ms" for k, v in sorted(gaps.items())))
    print("gaps >= 10 us by (kernel before -> kernel after):")
    for k, t in pairs.most_common(8):
        print(f"  {t / 1e6 / steps:7.3f} ms/step  {pcnt[k] / steps:5.1f} x {t / pcnt[k] / 1e3:6.1f} us  {k[0]}  ->  {k[1]}")
    for name, t in agg.most_common(14):
        print(f"  {t / 1e6 / steps:8.3f} ms/step  {cnt[name] / steps:6.1f} x {t / cnt[name] / 1e3:8.1f} us  {name}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 10)
