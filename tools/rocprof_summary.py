#!/usr/bin/env python
"""Turn a rocprofv3 --kernel-trace --stats result database (rocpd sqlite) into a small text summary for profiles/."""
import glob
import sqlite3
import sys


def main(path, out=None, top=30):
    dbs = glob.glob(path + "/**/*_results.db", recursive=True) if not path.endswith(".db") else [path]
    lines = []
    for dbp in dbs:
        cur = sqlite3.connect(dbp).cursor()
        rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
        total = sum(r[2] for r in rows)
        lines.append(f"# {dbp.split('/')[-1]}: {len(rows)} kernels, total GPU kernel time {total/1e3:.3f} ms (durations in us)")
        lines.append(f"{'kernel':100s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
        for name, calls, tot, avg, pct in rows[:top]:
            lines.append(f"{name[:100]:100s} {calls:7d} {tot:12.1f} {avg:10.2f} {pct:6.2f}")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
