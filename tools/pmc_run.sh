#!/bin/bash
# usage: tools/pmc_run.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...] -- <command...>
# One rocprofv3 --pmc pass per counter group (never combined with trace domains); prints per-kernel counter means.
tag=$1; shift
groups=()
while [ "$1" != "--" ]; do groups+=("$1"); shift; done
shift
export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-$(pwd)}
i=0
for g in "${groups[@]}"; do
  out=$root/gpurun_out/pmc_${tag}_$i
  rm -rf $out
  rocprofv3 --pmc $g --output-format csv -d $out -- "$@" > $out.log 2>&1
  python - "$out" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        per[(r['Kernel_Name'], r['Dispatch_Id'])][r['Counter_Name']] += float(r['Counter_Value'])
    for (k, d), cs in per.items():
        for c, v in cs.items():
            acc[k][c].append(v)
for k, cs in acc.items():
    if 'attn_kernel' not in k and 'mm1' not in k and 'mm2' not in k: continue
    print(k[:110])
    for c, vs in sorted(cs.items()):
        print(f"    {c:32s} n={len(vs):3d} mean={sum(vs)/len(vs):.4g}")
PY
  i=$((i+1))
done
