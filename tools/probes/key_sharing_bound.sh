#!/bin/bash
# Upper bound of what a key-range work order could buy the gathered kernel: the same launch (24 heads x 119 056 tokens,
# 7 296 sorted random keys per 192-query group) with every group of a head gathering (a) its own keys, (b) THE SAME keys -- every gather of
# a co-resident workgroup then hits L2, which is what phasing co-resident items over a shared key window approaches at best.  Time from
# tools/kbench.py, fabric-side fetch from a separate rocprofv3 --pmc FETCH_SIZE pass (doubled per MI355X_MICROARCH.md).
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp KB_HEADS=24
for same in 0 1; do
  export KB_SAME_INDICES=$same
  echo "== KB_SAME_INDICES=$same"
  python tools/kbench.py csp_hunyuan 2>/dev/null | grep " us "
  out=gpurun_out/pmc_keyshare_$same; rm -rf $out
  (cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OLDPWD/$out -- python $OLDPWD/tools/kbench.py csp_hunyuan > /dev/null 2>&1)
  python - $out <<'PY'
import csv, glob, sys, collections
per = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'csp96_kernel' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE':
            per[r['Dispatch_Id']] += float(r['Counter_Value'])
if per:
    v = sum(per.values()) / len(per)
    print(f"   csp96_kernel FETCH_SIZE {v:.0f} KiB raw per launch -> {2 * v * 1024 / 1e9:.2f} GB fetched per launch (x2 gfx950 correction), {len(per)} launches")
PY
  rm -rf $out
done
