#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of attn64_kernel<3> and topk_mask on the kbench mask step, current vs previous library
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp KB_HEADS=24
L=chipmunk_amd/lib/libchipmunk_hip.so
cp $L /tmp/cur.so
for t in cur prev cur prev; do
  if [ $t = cur ]; then cp /tmp/cur.so $L; else cp tools/bin/lib_prev.so $L; fi
  for ctr in FETCH_SIZE WRITE_SIZE; do
    out=/tmp/pmc_$t_$ctr; rm -rf $out
    (cd /tmp && rocprofv3 --pmc $ctr --output-format csv -d $out -- python $OLDPWD/tools/kbench.py maskstep_hunyuan > /dev/null 2>&1)
    python - $out $t $ctr <<'PY'
import csv, glob, sys, collections
per = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == sys.argv[3]:
            k = 'attn64<3>' if 'attn64_kernelILi3' in r['Kernel_Name'] or 'attn64_kernel<3>' in r['Kernel_Name'] else ('topk_mask' if 'topk_mask' in r['Kernel_Name'] else None)
            if k: per[k][r['Dispatch_Id']] += float(r['Counter_Value'])
for k, d in per.items():
    v = sum(d.values()) / len(d)
    print(f"{sys.argv[2]:5s} {k:10s} {sys.argv[3]} {v * 1024 / 1e9 * (2 if sys.argv[3] == 'FETCH_SIZE' else 1):8.2f} GB per launch ({len(d)} launches)")
PY
    rm -rf $out
  done
done
cp /tmp/cur.so $L
