#!/bin/bash
# Evidence for profiles/ from ONE GPU box: counter traffic on the bench's own launches, the driver-style bench line, and the rocprofv3
# kernel-trace summary of the same command.   usage: tools/evidence.sh <tag> <workload> [<workload> ...]
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
tag=$1; shift
for wl in "$@"; do
  twl=$wl; [ $wl = wan_c5 ] && twl=wan_c5_resident     # (the offloaded Wan run's pinned-host copies crash rocprofv3 --pmc: counters on the resident 8-block run)
  python tools/collect_bench_traffic.py $tag $twl > gpurun_out/${tag}_traffic_$wl.txt 2>&1
  cp gpurun_out/${tag}_pmc_traffic.json profiles/${tag}_pmc_traffic.json 2>/dev/null
  arg="--workload $wl"; [ $wl = hunyuan_c3 ] && arg=""
  python bench.py $arg > gpurun_out/${tag}_bench_$wl.json 2> gpurun_out/${tag}_bench_$wl.err
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $OLDPWD/gpurun_out/${tag}_prof_$wl -- python $OLDPWD/bench.py $arg --no-cpu-baseline > $OLDPWD/gpurun_out/${tag}_bench_${wl}_profiled_run.json 2> /dev/null)
  python tools/rocprof_summary.py gpurun_out/${tag}_prof_$wl gpurun_out/${tag}_bench_$wl.stats.txt > /dev/null
  rm -rf gpurun_out/${tag}_prof_$wl
  echo "== $wl"; cut -c1-220 gpurun_out/${tag}_bench_$wl.json; head -8 gpurun_out/${tag}_bench_$wl.stats.txt
done
