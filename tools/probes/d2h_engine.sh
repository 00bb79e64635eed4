# Which engine copies?  usage (GPU box): bash tools/probes/d2h_engine.sh   -> gpurun_out/d2h.log
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -f $R/gpurun_out/d2h.log
for probe in "d2h_offload_engine.py" "d2h_engine.py d2h busy"; do
  rm -rf /tmp/prof_off
  timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/prof_off -- python $R/tools/probes/$probe > /tmp/out_off.txt 2>/dev/null
  echo "== $probe" >> $R/gpurun_out/d2h.log
  cat /tmp/out_off.txt >> $R/gpurun_out/d2h.log
  python $R/tools/rocprof_summary.py /tmp/prof_off | cut -c1-170 | head -12 >> $R/gpurun_out/d2h.log
done
cat $R/gpurun_out/d2h.log
