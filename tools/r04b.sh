cd $GRAFT_REPO_ROOT
out=gpurun_out/r04b_probes.txt; : > $out
for env in "KB_LAYERS=8" "KB_LAYERS=1" "KB_LAYERS=8 KB_SAME_INDICES=1"; do
  for pr in 0 1 2 4 3; do
    echo "## $env mm1_probe=$pr" >> $out
    env $env python tools/kbench.py mm1 mm2 --opt mm1_probe=$pr 2>&1 | grep variant >> $out
  done
done
cat $out
