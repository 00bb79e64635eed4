cd $GRAFT_REPO_ROOT
out=gpurun_out/r04c_variants.txt; : > $out
export KB_LAYERS=8
for pr in 0 1 2 4; do
  echo "## mm1_probe=$pr" >> $out
  python tools/kbench.py mm1 --variants 0,1,3,4,5,6,7 --opt mm1_probe=$pr 2>&1 | grep variant >> $out
done
echo "## mm2 variants probe 0 / 4" >> $out
python tools/kbench.py mm2 --variants 0,1,2,5,6,10,13,14 2>&1 | grep variant >> $out
python tools/kbench.py mm2 --variants 0,1,2,5,6,10,13,14 --opt mm1_probe=4 2>&1 | grep variant >> $out
cat $out
