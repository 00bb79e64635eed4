cd $GRAFT_REPO_ROOT
export KB_LAYERS=8
python tools/kbench.py mm1 mm1s mm2 csp_flux dense_flux topkd 2>&1 | tee gpurun_out/r04a_kbench.txt
for k in mm1s mm2 csp_flux; do
  pat=mm1_kernel; [ $k = mm2 ] && pat=mm2_kernel; [ $k = csp_flux ] && pat=attn_kernel
  echo "== $k" | tee -a gpurun_out/r04a_pmc.txt
  python tools/pmc_counters.py $pat $k KB_LAYERS=8 2>&1 | tee -a gpurun_out/r04a_pmc.txt
done
