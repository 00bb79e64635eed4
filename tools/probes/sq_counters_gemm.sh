cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/r05a_sq_raw.txt; rm -f $out
run() { echo "== $*" >> $out; timeout 600 python tools/pmc_counters.py "$@" >> $out 2>&1; }
run "mm1_kernel<128, 64, 2, 2, false" mm1 KB_KEEP=3840 KB_LAYERS=8
run "mm2_kernel" mm2 KB_KEEP=3840 KB_LAYERS=8
run "mm1_kernel<128, 64, 2, 2, true" fp8_wan
run "attn_kernel<true, true" csp_flux
tail -60 $out
