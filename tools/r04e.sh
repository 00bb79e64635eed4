cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python bench.py --workload flux_c2 --no-cpu-baseline > gpurun_out/r04e_flux.json 2> gpurun_out/r04e_flux.err
(cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04e_fluxprof -- python $GRAFT_REPO_ROOT/bench.py --workload flux_c2 --no-cpu-baseline --dense-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/r04e_fluxprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r04e_fluxprof.err)
python tools/rocprof_summary.py gpurun_out/r04e_fluxprof gpurun_out/r04e_flux.stats.txt
rm -rf gpurun_out/r04e_fluxprof
cat gpurun_out/r04e_flux.json | cut -c1-400
