# D2H copy beside a GEMM loop under runtime settings: does any of them take the copy off the blit kernel?  (GPU box) -> gpurun_out/d2h_env.log
R=$GRAFT_REPO_ROOT
rm -f $R/gpurun_out/d2h_env.log
for v in "A=1" "HSA_ENABLE_SDMA=1" "GPU_FORCE_BLIT_COPY_SIZE=0" "GPU_FORCE_BLIT_COPY_SIZE=1" "DEBUG_HIP_FORCE_ASYNC_QUEUE=1" "AMD_DIRECT_DISPATCH=0" "DEBUG_CLR_LIMIT_BLIT_WG=4" "GPU_BLIT_ENGINE_TYPE=1"; do
  echo "== $v" >> $R/gpurun_out/d2h_env.log
  env $v timeout 90 python $R/tools/probes/d2h_engine.py d2h busy 2>/dev/null | tail -1 >> $R/gpurun_out/d2h_env.log
  echo "rc=$?" >> $R/gpurun_out/d2h_env.log
done
cat $R/gpurun_out/d2h_env.log
