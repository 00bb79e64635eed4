export TMPDIR=/tmp
root=$(pwd)
out=$root/gpurun_out/tl_hy; rm -rf $out
(cd /tmp && rocprofv3 --kernel-trace -d $out -- python $root/bench.py --no-cpu-baseline --no-legs --dense-steps 0 > $out.json 2> $out.err)
python tools/step_timeline.py $out "csp96_kernel" 58 10
out=$root/gpurun_out/tl_wan; rm -rf $out
(cd /tmp && rocprofv3 --kernel-trace -d $out -- python $root/bench.py --workload wan_c5 --no-cpu-baseline --dense-steps 0 > $out.json 2> $out.err)
python tools/step_timeline.py $out "csp96_kernel" 30 10
