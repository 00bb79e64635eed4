#!/bin/bash
# same-box A/B of the HunyuanVideo gathered and dense launches: current library vs tools/bin/libchipmunk_hip_prev.so
L=chipmunk_amd/lib/libchipmunk_hip.so
cp $L /tmp/cur.so
for i in 1 2; do
  for c in ${@:-csp_hunyuan}; do echo -n "new  "; python tools/kbench.py $c; done
  cp tools/bin/libchipmunk_hip_prev.so $L
  for c in ${@:-csp_hunyuan}; do echo -n "prev "; python tools/kbench.py $c; done
  cp /tmp/cur.so $L
done
