#!/bin/bash
# same-box A/B of the HunyuanVideo dense launch: current library vs tools/bin/<name>.so (default libchipmunk_a64_0.so)
BASE=${1:-libchipmunk_a64_0.so}
L=chipmunk_amd/lib/libchipmunk_hip.so
cp $L /tmp/cur.so
for i in 1 2; do
  echo -n "new  "; python tools/kbench.py dense_hunyuan
  cp tools/bin/$BASE $L; echo -n "base "; python tools/kbench.py dense_hunyuan; cp /tmp/cur.so $L
done
