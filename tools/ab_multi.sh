#!/bin/bash
# same-box comparison of library builds: tools/ab_multi.sh "<kbench cases>" cur tools/bin/libA.so tools/bin/libB.so ...   ("cur" = the tree's library)
# extra kbench arguments through KB_ARGS
L=chipmunk_amd/lib/libchipmunk_hip.so
cp $L /tmp/cur.so
cases=$1; shift
for rep in 1 2 3; do
  for t in "$@"; do
    if [ "$t" = cur ]; then cp /tmp/cur.so $L; else cp $t $L; fi
    for c in $cases; do timeout 300 python tools/kbench.py $c $KB_ARGS 2>/dev/null | grep " us" | sed "s|^|$(basename $t)  |"; done
  done
done
cp /tmp/cur.so $L
