#!/bin/bash
# same-box comparison of library builds tools/bin/libchipmunk_a64_<tag>.so: tools/ab_variants.sh "<kbench cases>" tag1 tag2 ...
L=chipmunk_amd/lib/libchipmunk_hip.so
cp $L /tmp/cur.so
cases=$1; shift
for rep in 1 2; do
  for t in "$@"; do
    cp tools/bin/libchipmunk_a64_$t.so $L
    for c in $cases; do echo -n "$t  "; timeout 120 python tools/kbench.py $c 2>/dev/null | grep variant; done
  done
done
cp /tmp/cur.so $L
