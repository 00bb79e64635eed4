#!/bin/bash
# run a command with the current library and with tools/bin/libchipmunk_hip_prev.so on the same box: tools/ab_lib.sh <cmd...>
L=chipmunk_amd/lib/libchipmunk_hip.so
cp $L /tmp/cur.so
echo "== new"; "$@"
cp tools/bin/libchipmunk_hip_prev.so $L
echo "== prev"; "$@"
cp /tmp/cur.so $L
