#!/bin/bash
# The forms that were measured against the shipped kernels and are NOT in the product library: the work-balanced gathered attention launch
# (attn.hip BAL, option attn_balanced; -DCHIPMUNK_ATTN_PROBES) and the GEMM1 / GEMM2 forms: tile shapes (mm1_variant
# 1, 3-10; mm2_variant 1-16) and the two producer / consumer GEMM1 forms (20: mlp_pc.h, 128 x 256 tiles; 21: mlp_pp.h, DMA stream across
# tile boundaries).  Builds the library with them into tools/bin/forms/libchipmunk_hip.so -- same name, so that
#   LD_LIBRARY_PATH=tools/bin/forms CHIPMUNK_HIP_LIB=tools/bin/forms/libchipmunk_hip.so python tools/kbench.py mm1s --variants 0,20,21
# binds both the ctypes view and the torch registry to it (tests/test_gpu_mlp_forms.py runs the parity suite that way).
cd "$(dirname "$0")/../../.."
mkdir -p tools/bin/forms
src=""
for f in attn attn64 attn96 mlp indexed_io rowwise capi; do src="$src chipmunk_amd/csrc/$f.hip"; done
exec /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DCHIPMUNK_MM1_PROBES -DCHIPMUNK_ATTN_PROBES -Itools/probes/mm1_forms \
  -o tools/bin/forms/libchipmunk_hip.so $src
