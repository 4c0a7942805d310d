#!/bin/bash
# tools/build_attn_variant.sh <name> <extra hipcc flags...>: a library tools/ubench/libidf_<name>.so whose attention4w.hip is built
# with the trace and the given extra flags (timing experiments; the other objects are the shipped ones)
set -e
name=$1; shift
cd "$(dirname "$0")/../instancediffusion_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form"
hipcc $FLAGS -DIDF_ATTN4W_TRACE "$@" -c attention4w.hip -o build/attention4w_$name.o
OBJS=""
for f in gemm_conv gemm_big mlp_fused attention attention4 attention8 norms scaleu misc convnext; do OBJS="$OBJS build/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS build/attention4w_$name.o -o ../../tools/ubench/libidf_$name.so
echo built libidf_$name.so
