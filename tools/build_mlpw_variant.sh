#!/bin/bash
# tools/build_mlpw_variant.sh <name> [VAR=value ...] [-- extra hipcc flags]: a library tools/ubench/v_<name>/libidf_gfx950.so whose
# mlp_fused.hip is built on a stream generated with the given tools/gen_mlpw_stream.py options (MW_LA, MW_PRE_DMA, MW_MAXV,
# MW_NO_VALU / MW_NO_DMA / MW_NO_MFMA = 1: timing experiments); the other objects are the shipped ones.
#   LD_LIBRARY_PATH=tools/ubench/v_<name> tools/ubench/mlp_harness 5
set -e
name=$1; shift
envs=""; while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs="$envs $1"; shift; done
[ "$1" = "--" ] && shift
root="$(cd "$(dirname "$0")/.." && pwd)"
cd "$root/instancediffusion_amd/csrc"
mkdir -p build "$root/tools/ubench/v_$name"
env $envs python "$root/tools/gen_mlpw_stream.py" -o build/mlpw_$name.inc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form"
hipcc $FLAGS -DMLPW_STREAM_INC="\"build/mlpw_$name.inc\"" "$@" -c mlp_fused.hip -o build/mlp_fused_$name.o
OBJS=""
for f in gemm_conv gemm_big attention attention4 attention4w attention8 norms scaleu misc convnext; do OBJS="$OBJS build/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS build/mlp_fused_$name.o -o "$root/tools/ubench/v_$name/libidf_gfx950.so"
echo built v_$name
