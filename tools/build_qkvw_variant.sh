#!/bin/bash
# tools/build_qkvw_variant.sh <name> [VAR=value ...] [-- extra hipcc flags]: tools/ubench/v_<name>/libidf_gfx950.so whose qkv_fused.hip is
# built on a stream generated with the given tools/gen_qkvw_stream.py options (QW_LA, QW_PRE_DMA, QW_MAXV, QW_NO_EPI / QW_NO_DMA = 1).
set -e
name=$1; shift
envs=""; while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs="$envs $1"; shift; done
[ "$1" = "--" ] && shift
root="$(cd "$(dirname "$0")/.." && pwd)"
cd "$root/instancediffusion_amd/csrc"
mkdir -p build "$root/tools/ubench/v_$name"
env $envs python "$root/tools/gen_qkvw_stream.py" -o build/qkvw_$name.inc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form"
hipcc $FLAGS -DQKVW_STREAM_INC="\"build/qkvw_$name.inc\"" "$@" -c qkv_fused.hip -o build/qkv_fused_$name.o
OBJS=""
for f in gemm_conv gemm_big mlp_fused attention attention4 attention4w attention8 norms scaleu misc convnext; do OBJS="$OBJS build/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS build/qkv_fused_$name.o -o "$root/tools/ubench/v_$name/libidf_gfx950.so"
echo built v_$name
