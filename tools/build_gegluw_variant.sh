#!/bin/bash
# tools/build_gegluw_variant.sh <name> [VAR=value ...] [-- extra hipcc flags]: tools/ubench/v_<name>/libidf_gfx950.so whose geglu_fused.hip
# is built on a stream generated with the given tools/gen_gegluw_stream.py options (GW_LA, GW_PRE_DMA, GW_MAXV, GW_NO_EPI / GW_NO_DMA = 1).
#   IDF_LIB_PATH=tools/ubench/v_<name>/libidf_gfx950.so python tools/geglu_ab.py
set -e
name=$1; shift
envs=""; while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs="$envs $1"; shift; done
[ "$1" = "--" ] && shift
root="$(cd "$(dirname "$0")/.." && pwd)"
cd "$root/instancediffusion_amd/csrc"
mkdir -p build "$root/tools/ubench/v_$name"
env $envs python "$root/tools/gen_gegluw_stream.py" -o build/gegluw_$name.inc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form"
hipcc $FLAGS -DGEGLUW_STREAM_INC="\"build/gegluw_$name.inc\"" "$@" -c geglu_fused.hip -o build/geglu_fused_$name.o
OBJS=""
for f in gemm_conv gemm_big mlp_fused qkv_fused attention attention4 attention4w attention8 norms scaleu misc convnext; do OBJS="$OBJS build/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS build/geglu_fused_$name.o -o "$root/tools/ubench/v_$name/libidf_gfx950.so"
echo built v_$name
