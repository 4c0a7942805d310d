#!/bin/bash
# Build libidf_gfx950.so (all HIP kernels + the C ABI) for gfx950.  hipcc cross-compiles without a GPU.
# Incremental by default (a file is recompiled when it or a header is newer than its object); `build.sh --clean`
# recompiles everything.  A failed compile fails the build: the stale object is removed first and every PID is waited for.
set -e
cd "$(dirname "$0")"
OUT=../libidf_gfx950.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form"
[ "$1" = "--clean" ] && rm -rf build
mkdir -p build
OBJS=""
PIDS=""
for f in gemm_conv gemm_big mlp_fused qkv_fused qkv640_fused geglu_fused attention attention4 attention4w attention8 norms scaleu misc convnext; do
  stale=0
  [ -f build/$f.o ] || stale=1
  for dep in $f.hip common.h gemm_core.h attn_core.h mw_prims.h mlpw_stream.inc qkvw_stream.inc gegluw_stream.inc qkv640w_stream.inc ../../include/idf.h; do
    [ $dep -nt build/$f.o ] && stale=1
  done
  if [ $stale = 1 ]; then
    rm -f build/$f.o
    /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o build/$f.o &
    PIDS="$PIDS $!"
  fi
  OBJS="$OBJS build/$f.o"
done
for pid in $PIDS; do
  wait $pid || { echo "build.sh: a compile failed" >&2; exit 1; }
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $OUT
echo "built $(readlink -f $OUT)"
