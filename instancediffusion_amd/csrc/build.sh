#!/bin/bash
# Build libidf_gfx950.so (all HIP kernels + the C ABI) for gfx950.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
OUT=../libidf_gfx950.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form"
OBJS=""
for f in gemm_conv gemm_big attention attention2 norms scaleu misc convnext; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ common.h -nt build/$f.o ] || [ gemm_core.h -nt build/$f.o ] || [ attn_core.h -nt build/$f.o ] || [ ../../include/idf.h -nt build/$f.o ]; then
    mkdir -p build
    /opt/rocm/bin/hipcc $FLAGS -c $f.hip -o build/$f.o &
  fi
  OBJS="$OBJS build/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $OUT
echo "built $(readlink -f $OUT)"
