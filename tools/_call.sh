#!/bin/bash
# scratch: GPU call script of the moment (see tools/gpu_call.sh for the runner)
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for kt in 4 8 16 32; do
  echo "== IDF_RING_SLICE_KT=$kt"
  IDF_RING_SLICE_KT=$kt timeout 300 tools/ubench/small_shapes 20 3 16 | grep -E "ring256 +[0-9.]+ \(1\)|forward-weighted GEMM" | awk '{ if ($0 ~ /forward-weighted/) print; else print $0 }' | cut -c1-150
done
