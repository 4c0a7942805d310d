#!/bin/bash
# scratch: GPU call script of the moment (see tools/gpu_call.sh for the runner)
export TMPDIR=/tmp PYTHONUNBUFFERED=1
for w in default 1; do
  echo "== IDF_TILE_WIDE=$w"
  if [ $w = default ]; then timeout 300 tools/ubench/small_shapes 20 3 16 > /tmp/o.txt; else IDF_TILE_WIDE=$w timeout 300 tools/ubench/small_shapes 20 3 16 > /tmp/o.txt; fi
  grep -E "N320 |forward-weighted GEMM" /tmp/o.txt | cut -c1-175
done
