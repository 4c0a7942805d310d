#!/bin/bash
# round-3 GPU call 20: 4-wave experiment, where the wave enqueues its 18 pieces: 1 / 2 / 3 / 6 per position from the start of
# the K-tile, or all 18 before the first MFMA
mkdir -p gpurun_out
for v in w8 w4 w4p2 w4p3 w4p6 w4p18; do
  echo "== variant $v" >> gpurun_out/r03_big_w4_fill.log
  timeout 60 tools/ubench/big_trace_$v 10 >> gpurun_out/r03_big_w4_fill.log 2>&1
done
