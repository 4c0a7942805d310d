#!/bin/bash
# round-3 GPU call 22: order of the LDS-DMA pieces of a K-tile: weight rows first (library) vs activation rows first
mkdir -p gpurun_out
for v in d0 af d0 af d0 af; do
  echo "== variant $v" >> gpurun_out/r03_big_piece_order.log
  timeout 60 tools/ubench/big_trace_$v 10 >> gpurun_out/r03_big_piece_order.log 2>&1
done
