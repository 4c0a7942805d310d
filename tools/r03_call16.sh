#!/bin/bash
# round-3 GPU call 16: VGPR-free residual prefetch (4-B LDS-DMA touches two K-tiles before the epilogue), off / on interleaved
mkdir -p gpurun_out
for v in 0 1 0 1; do
  echo "== variant $v" >> gpurun_out/r03_big_trace_resprefetch.log
  timeout 60 tools/ubench/big_trace_rp$v 10 >> gpurun_out/r03_big_trace_resprefetch.log 2>&1
done
