#!/bin/bash
# round-3 GPU call 19: the 4-wave / 512-register experiment of the persistent kernel (wave tile 128 x 160, accumulators pinned
# to AGPRs + VGPRs by inline-asm MFMAs, spread fill) against the library's 8-wave kernel: timing, checksums, cycle trace
mkdir -p gpurun_out
for v in w8 w4 w8 w4; do
  echo "== variant $v" >> gpurun_out/r03_big_w4_ab.log
  timeout 60 tools/ubench/big_trace_$v 10 >> gpurun_out/r03_big_w4_ab.log 2>&1
done
echo "== variant w4t" >> gpurun_out/r03_big_w4_trace.log; timeout 60 tools/ubench/big_trace_w4t 10 >> gpurun_out/r03_big_w4_trace.log 2>&1
