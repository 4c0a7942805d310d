#!/bin/bash
# round-3 GPU call 21: conv gather fast path (tap = wave-uniform offset + 9-bit padding mask per row) against the previous kernel
mkdir -p gpurun_out
for v in w8 cv w8 cv w8 cv; do
  echo "== variant $v" >> gpurun_out/r03_big_conv_fastpath.log
  timeout 60 tools/ubench/big_trace_$v 10 >> gpurun_out/r03_big_conv_fastpath.log 2>&1
done
