#!/bin/bash
# round-3 GPU call 13: K-loop fill-schedule variants of the persistent kernel (tools/ubench/big_trace.hip built with
# -DIDF_LATE_LOW / -DIDF_LATE_NUM,DEN / -DIDF_LATE_PRIO), same box, back to back
mkdir -p gpurun_out
for v in 0 1 2 3 4 5 0; do
  echo "== variant $v" >> gpurun_out/r03_big_trace_variants.log
  timeout 60 tools/ubench/big_trace_v$v 10 >> gpurun_out/r03_big_trace_variants.log 2>&1
done
grep -v "wgM\|wave4" gpurun_out/r03_big_trace_variants.log | cut -c1-200
