#!/bin/bash
# round-3 GPU call 17: timing A/B (no trace instrumentation) of the fill schedules, three interleaved rounds, 20 launches each
mkdir -p gpurun_out
for r in 1 2 3; do for v in 0 6 8 10 2 1; do
  echo "== variant $v" >> gpurun_out/r03_big_sched_ab.log
  timeout 60 tools/ubench/big_trace_n$v 20 >> gpurun_out/r03_big_sched_ab.log 2>&1
done; done
