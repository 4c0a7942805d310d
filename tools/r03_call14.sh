#!/bin/bash
# round-3 GPU call 14: fill schedules built on the finding that MFMA issue is oldest-wave-first (mfma_sustain): the OLDER waves
# 0-3 enqueue late (3/4, end, 5/8, 7/8 of their MFMAs), the younger ones right behind the barrier; v9 = r3 roles, late at 1/8
mkdir -p gpurun_out
for v in 0 2 6 7 8 9 10 0; do
  echo "== variant $v" >> gpurun_out/r03_big_trace_variants2.log
  timeout 60 tools/ubench/big_trace_v$v 10 >> gpurun_out/r03_big_trace_variants2.log 2>&1
done
