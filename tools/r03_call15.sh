#!/bin/bash
# round-3 GPU call 15: residual prefetch depth in the persistent kernel's epilogue (IDF_RES_PF = 0 (load at use), 2, 3, 4, 6)
mkdir -p gpurun_out
for v in 0 2 3 4 6 0; do
  echo "== variant $v" >> gpurun_out/r03_big_trace_respf.log
  timeout 60 tools/ubench/big_trace_pf$v 10 >> gpurun_out/r03_big_trace_respf.log 2>&1
done
