#!/bin/bash
# round-3 GPU call 12: torch-free micro-benchmarks -- does an LDS-DMA piece hold its wave (dma_mix), and the per-segment
# cycle trace of the persistent GEMM / conv kernel on the forward's shapes (big_trace)
mkdir -p gpurun_out
timeout 60 tools/ubench/dma_mix > gpurun_out/r03_ubench_dma_mix.log 2>&1; echo "dma_mix rc=$?"
timeout 120 tools/ubench/big_trace 10 > gpurun_out/r03_big_trace.log 2>&1; echo "big_trace rc=$?"
cat gpurun_out/r03_ubench_dma_mix.log; cut -c1-250 gpurun_out/r03_big_trace.log
