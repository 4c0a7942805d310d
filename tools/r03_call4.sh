#!/bin/bash
# round-3 GPU call 4: full GPU suite (statistics from the epilogue, 8-wave attention default), shape profile, short bench
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 1100 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r03_gpu_suite_d.log 2>&1
echo "suite rc=$?"; tail -3 gpurun_out/r03_gpu_suite_d.log; grep -h "FAILED\|Error\|out_stats from\|\[bound\]" gpurun_out/r03_gpu_suite_d.log | cut -c1-200 | head -20
timeout 300 python tools/shape_profile.py 64 > gpurun_out/r03_shape_profile_B64_d.log 2>&1; head -8 gpurun_out/r03_shape_profile_B64_d.log; tail -1 gpurun_out/r03_shape_profile_B64_d.log
timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-dtype --no-ref-batch-leg > gpurun_out/r03_bench_d.json 2> gpurun_out/r03_bench_d.err
echo "bench rc=$?"; cut -c1-330 gpurun_out/r03_bench_d.json
