#!/bin/bash
# round-3 GPU call 6: forward width A/B on one box -- 64-row (32 images) vs 128-row (64 images) forwards
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 500 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-dtype --no-ref-batch-leg --no-roofline > gpurun_out/r03_bench_w64.json 2> gpurun_out/r03_bench_w64.err
echo "64-row: $(cut -c1-250 gpurun_out/r03_bench_w64.json)"
timeout 700 python bench.py --images-per-gpu 64 --max-units 64 --steps 1 --warmup 1 --no-cpu-baseline --no-alt-dtype --no-ref-batch-leg --no-roofline > gpurun_out/r03_bench_w128.json 2> gpurun_out/r03_bench_w128.err
echo "128-row: $(cut -c1-250 gpurun_out/r03_bench_w128.json)"; tail -3 gpurun_out/r03_bench_w128.err
timeout 300 python tools/shape_profile.py 128 > gpurun_out/r03_shape_profile_B128.log 2>&1; tail -1 gpurun_out/r03_shape_profile_B128.log
