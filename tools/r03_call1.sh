#!/bin/bash
# round-3 GPU call 1: full GPU suite on the pruned / fused library, shape profiles at 64 / 18 / 2 rows, attention A/B, short bench
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1100 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r03_gpu_suite_a.log 2>&1
echo "suite rc=$?" ; tail -5 gpurun_out/r03_gpu_suite_a.log
for b in 64 18 2; do
  timeout 300 python tools/shape_profile.py $b > gpurun_out/r03_shape_profile_B${b}_a.log 2>&1
  tail -1 gpurun_out/r03_shape_profile_B${b}_a.log
done
timeout 200 python tools/attn_ab.py 64 1,2 > gpurun_out/r03_attn_ab_B64_a.log 2>&1; tail -12 gpurun_out/r03_attn_ab_B64_a.log
timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-dtype --no-ref-batch-leg > gpurun_out/r03_bench_a.json 2> gpurun_out/r03_bench_a.err
echo "bench rc=$?"; cut -c1-600 gpurun_out/r03_bench_a.json
