#!/bin/bash
# round-3 GPU call 2: fused q|k|v after the register fix, attention A/B (interleaved), small-batch kernel-time vs wall study
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -s -p no:cacheprovider -k "fused_qkv or attention or conv_in or gemm_big or self_stats" > gpurun_out/r03_kernel_tests_b.log 2>&1
echo "kernel tests rc=$?"; tail -3 gpurun_out/r03_kernel_tests_b.log
timeout 400 python -m pytest tests/test_engine_gpu.py -q -s -p no:cacheprovider -k "bench_width" > gpurun_out/r03_engine_benchwidth_b.log 2>&1
echo "engine rc=$?"; grep -h "parity\|dispatch\|passed\|failed" gpurun_out/r03_engine_benchwidth_b.log | cut -c1-220
timeout 300 python tools/shape_profile.py 64 > gpurun_out/r03_shape_profile_B64_b.log 2>&1; head -12 gpurun_out/r03_shape_profile_B64_b.log; tail -1 gpurun_out/r03_shape_profile_B64_b.log
timeout 300 python tools/attn_ab.py 64 1,2 > gpurun_out/r03_attn_ab_B64_b.log 2>&1; tail -12 gpurun_out/r03_attn_ab_B64_b.log
for b in 2 18; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_rocprof_fwd_B${b} -- python $R/tools/profile_forward.py $b 20 graph > $R/gpurun_out/r03_fwd_graph_B${b}.log 2>&1 )
  tail -2 gpurun_out/r03_fwd_graph_B${b}.log
done
( cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r03_pmc_attn_write -- python $R/tools/attn_only.py 64 2 > $R/gpurun_out/r03_pmc_attn_write.log 2>&1 )
find gpurun_out -name "*kernel_trace.csv" -size +8M -delete; find gpurun_out -name "*.db" -delete
timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-dtype --no-ref-batch-leg > gpurun_out/r03_bench_b.json 2> gpurun_out/r03_bench_b.err
echo "bench rc=$?"; cut -c1-330 gpurun_out/r03_bench_b.json
