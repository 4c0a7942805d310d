#!/bin/bash
# round-3 GPU call 7: hybrid tail split -- tests, 18-row forward A/B (IDF_GEMM_BIG auto with / without the hybrid path is not
# switchable at run time, so compare with the committed graph-replay time of the same day), other configs
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -s -p no:cacheprovider -k "hybrid or split_k or gemm_big or conv3x3_big or fused_qkv or out_stats" > gpurun_out/r03_hybrid_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r03_hybrid_tests.log
timeout 400 python -m pytest tests/test_engine_gpu.py tests/test_samplers_gpu.py -q -s -p no:cacheprovider -k "not bench_width and not s50" > gpurun_out/r03_hybrid_engine_tests.log 2>&1
echo "engine/sampler rc=$?"; tail -2 gpurun_out/r03_hybrid_engine_tests.log
for b in 18 16 8 36; do timeout 300 python tools/profile_forward.py $b 20 graph 2>&1 | grep "graph replay" | tee -a gpurun_out/r03_graph_replay_hybrid.log; done
timeout 300 python tools/shape_profile.py 18 > gpurun_out/r03_shape_profile_B18_hybrid.log 2>&1; head -14 gpurun_out/r03_shape_profile_B18_hybrid.log; tail -1 gpurun_out/r03_shape_profile_B18_hybrid.log
timeout 600 python tools/run_configs.py c2 c5p > gpurun_out/r03_configs_hybrid.log 2>&1; grep -h img_per_s gpurun_out/r03_configs_hybrid.log | cut -c1-160
timeout 300 python bench.py --images-per-gpu 8 --steps 1 --warmup 1 --no-cpu-baseline --no-alt-dtype --no-ref-batch-leg --no-roofline 2>/dev/null | cut -c1-200
