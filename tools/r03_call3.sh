#!/bin/bash
# round-3 GPU call 3: 8-wave attention A/B + its tests, un-profiled graph-replay times at 2 / 18 / 64 rows, C2
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -s -p no:cacheprovider -k "attention_v2 or attention_v4" > gpurun_out/r03_attn_tests_c.log 2>&1
echo "attn tests rc=$?"; tail -2 gpurun_out/r03_attn_tests_c.log
timeout 300 python tools/attn_ab.py 64 1,3 > gpurun_out/r03_attn_ab_B64_c.log 2>&1; tail -12 gpurun_out/r03_attn_ab_B64_c.log
ATTN_AB_DTYPE=fp16 timeout 300 python tools/attn_ab.py 64 1,3 > gpurun_out/r03_attn_ab_B64_c_fp16.log 2>&1; grep "64^2" gpurun_out/r03_attn_ab_B64_c_fp16.log
for b in 2 18 64; do
  timeout 300 python tools/profile_forward.py $b 20 graph 2>&1 | grep "graph replay" | tee -a gpurun_out/r03_graph_replay_times.log
done
timeout 300 python tools/run_configs.py c2 > gpurun_out/r03_configs_c2.log 2>&1; tail -3 gpurun_out/r03_configs_c2.log; cp gpurun_out/configs.json gpurun_out/r03_configs_c2.json
