#!/bin/bash
# round-3 GPU call 11: default bench line + graph-replay times on the final library
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python bench.py > gpurun_out/r03_bench_final2.json 2> gpurun_out/r03_bench_final2.err
echo "bench rc=$?"; cut -c1-260 gpurun_out/r03_bench_final2.json
for b in 18 64 128; do timeout 300 python tools/profile_forward.py $b 20 graph 2>&1 | grep "graph replay" | tee -a gpurun_out/r03_graph_replay_final2.log; done
