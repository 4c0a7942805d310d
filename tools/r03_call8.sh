#!/bin/bash
# round-3 GPU call 8: full GPU suite + smoke + short bench on the final library (hybrid tail split with >= 4 K-tiles per slice)
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r03_gpu_suite_final.log 2>&1
echo "suite rc=$?"; tail -3 gpurun_out/r03_gpu_suite_final.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 python tools/profile_forward.py 18 20 graph 2>&1 | grep "graph replay"
timeout 300 python tools/profile_forward.py 64 20 graph 2>&1 | grep "graph replay"
timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-dtype > gpurun_out/r03_bench_final_short.json 2> gpurun_out/r03_bench_final_short.err
echo "bench rc=$?"; cut -c1-260 gpurun_out/r03_bench_final_short.json
