#!/bin/bash
# round-3 GPU call 10: period-32 GEGLU (tests, in-model A/B on one box), bench with the broadcast negative context
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 400 python -m pytest tests/test_kernels_gpu.py -q -s -p no:cacheprovider -k "geglu or hybrid" > gpurun_out/r03_geglu_tests.log 2>&1
echo "tests rc=$?"; grep -h "GEGLU period\|passed\|failed" gpurun_out/r03_geglu_tests.log | cut -c1-200
for pd in 64 32; do
  IDF_GEGLU_PERIOD=$pd timeout 300 python tools/shape_profile.py 64 > gpurun_out/r03_shape_profile_B64_geglu$pd.log 2>&1
  echo "period $pd: $(grep geglu gpurun_out/r03_shape_profile_B64_geglu$pd.log | tr '\n' ' ' | cut -c1-420) $(tail -1 gpurun_out/r03_shape_profile_B64_geglu$pd.log)"
done
timeout 400 python -m pytest tests/test_engine_gpu.py -q -s -p no:cacheprovider -k "bench_width and not c4" > gpurun_out/r03_engine_geglu32.log 2>&1
echo "engine rc=$?"; grep -h "parity\|passed\|failed" gpurun_out/r03_engine_geglu32.log | cut -c1-200
timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-dtype --no-ref-batch-leg > gpurun_out/r03_bench_geglu32.json 2> gpurun_out/r03_bench_geglu32.err
echo "bench rc=$?"; cut -c1-250 gpurun_out/r03_bench_geglu32.json
IDF_GEGLU_PERIOD=64 timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-dtype --no-ref-batch-leg --no-roofline > gpurun_out/r03_bench_geglu64.json 2> gpurun_out/r03_bench_geglu64.err
echo "bench64 rc=$?"; cut -c1-250 gpurun_out/r03_bench_geglu64.json
