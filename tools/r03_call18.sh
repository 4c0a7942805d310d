#!/bin/bash
# round-3 GPU call 18: the new fill schedule as the library default -- harness checksums (new default vs round-2 schedule vs
# the 7/8 variant), the GEMM / conv kernel tests, and a same-box A/B of the graph-replayed 64-row forward against the
# round-2-schedule library (IDF_LIB_PATH)
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for v in n0 new n10; do echo "== variant $v" >> gpurun_out/r03_big_sched_default.log; timeout 60 tools/ubench/big_trace_$v 10 >> gpurun_out/r03_big_sched_default.log 2>&1; done
timeout 280 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or conv3 or qkv or geglu or stats or hybrid or big" > gpurun_out/r03_kernel_tests_new_schedule.log 2>&1
echo "pytest rc=$?"; tail -2 gpurun_out/r03_kernel_tests_new_schedule.log
for lib in libidf_gfx950.so libidf_gfx950_r2sched.so; do
  echo "== $lib" | tee -a gpurun_out/r03_forward_ab_schedule.log
  IDF_LIB_PATH=$PWD/instancediffusion_amd/$lib timeout 200 python tools/profile_forward.py 64 30 graph 2>&1 | grep "graph replay" | tee -a gpurun_out/r03_forward_ab_schedule.log
done
