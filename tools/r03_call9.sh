#!/bin/bash
# round-3 GPU call 9: the two tests the hybrid split broke (now opt-in), the hybrid test itself, the kernel file
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_vae_gpu.py tests/test_engine_gpu.py -q -s -p no:cacheprovider > gpurun_out/r03_gpu_suite_final2.log 2>&1
echo "rc=$?"; tail -3 gpurun_out/r03_gpu_suite_final2.log; grep -h "FAILED" gpurun_out/r03_gpu_suite_final2.log | head
