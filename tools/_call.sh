#!/bin/bash
# scratch: GPU call script of the moment (see tools/gpu_call.sh for the runner)
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -k "ring or attention_v2_declines" 2>&1 | tail -4
