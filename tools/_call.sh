#!/bin/bash
# scratch: GPU call script of the moment (see tools/gpu_call.sh for the runner)
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "== pytest groupnorm"; timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -p no:cacheprovider -k "groupnorm" 2>&1 | tail -8
echo "== GroupNorm two launches vs one pass (graph replay)"; SMALL_SHAPES_GN_ONLY=1 timeout 300 tools/ubench/small_shapes 10 3 128 | grep -E "groupnorm|GroupNorm"
