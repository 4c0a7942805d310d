#!/bin/bash
# scratch: GPU call script of the moment (see tools/gpu_call.sh for the runner)
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/r04_ab; mkdir -p $O
for b in 2 16; do
  echo "== $b rows, round-3 dispatch (IDF_GEMM_RING=0 IDF_BIG_MIN_EFF=80)"
  IDF_GEMM_RING=0 IDF_BIG_MIN_EFF=80 timeout 300 python tools/profile_forward.py $b 20 graph 2>&1 | grep "graph replay"
  echo "== $b rows, ring 256 only"
  IDF_BIG_MIN_EFF=80 timeout 300 python tools/profile_forward.py $b 20 graph 2>&1 | grep "graph replay"
  echo "== $b rows, bar 50 only"
  IDF_GEMM_RING=0 timeout 300 python tools/profile_forward.py $b 20 graph 2>&1 | grep "graph replay"
  echo "== $b rows, defaults (ring 256, bar 50)"
  timeout 300 python tools/profile_forward.py $b 20 graph 2>&1 | grep "graph replay"
  IDF_GEMM_RING=0 IDF_BIG_MIN_EFF=80 timeout 300 python tools/shape_profile.py $b > $O/shape_B${b}_r3.log 2>&1
  timeout 300 python tools/shape_profile.py $b > $O/shape_B${b}_new.log 2>&1
  tail -1 $O/shape_B${b}_r3.log; tail -1 $O/shape_B${b}_new.log
done
