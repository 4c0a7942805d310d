#!/bin/bash
# One parametrised GPU-box runner (replaces the per-call scripts of round 3): `tools/gpu_call.sh <name> <cmd...>` runs
# <cmd> under `timeout` from the repo root and tees its output to gpurun_out/<name>.log.
#   gpurun --timeout 300 -- 'bash tools/gpu_call.sh r04_big_sched 120 tools/ubench/big_sched 5 3'
set -o pipefail
name=$1; limit=$2; shift 2
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "== $(date -u +%FT%TZ) $*" >> gpurun_out/$name.log
timeout "$limit" "$@" 2>&1 | tee -a gpurun_out/$name.log | tail -n 400
echo "rc=$?" >> gpurun_out/$name.log
