#!/bin/bash
# scratch: GPU call script of the moment (see tools/gpu_call.sh for the runner)
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "== pytest gemm/conv kernels"; timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -p no:cacheprovider -k "gemm or conv or big or ring" 2>&1 | tail -4
echo "== 2-row kernel trace"
O=$PWD/gpurun_out/r04_b2_trace2; mkdir -p $O; R=$PWD
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/graph -- python $R/tools/profile_forward.py 2 20 graph 2>&1 | grep -E "graph replay" )
python3 - "$(ls $O/graph/*/*kernel_stats.csv | head -1)" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
print(f"kernel-trace: {calls} kernel launches, sum of durations {tot/1e6:.2f} ms")
for r in sorted(rows, key=lambda r: -int(r["TotalDurationNs"]))[:16]:
    print(f'{int(r["Calls"]):7d} x {float(r["AverageNs"])/1e3:8.2f} us  {float(r["Percentage"]):5.1f} %  {r["Name"][:120]}')
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
