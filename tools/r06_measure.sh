#!/bin/bash
# Round-6 measurement pass on the GPU box: `gpurun --timeout 2700 -- 'bash tools/r06_measure.sh [part]'` from the repo root.
#   part = bench   : smoke, default bench line, rocprofv3 kernel-trace summary of the bench command
#   part = pmc     : FETCH_SIZE / WRITE_SIZE passes + eager kernel stats at 128 rows, norm bandwidth table, shape profile, SQ counters
#                    of the attention kernels (torch-free harness)
#   part = b256    : end of round 6, the sampler's default max_units 64 -> 128 (256-row phase-1 forwards): the headline-trajectory test
#                    (32 identical images through 256- and 64-row forwards must stay bitwise equal), PMC passes + kernel stats at 256 rows
#                    folded into profiles/pmc_traffic.json["256"], rocprofv3 kernel-trace summary of the bench command, default bench line
#   part = configs : other BASELINE configs, graph-replay times per forward width, 2-rank gloo bench on one GPU
# (the GPU test suite is its own call: tools/gpu_call.sh r06_gpu_suite 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider)
set -x
export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=$PWD
PART=${1:-bench}
O=$R/gpurun_out/r06_final
mkdir -p $O
if [ "$PART" = bench ]; then
  timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
  timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-dtype --no-ref-batch-leg > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err )
fi
if [ "$PART" = pmc ]; then
  ( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python $R/tools/profile_forward.py 128 2 > $O/pmc_fetch.log 2>&1 )
  ( cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python $R/tools/profile_forward.py 128 2 > $O/pmc_write.log 2>&1 )
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_fwd -- python $R/tools/profile_forward.py 128 3 > $O/fwd_stats.log 2>&1 )
  python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write 128 $O/pmc_traffic_b128.json $O/pmc_traffic_table.json > $O/pmc_summary.log 2>&1; cat $O/pmc_summary.log
  python tools/norm_bandwidth.py $O/pmc_traffic_b128.json $(ls $O/rocprof_fwd/*/*kernel_stats.csv | head -1) 3 $O/norm_bandwidth.json > $O/norm_bandwidth.log 2>&1; tail -12 $O/norm_bandwidth.log
  timeout 300 python tools/shape_profile.py 128 > $O/shape_profile_B128.log 2>&1; tail -1 $O/shape_profile_B128.log
  # SQ counters of the attention kernels (north_star: MFMA utilisation on attention), torch-free harness, separate passes
  P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE"
  ( cd /tmp && HARNESS_REPS=2 HARNESS_ATTN8=1 timeout 200 rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d $O/attn_sq -- $R/tools/ubench/attn_harness $R/instancediffusion_amd/libidf_gfx950.so 128 1,4,5 > $O/attn_sq.log 2>&1 )
  ( cd /tmp && HARNESS_REPS=2 HARNESS_ATTN8=1 timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/attn_fetch -- $R/tools/ubench/attn_harness $R/instancediffusion_amd/libidf_gfx950.so 128 1,4,5 > $O/attn_fetch.log 2>&1 )
  ( cd /tmp && HARNESS_REPS=2 HARNESS_ATTN8=1 timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/attn_write -- $R/tools/ubench/attn_harness $R/instancediffusion_amd/libidf_gfx950.so 128 1,4,5 > $O/attn_write.log 2>&1 )
  python tools/sq_counters.py $O/attn_sq_summary.csv $O/attn_sq --match attn > $O/attn_sq_summary.txt 2>&1; cat $O/attn_sq_summary.txt | cut -c1-250
  python tools/pmc_summary.py $O/attn_fetch $O/attn_write 128 $O/attn_pmc_traffic.json > $O/attn_pmc_summary.log 2>&1; grep -i attn $O/attn_pmc_summary.log | cut -c1-200
fi
if [ "$PART" = b256 ]; then
  O=$R/gpurun_out/r06_b256; mkdir -p $O
  timeout 300 python -m pytest tests/test_samplers_gpu.py -q -m gpu -s -k "headline_trajectory_s50_n8_at_bench_width" > $O/headline_256.log 2>&1; grep -h "parity\|passed\|failed" $O/headline_256.log | cut -c1-400
  ( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python $R/tools/profile_forward.py 256 2 > $O/pmc_fetch.log 2>&1 )
  ( cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python $R/tools/profile_forward.py 256 2 > $O/pmc_write.log 2>&1 )
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_fwd -- python $R/tools/profile_forward.py 256 3 > $O/fwd_stats.log 2>&1 )
  python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write 256 $O/pmc_traffic_b256.json $O/pmc_traffic_table.json > $O/pmc_summary.log 2>&1; tail -8 $O/pmc_summary.log
  python - <<PY
import json
tab = json.load(open("$O/pmc_traffic_table.json"))["256"]
for v in tab.values():
    v["source"] = "r06_rocprof/pmc_traffic_b256.json"
p = "$R/profiles/pmc_traffic.json"
d = json.load(open(p)); d["256"] = tab
json.dump(d, open(p, "w"), indent=1)
json.dump(d, open("$O/pmc_traffic_merged.json", "w"), indent=1)
PY
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-dtype --no-ref-batch-leg > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err )
  timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
fi
if [ "$PART" = dtype ]; then
  # fp16 vs bf16, kernel by kernel, on ONE box back to back: eager 128-row forwards under the kernel trace
  for dt in bf16 fp16 bf16 fp16; do
    ( cd /tmp && PROFILE_DTYPE=$dt timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dtype_$dt$RANDOM -- python $R/tools/profile_forward.py 128 3 > $O/dtype_$dt.log 2>&1 )
  done
  B1=$(ls $O/dtype_bf16*/*/*kernel_stats.csv | head -1); F1=$(ls $O/dtype_fp16*/*/*kernel_stats.csv | head -1)
  B2=$(ls $O/dtype_bf16*/*/*kernel_stats.csv | tail -1); F2=$(ls $O/dtype_fp16*/*/*kernel_stats.csv | tail -1)
  python tools/dtype_kernel_diff.py $B1 $F1 3 > $O/dtype_kernel_diff_run1.txt; python tools/dtype_kernel_diff.py $B2 $F2 3 > $O/dtype_kernel_diff_run2.txt
  head -30 $O/dtype_kernel_diff_run1.txt; tail -1 $O/dtype_kernel_diff_run2.txt
  for dt in bf16 fp16 bf16 fp16; do PROFILE_DTYPE=$dt timeout 300 python tools/profile_forward.py 128 20 graph 2>&1 | grep "graph replay" | sed "s/^/$dt /" | tee -a $O/dtype_graph_replay.log; done
fi
if [ "$PART" = configs ]; then
  timeout 600 python tools/run_configs.py c2 c2x8 c4 c5p c5s > $O/configs.log 2>&1; cp gpurun_out/configs.json $O/configs.json; grep -h img_per_s $O/configs.log | cut -c1-200
  for b in 2 4 8 16 18 36 64 72 128; do timeout 300 python tools/profile_forward.py $b 20 graph 2>&1 | grep "graph replay" | tee -a $O/graph_replay_times.log; done
  IDF_BENCH_SINGLE_DEVICE=1 IDF_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 1 --warmup 1 --images-per-gpu 8 --no-alt-dtype > $O/bench_2rank_gloo.log 2>&1; tail -1 $O/bench_2rank_gloo.log | cut -c1-400
fi
find $O -name "*kernel_trace.csv" -size +6M -delete
find $O -name "*.db" -delete
du -sh $O
