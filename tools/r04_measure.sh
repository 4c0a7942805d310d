#!/bin/bash
# Round-4 measurement pass on the GPU box: `gpurun --timeout 2400 -- 'bash tools/r04_measure.sh [final]'` from the repo root.
# Most important first (a call that runs out of time still leaves the suite, the bench line and the profiles behind).
# Without an argument: the whole pass incl. the two --pmc passes and the eager 128-row kernel trace (first pass of the round,
# outputs under gpurun_out/r04_final/).  `final`: the re-run on the round's last code -- suite, smoke, bench, kernel trace of the
# bench command, other configs, graph-replay times, 2-rank run (gpurun_out/r04_final2/); the PMC / eager-trace artefacts of
# the first pass stay valid (nothing on the 128-row path changed behind them except the dispatch of eight 8x8-level GEMMs).
set -x
export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=$PWD
MODE=${1:-full}
O=$R/gpurun_out/r04_final; [ "$MODE" = final ] && O=$R/gpurun_out/r04_final2
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/gpu_suite.log 2>&1; tail -2 $O/gpu_suite.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt-dtype --no-ref-batch-leg > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err )
if [ "$MODE" != final ]; then
  ( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python $R/tools/profile_forward.py 128 2 > $O/pmc_fetch.log 2>&1 )
  ( cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python $R/tools/profile_forward.py 128 2 > $O/pmc_write.log 2>&1 )
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_fwd -- python $R/tools/profile_forward.py 128 3 > $O/fwd_stats.log 2>&1 )
  python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write 128 $O/pmc_traffic_b128.json $O/pmc_traffic_table.json > $O/pmc_summary.log 2>&1; cat $O/pmc_summary.log
  python tools/norm_bandwidth.py $O/pmc_traffic_b128.json $(ls $O/rocprof_fwd/*/*kernel_stats.csv | head -1) 3 $O/norm_bandwidth.json > $O/norm_bandwidth.log 2>&1; tail -12 $O/norm_bandwidth.log
  timeout 300 python tools/shape_profile.py 128 > $O/shape_profile_B128.log 2>&1; tail -1 $O/shape_profile_B128.log
  timeout 60 tools/ubench/mlp_harness 5 > $O/mlp_harness.log 2>&1; grep "M 524288" $O/mlp_harness.log
fi
timeout 600 python tools/run_configs.py c2 c4 c5p c5s > $O/configs.log 2>&1; cp gpurun_out/configs.json $O/configs.json; grep -h img_per_s $O/configs.log | cut -c1-200
for b in 2 16 64 128; do timeout 300 python tools/profile_forward.py $b 20 graph 2>&1 | grep "graph replay" | tee -a $O/graph_replay_times.log; done
IDF_BENCH_SINGLE_DEVICE=1 IDF_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 1 --warmup 1 --images-per-gpu 8 --no-alt-dtype > $O/bench_2rank_gloo.log 2>&1; tail -1 $O/bench_2rank_gloo.log | cut -c1-400
[ "$MODE" != final ] && { timeout 200 python tools/vae_bench.py 4 5 > $O/vae_bench.log 2>&1; tail -2 $O/vae_bench.log; }
find $O -name "*kernel_trace.csv" -size +6M -delete
find $O -name "*.db" -delete
du -sh $O
