#!/bin/bash
# Round-2 measurement pass on the GPU box (run through gpurun from the repo root): kernel-trace stats of the default bench
# command, the two PMC traffic passes over eager forwards at batch 64, the non-headline configs, the plain bench line.
set -x
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r02_final
mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_bench -- python $R/bench.py --no-cpu-baseline --no-alt-dtype > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err )
( cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python $R/tools/profile_forward.py 64 2 > $O/pmc_fetch.log 2>&1 )
( cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python $R/tools/profile_forward.py 64 2 > $O/pmc_write.log 2>&1 )
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/rocprof_fwd -- python $R/tools/profile_forward.py 64 3 > $O/fwd_stats.log 2>&1 )
python tools/run_configs.py c2 c4 c5p c5s > $O/configs.log 2>&1; cp gpurun_out/configs.json $O/configs.json
# keep only the summaries (the raw traces are hundreds of MB)
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*.db" -delete
du -sh $O
ls -R $O | head -60
