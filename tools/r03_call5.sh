#!/bin/bash
# round-3 GPU call 5: attention after the spill fix (tests, A/B, WRITE_SIZE), in-model A/B of the attention geometry and of the
# LayerNorm-statistics source on ONE box
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -s -p no:cacheprovider -k "attention_v2 or attention_v4 or out_stats" > gpurun_out/r03_attn_tests_e.log 2>&1
echo "tests rc=$?"; tail -2 gpurun_out/r03_attn_tests_e.log
timeout 300 python tools/attn_ab.py 64 1,2 > gpurun_out/r03_attn_ab_B64_e.log 2>&1; grep "64^2\|96^2" gpurun_out/r03_attn_ab_B64_e.log
( cd /tmp && timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r03_pmc_attn_write_e -- python $R/tools/attn_only.py 64 2 > $R/gpurun_out/r03_pmc_attn_write_e.log 2>&1 )
( cd /tmp && timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/r03_pmc_attn_fetch_e -- python $R/tools/attn_only.py 64 2 > $R/gpurun_out/r03_pmc_attn_fetch_e.log 2>&1 )
grep -h "attn4" gpurun_out/r03_pmc_attn_write_e/*/*counter_collection.csv | awk -F, '{print $(NF-3), $(NF-2)}' | head -4
grep -h "attn4" gpurun_out/r03_pmc_attn_fetch_e/*/*counter_collection.csv | awk -F, '{print $(NF-3), $(NF-2)}' | head -4
for m in 1 2; do
  IDF_ATTN2=$m timeout 300 python tools/shape_profile.py 64 > gpurun_out/r03_shape_profile_B64_attn$m.log 2>&1
  echo "IDF_ATTN2=$m: $(grep 'attn Nq4096 C320 n0=4096' gpurun_out/r03_shape_profile_B64_attn$m.log | tr '\n' ' ') $(tail -1 gpurun_out/r03_shape_profile_B64_attn$m.log)"
done
for m in 0 2; do
  IDF_LN_SELF=$m timeout 300 python tools/shape_profile.py 64 > gpurun_out/r03_shape_profile_B64_lnself$m.log 2>&1
  echo "IDF_LN_SELF=$m: $(tail -1 gpurun_out/r03_shape_profile_B64_lnself$m.log)"
done
find gpurun_out -name "*kernel_trace.csv" -size +8M -delete; find gpurun_out -name "*.db" -delete
