#!/bin/bash
# SQ counter pass over the torch-free attention harness (d = 40 shapes), one rocprofv3 --pmc pass per counter set.
#   bash tools/attn_pmc_harness.sh <outdir> <batch> <modes> [shape filter]      (on the MI355X box)
OUT=$1; B=${2:-128}; MODES=${3:-1,4}; export HARNESS_SHAPES=${4:-gated 64}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HARNESS_REPS=2
mkdir -p $OUT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pass$i -- tools/ubench/attn_harness instancediffusion_amd/libidf_gfx950.so $B $MODES > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
python tools/sq_counters.py $OUT/summary.csv $OUT/pass1 $OUT/pass2 --match attn4 --useful 0.714 > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
