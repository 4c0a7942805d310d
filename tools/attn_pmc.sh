#!/bin/bash
# SQ / TCC counter passes over the gated self-attention launch (tools/attn_only.py) for one kernel mode.
#   bash tools/attn_pmc.sh <mode> <outdir>        (on the MI355X box; counters in separate passes, kernel-trace only)
set -e
MODE=$1; OUT=$2; B=${3:-64}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p $OUT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU"
P3="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"
P4="SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_BF16 SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_INSTS_SMEM"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pass$i -- python tools/attn_only.py $B 3 $MODE > $OUT/pass$i.log 2>&1 || echo "pass $i failed (see $OUT/pass$i.log)"
done
python tools/sq_counters.py $OUT/summary.csv $OUT/pass1 $OUT/pass2 $OUT/pass3 $OUT/pass4 --match attn --useful 0.714 > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
