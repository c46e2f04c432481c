#!/bin/bash
# PMC utilisation passes (separate from any trace domain other than --kernel-trace) of a command -> gpurun_out/<tag>_pmc_util.md
#   bash tools/prof_pmc.sh <tag> <match> -- <command ...>          (run on the GPU box)
TAG=$1; MATCH=$2; shift 2
[ "$1" == "--" ] && shift
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
P2="TA_TA_BUSY GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
DBS=""
i=0
for P in "$P1" "$P2"; do
  i=$((i+1)); rm -rf /tmp/pmc_${TAG}_$i
  rocprofv3 --kernel-trace --pmc $P -d /tmp/pmc_${TAG}_$i -o p -- "$@" > $OUT/${TAG}_pmc${i}_out.txt 2> /tmp/pmc_${TAG}_$i.err
  DB=$(find /tmp/pmc_${TAG}_$i -name "*.db" | head -1)
  [ -z "$DB" ] && { echo "pass $i produced no database"; tail -5 /tmp/pmc_${TAG}_$i.err; continue; }
  DBS="$DBS $DB"
done
python $R/tools/pmc_summary.py $DBS --match "$MATCH" > $OUT/${TAG}_pmc_util.md 2> $OUT/${TAG}_pmc_err.txt
head -30 $OUT/${TAG}_pmc_util.md | cut -c1-260
