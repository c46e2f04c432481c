#!/bin/bash
# Which phase of the whole-ResBlock kernels do the LDS bank-conflict cycles belong to?  (conflict share 22-25 % in the k = 3 kernels at
# C <= 64 against 2 % in vpair.)  SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE / SQ_INSTS_LDS per kernel of the vocoder micro-benchmark on the ABLATION
# library, with a phase of rblock_kernel switched off per run (DTTS_VCONV_DBG bits << 4: 16 no contractions, 32 no epilogue, 128 no activation rewrites):
# the counters a phase takes with it when it is skipped are that phase's.  Run on the GPU box: bash tools/lds_conflict_attrib.sh [outdir]
OUT=${1:-gpurun_out/lds_attrib}; R=$(pwd); mkdir -p $OUT; export TMPDIR=/tmp
L=$R/dict_tts_amd/libdicttts_abl.so
[ -f $L ] || { echo "build the ablation library first: make -C dict_tts_amd/csrc ablate"; exit 1; }
for v in 0 16 32 128; do
  rm -rf /tmp/lds_$v
  (cd /tmp && DTTS_VCONV_DBG=$v rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS GRBM_GUI_ACTIVE -d /tmp/lds_$v -o p -- \
      python $R/tools/voc_bench.py --lib $L --precision f16 --iters 2 > /dev/null 2> /tmp/lds_$v.err)
  DB=$(find /tmp/lds_$v -name "*.db" | head -1)
  echo "== DTTS_VCONV_DBG=$v ($([ $v = 0 ] && echo "all phases" || ([ $v = 16 ] && echo "rblock: no contractions") || ([ $v = 32 ] && echo "rblock: no epilogue") || echo "rblock: no activation rewrites"))"
  python $R/tools/pmc_summary.py $DB --match rblock_kernel | grep "^- " | sed 's/void dtts:://; s/(dtts::RBlockParams)//'
done | tee $OUT/lds_conflict_attrib.txt
