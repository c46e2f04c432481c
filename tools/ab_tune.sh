#!/bin/bash
# same-box A/B of DTTS_TUNE settings on the vocoder micro-benchmark (run on the GPU box): tools/ab_tune.sh <rounds> <tuneA> <tuneB> ... [trace]
# alternates the settings through tools/voc_bench.py; with a last argument "trace" also one rocprofv3 kernel trace per setting.
# DTTS_TUNE is honoured only by the ablation build (make -C dict_tts_amd/csrc ablate -> dict_tts_amd/libdicttts_abl.so), which is selected
# by PATH (voc_bench.py --lib): the release library is never overwritten.
set -e
ABL=$(pwd)/dict_tts_amd/libdicttts_abl.so
[ -f $ABL ] || { echo "build the ablation library first: make -C dict_tts_amd/csrc ablate"; exit 1; }
N=$1; shift
TR=""; ARGS=()
for a in "$@"; do if [ "$a" = "trace" ]; then TR=1; else ARGS+=("$a"); fi; done
for i in $(seq $N); do
  for t in "${ARGS[@]}"; do
    echo -n "DTTS_TUNE=$t: "; DTTS_TUNE=$t python tools/voc_bench.py --lib $ABL --precision f16 --iters 10 | tail -1 | cut -c1-100
  done
done
if [ -n "$TR" ]; then
  for t in "${ARGS[@]}"; do
    DTTS_TUNE=$t LIB=$ABL bash tools/prof_voc.sh tune$t f16 > /dev/null
    echo "== DTTS_TUNE=$t"; grep "rblock_kernel\|vpair_kernel\|vconv_kernel" gpurun_out/tune${t}_voc_trace.md | head -14 | cut -c1-150
  done
fi
