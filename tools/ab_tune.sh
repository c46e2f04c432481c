#!/bin/bash
# same-box A/B of DTTS_TUNE settings on the vocoder micro-benchmark (run on the GPU box): tools/ab_tune.sh <rounds> <tuneA> <tuneB> ... [trace]
# alternates the settings through tools/voc_bench.py; with a last argument "trace" also one rocprofv3 kernel trace per setting
# DTTS_TUNE is honoured only by the ablation build (make -C dict_tts_amd/csrc ablate -> dict_tts_amd/libdicttts_abl.so): it is swapped in here
cp dict_tts_amd/libdicttts_hip.so /tmp/rel.so
cp dict_tts_amd/libdicttts_abl.so dict_tts_amd/libdicttts_hip.so || exit 1
trap 'cp /tmp/rel.so dict_tts_amd/libdicttts_hip.so' EXIT
N=$1; shift
TR=""; ARGS=()
for a in "$@"; do if [ "$a" = "trace" ]; then TR=1; else ARGS+=("$a"); fi; done
for i in $(seq $N); do
  for t in "${ARGS[@]}"; do
    echo -n "DTTS_TUNE=$t: "; DTTS_TUNE=$t python tools/voc_bench.py --precision f16 --iters 10 | tail -1 | cut -c1-100
  done
done
if [ -n "$TR" ]; then
  for t in "${ARGS[@]}"; do
    DTTS_TUNE=$t bash tools/prof_voc.sh tune$t f16 > /dev/null
    echo "== DTTS_TUNE=$t"; grep "rblock_kernel\|vpair_kernel\|vconv_kernel" gpurun_out/tune${t}_voc_trace.md | head -14 | cut -c1-150
  done
fi
