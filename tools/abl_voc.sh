#!/bin/bash
# phase ablation of the fused vocoder kernels (run on the GPU box): needs dict_tts_amd/libdicttts_abl.so = a build with -DDTTS_ABLATE;
#   tools/abl_voc.sh <DTTS_VCONV_DBG values...>   (rblock bits << 4, vpair bits << 8: 1 contractions, 2 epilogue, 4 x load, 8 rewrites)
cp dict_tts_amd/libdicttts_hip.so /tmp/rel.so
cp dict_tts_amd/libdicttts_abl.so dict_tts_amd/libdicttts_hip.so
for v in $@; do
  DTTS_VCONV_DBG=$v timeout 300 bash tools/prof_voc.sh ab$v f16 > /dev/null
  echo "== dbg $v"; grep "vpair_kernel\|rblock_kernel" gpurun_out/ab${v}_voc_trace.md | head -7 | cut -c12-130
done
cp /tmp/rel.so dict_tts_amd/libdicttts_hip.so
