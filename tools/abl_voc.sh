#!/bin/bash
# phase ablation of the fused vocoder kernels (run on the GPU box): needs dict_tts_amd/libdicttts_abl.so = a build with -DDTTS_ABLATE
# (make -C dict_tts_amd/csrc ablate), selected by PATH — the release library is never overwritten;
#   tools/abl_voc.sh <DTTS_VCONV_DBG values...>   (rblock bits << 4, vpair bits << 8: 1 contractions, 2 epilogue, 4 x load, 8 rewrites)
set -e
ABL=$(pwd)/dict_tts_amd/libdicttts_abl.so
[ -f $ABL ] || { echo "build the ablation library first: make -C dict_tts_amd/csrc ablate"; exit 1; }
for v in $@; do
  DTTS_VCONV_DBG=$v LIB=$ABL timeout 300 bash tools/prof_voc.sh ab$v f16 > /dev/null
  echo "== dbg $v"; grep "vpair_kernel\|rblock_kernel" gpurun_out/ab${v}_voc_trace.md | head -7 | cut -c12-130
done
