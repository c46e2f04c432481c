#!/bin/bash
# same-box A/B of several builds of the library on the vocoder micro-benchmark (run on the GPU box).  The builds are selected by PATH
# (voc_bench.py --lib): nothing is copied over the in-tree release library.
#   tools/ab_libs.sh <rounds> [trace] a.so b.so ...   ("rel" = the in-tree release library)
# with "trace": one rocprofv3 kernel trace per build as well -> gpurun_out/ab_<name>_voc_trace.md and the fused kernels' rows
N=$1; shift
TR=""; if [ "$1" = "trace" ]; then TR=1; shift; fi
R=$(pwd); export TMPDIR=/tmp
lib_of() { if [ "$1" = "rel" ]; then echo dict_tts_amd/libdicttts_hip.so; else echo $1; fi; }
for i in $(seq $N); do
  for so in "$@"; do
    echo -n "$(basename $so .so): "; python tools/voc_bench.py --lib $(lib_of $so) --precision f16 --iters 10 | tail -2 | tr "\n" " " | cut -c1-150; echo
  done
done
if [ -n "$TR" ]; then
  for so in "$@"; do
    TAG=ab_$(basename $so .so); L=$R/$(lib_of $so)
    rm -rf /tmp/prof_$TAG
    (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o p -- python $R/tools/voc_bench.py --lib $L --precision f16 --iters 5 > /dev/null 2> /tmp/prof_$TAG.err)
    python tools/rocpd_summary.py $(find /tmp/prof_$TAG -name "*.db" | head -1) --detail "kernel<" > gpurun_out/${TAG}_voc_trace.md
    echo "== $TAG"; grep "rblock_kernel\|vpair_kernel\|vconv_kernel" gpurun_out/${TAG}_voc_trace.md | head -${TRN:-16} | cut -c1-150
  done
fi
