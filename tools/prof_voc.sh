#!/bin/bash
# kernel trace of the vocoder micro-benchmark (tools/voc_bench.py) -> gpurun_out/<tag>_voc_trace.md   (run on the GPU box)
#   LIB=<path> selects another build of the library (voc_bench.py --lib): nothing is ever copied over the in-tree release library
TAG=${1:-x}; PREC=${2:-f16}; TUNE=${3:-0}
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o p -- python $R/tools/voc_bench.py ${LIB:+--lib $LIB} --precision $PREC --tune $TUNE --iters 5 > $OUT/${TAG}_voc_bench.txt 2> /tmp/prof_$TAG.err
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB --detail "kernel<" > $OUT/${TAG}_voc_trace.md 2>> /tmp/prof_$TAG.err
tail -2 $OUT/${TAG}_voc_bench.txt
