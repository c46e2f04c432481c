#!/bin/bash
# same-box A/B of two builds of the library on the vocoder micro-benchmark (run on the GPU box):
#   tools/ab_voc.sh <other.so> [rounds] [trace]  -> alternates <other.so> and the in-tree library, prints voc_bench's line for each;
#   with a third argument also one rocprofv3 kernel trace per build (gpurun_out/ab_{base,new}_voc_trace.md) and the fused kernels' rows
OTHER=$1; N=${2:-2}
cp dict_tts_amd/libdicttts_hip.so /tmp/new.so
for i in $(seq $N); do
  cp $OTHER dict_tts_amd/libdicttts_hip.so; echo -n "base: "; python tools/voc_bench.py --precision f16 --iters 10 | tail -1
  cp /tmp/new.so dict_tts_amd/libdicttts_hip.so; echo -n "new:  "; python tools/voc_bench.py --precision f16 --iters 10 | tail -1
done
if [ -n "$3" ]; then
  cp $OTHER dict_tts_amd/libdicttts_hip.so; bash tools/prof_voc.sh ab_base f16 > /dev/null
  cp /tmp/new.so dict_tts_amd/libdicttts_hip.so; bash tools/prof_voc.sh ab_new f16 > /dev/null
  for t in base new; do echo "== $t"; grep "rblock_kernel\|vpair_kernel\|vconv_kernel" gpurun_out/ab_${t}_voc_trace.md | head -12 | cut -c1-140; done
fi
