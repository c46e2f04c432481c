python -m pytest tests -m gpu -x -q -k "hifigan or config2 or config4 or b1_wave or memory_safety_vocoder or guard" 2>&1 | tail -15 > gpurun_out/r4_t3.log
for i in 1 2 3; do
  echo -n "old(tune128): "; python tools/voc_bench.py --tune 128 --iters 10 | tail -1
  echo -n "new(pingpong): "; python tools/voc_bench.py --iters 10 | tail -1
done > gpurun_out/r4_ab1.txt 2>&1
bash tools/prof_voc.sh pp f16 0 > /dev/null; bash tools/prof_voc.sh old f16 128 > /dev/null
tail -5 gpurun_out/r4_t3.log; cat gpurun_out/r4_ab1.txt
grep "rblock" gpurun_out/pp_voc_trace.md | cut -c1-160; echo; grep "rblock" gpurun_out/old_voc_trace.md | cut -c1-160
