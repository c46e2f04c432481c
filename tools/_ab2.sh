python -m pytest tests -m gpu -x -q -k "hifigan or memory_safety_vocoder" 2>&1 | tail -3
for i in 1 2; do
  echo -n "old(tune128): "; python tools/voc_bench.py --tune 128 --iters 10 2>/dev/null | tail -1
  echo -n "new(pingpong): "; python tools/voc_bench.py --iters 10 2>/dev/null| tail -1
done
bash tools/prof_voc.sh pp f16 0 > /dev/null
grep "rblock" gpurun_out/pp_voc_trace.md | cut -c1-160
bash tools/_st.sh
