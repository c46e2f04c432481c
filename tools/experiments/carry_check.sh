# md5 of the waveform: builds x shapes (debugging the left-carry chunks)
for shape in "--B 70 --T 64" "--B 60 --T 740"; do
  for cfg in "$@"; do
    echo -n "$shape $cfg: "; python tools/voc_bench.py $shape $cfg --iters 2 2>/dev/null | grep md5
  done
done
