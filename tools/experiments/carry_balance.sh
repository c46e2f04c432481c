# left-carry chunks on the 640-row tile (one-set contraction), chunk shortening 0 / 25 / 40 %, against the plain kernels (tune bit 15) — vocoder ms / forward and waveform md5
for i in 1 2 3; do
  echo -n "plain:   "; python tools/voc_bench.py --lib build/x/carry0.so --tune 32768 --iters 10 2>/dev/null | tr "\n" " " | cut -c1-150; echo
  for v in 0 25 40; do echo -n "carry$v: "; python tools/voc_bench.py --lib build/x/carry$v.so --iters 10 2>/dev/null | tr "\n" " " | cut -c1-150; echo; done
done
for v in 0 25 40; do echo -n "70x64 carry$v: "; python tools/voc_bench.py --B 70 --T 64 --lib build/x/carry$v.so --iters 2 2>/dev/null | grep md5; done
