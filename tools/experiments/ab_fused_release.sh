# pipelined bench, release library: default launches vs the stage-fused forms (tune bits 9 / 12), after the one-set contractions (LABNOTES (Y))
for i in 1 2; do for t in 0 512 4096 4608; do echo -n "tune=$t: "; python bench.py --no-cpu-baseline --no-side --voc-tune $t 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print(f\"{d['value']:.0f} frames/s  {d['ms_per_step']:.3f} ms/step  family span {r['kernel_ms_per_step']:.3f}  frac {r['frac']:.4f}\")"; done; done
