#!/bin/bash
# same-box A/B in the PIPELINED bench (two streams) of: persistent ResBlock grids leaving DTTS_CU_RESERVE compute units free x tune_flags
# (ablation build, selected by path: bench.py --lib).  tools/ab_reserve.sh <rounds> "<tune>:<reserve>" ...
ABL=$(pwd)/dict_tts_amd/libdicttts_abl.so
[ -f $ABL ] || { echo "build the ablation library first: make -C dict_tts_amd/csrc ablate"; exit 1; }
N=$1; shift
for i in $(seq $N); do
  for tr in "$@"; do
    t=${tr%%:*}; r=${tr##*:}
    echo -n "tune=$t reserve=$r: "
    DTTS_CU_RESERVE=$r python bench.py --lib $ABL --no-cpu-baseline --no-side --voc-tune $t 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print(f\"{d['value']:.0f} frames/s  {d['ms_per_step']:.3f} ms/step  family span {r['kernel_ms_per_step']:.3f}  frac {r['frac']:.4f}\")"
  done
done
