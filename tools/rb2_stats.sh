#!/bin/bash
# per-phase cycle sums of the two-group ResBlock kernel (rblock2.hip: ablation library only, tune bit 7; LABNOTES round 4 (C1)), run on the GPU box.
# The ablation build is selected by PATH (voc_bench.py --lib): the release library is never overwritten.
set -e
ABL=$(pwd)/dict_tts_amd/libdicttts_abl.so
[ -f $ABL ] || { echo "build the ablation library first: make -C dict_tts_amd/csrc ablate"; exit 1; }
DTTS_RB_STATS=1 python tools/voc_bench.py --lib $ABL --tune 128 --iters 1 2>&1 | grep "rblock2\|frames" | tail -8
