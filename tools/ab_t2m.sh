#!/bin/bash
# same-box A/B of library builds on the text->mel micro-benchmarks (run on the GPU box): tools/ab_t2m.sh <rounds> a.so b.so ...  ("rel" = the
# in-tree release library).  Alternates the builds through tools/t2m_bench.py (B=60) and tools/b1_bench.py (B=1, end to end); the builds are
# selected by PATH (--lib): nothing is copied over the release library.
N=$1; shift
lib_of() { if [ "$1" = "rel" ]; then echo dict_tts_amd/libdicttts_hip.so; else echo $1; fi; }
for i in $(seq $N); do
  for so in "$@"; do
    echo "$(basename $so .so): $(python tools/t2m_bench.py --lib $(lib_of $so) | tail -1)"
    echo "$(basename $so .so): $(python tools/b1_bench.py 30 --lib $(lib_of $so) | tail -1 | cut -c1-100)"
  done
done
