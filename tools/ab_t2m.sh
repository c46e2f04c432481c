#!/bin/bash
# same-box A/B of library builds on the text->mel micro-benchmarks (run on the GPU box): tools/ab_t2m.sh <rounds> a.so b.so ...
# alternates the in-tree library and the given builds through tools/t2m_bench.py (B=60) and tools/b1_bench.py (B=1, end to end)
N=$1; shift
cp dict_tts_amd/libdicttts_hip.so /tmp/cur.so
for i in $(seq $N); do
  for so in /tmp/cur.so "$@"; do
    cp $so dict_tts_amd/libdicttts_hip.so
    echo "$(basename $so): $(python tools/t2m_bench.py | tail -1)"
    echo "$(basename $so): $(python tools/b1_bench.py 30 | tail -1 | cut -c1-100)"
  done
done
cp /tmp/cur.so dict_tts_amd/libdicttts_hip.so
