#!/bin/bash
# alternate several library builds on the vocoder micro-benchmark: tools/_multi_ab.sh <rounds> a.so b.so ...
N=$1; shift
cp dict_tts_amd/libdicttts_hip.so /tmp/cur.so
for i in $(seq $N); do
  for so in /tmp/cur.so "$@"; do
    cp $so dict_tts_amd/libdicttts_hip.so; echo -n "$(basename $so): "; python tools/voc_bench.py --precision f16 --iters 10 | tail -1 | cut -c1-90
  done
done
cp /tmp/cur.so dict_tts_amd/libdicttts_hip.so
