# per-phase cycle stamps of the two-group ResBlock kernel (rblock2.hip): needs dict_tts_amd/libdicttts_abl.so (make -C dict_tts_amd/csrc ablate); run on the GPU box
cp dict_tts_amd/libdicttts_hip.so /tmp/rel.so
cp dict_tts_amd/libdicttts_abl.so dict_tts_amd/libdicttts_hip.so
DTTS_RB_STATS=1 python tools/voc_bench.py --tune 128 --iters 1 2>&1 | grep "rblock2\|frames" | tail -8
cp /tmp/rel.so dict_tts_amd/libdicttts_hip.so
