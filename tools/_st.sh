cp dict_tts_amd/libdicttts_hip.so /tmp/rel.so
cp dict_tts_amd/libdicttts_abl.so dict_tts_amd/libdicttts_hip.so
DTTS_RB_STATS=1 python tools/voc_bench.py --iters 1 2>&1 | grep "rblock2\|frames" | tail -8
cp /tmp/rel.so dict_tts_amd/libdicttts_hip.so
