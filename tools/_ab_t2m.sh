cp dict_tts_amd/libdicttts_hip.so /tmp/new.so
for i in 1 2; do for so in dict_tts_amd/libdicttts_base.so /tmp/new.so; do
cp $so dict_tts_amd/libdicttts_hip.so; echo -n "$(basename $so): "; python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages']; print(round(d['value']), round(d['ms_per_step'],2), s['text2mel'])"; python tools/b1_bench.py | tail -1
done; done
cp /tmp/new.so dict_tts_amd/libdicttts_hip.so
