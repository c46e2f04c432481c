for i in 1 2; do for l in dict_tts_amd/libdicttts_hip.so build/x/s2s.so; do echo "$l $(python tools/s2pa_probe.py --lib $l 2>&1 | tail -1 | cut -c1-120)"; done; done
bash tools/prof_cmd.sh s2t --detail s2pa -- python tools/s2pa_probe.py > /dev/null 2>&1; grep "s2pa" gpurun_out/s2t_trace.md | head -4 | cut -c1-140
