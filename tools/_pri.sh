for i in 1 2; do
for p in 0 -1; do echo -n "prio $p: "; python bench.py --no-cpu-baseline --no-side --voc-priority $p | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['kernel_ms_per_step'], r['frac'], r['isolated']['kernel_ms_per_forward'] if 'isolated' in r else '')"; done
echo -n "tune 4096: "; python bench.py --no-cpu-baseline --no-side --voc-tune 4096 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['kernel_ms_per_step'], r['frac'])"
done
