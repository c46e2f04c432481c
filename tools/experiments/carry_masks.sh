for m in 0 1 2 4 8 16 32 63; do echo -n "mask $m: "; python tools/voc_bench.py --B 70 --T 64 --lib build/x/km$m.so --iters 2 2>/dev/null | grep md5; done
