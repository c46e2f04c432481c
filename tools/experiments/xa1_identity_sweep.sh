# the one-set contractions (release) against the double-buffered build (tools/build_variant.sh dbuf "-DRB_XA1=0 -DRB_XA1_32=0 -DVP_XA1=0" rblock.hip vpair.hip):
# waveform md5 over shapes that hit every tile configuration (small-grid variants, ragged / uniform, one long utterance), f16 and bf16x3
for prec in f16 bf16x3 bf16; do
for shape in "--B 1 --T 37" "--B 1 --T 444" "--B 1 --T 1548" "--B 3 --T 200" "--B 8 --T 300" "--B 17 --T 93" "--B 60 --T 740" "--B 70 --T 64 --uniform" "--B 128 --T 400" "--B 2 --T 5012"; do
  a=$(python tools/voc_bench.py $shape --precision $prec --iters 1 2>/dev/null | grep md5 | cut -d" " -f3)
  b=$(python tools/voc_bench.py $shape --precision $prec --iters 1 --lib build/x/dbuf.so 2>/dev/null | grep md5 | cut -d" " -f3)
  [ "$a" = "$b" ] && r=same || r=DIFFERENT
  echo "$prec $shape: $a $b $r"
done; done
