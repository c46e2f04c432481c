set -x
export TMPDIR=/tmp
ABL=$(pwd)/dict_tts_amd/libdicttts_abl.so
mkdir -p gpurun_out/c1
( for r in 0 4 8 16 32; do echo -n "voc alone reserve=$r: "; DTTS_CU_RESERVE=$r python tools/voc_bench.py --lib $ABL --precision f16 --iters 10 | tail -1 | cut -c1-120; done ) > gpurun_out/c1/voc_reserve.txt 2>&1
bash tools/ab_reserve.sh 2 0:0 0:4 0:8 0:16 4608:0 4608:4 4608:8 4608:16 512:8 > gpurun_out/c1/ab_reserve.txt 2>&1
