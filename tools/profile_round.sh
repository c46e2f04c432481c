#!/bin/bash
# Round-end measurement set, run ON THE GPU BOX from the repo root (gpurun -- 'bash tools/profile_round.sh <tag>').  Everything is
# summarised on the box; only small text files land under gpurun_out/<tag>/ (copy the ones to keep into profiles/):
#   bench.json               default bench.py line (with cpu_baseline)
#   kernel_trace.md          rocprofv3 --kernel-trace --stats of `bench.py --no-cpu-baseline --no-side` (2 streams, as benched)
#   kernel_trace_serial.md   the same with --no-pipeline (one stream: the kernels alone on the GPU)
#   pmc_traffic.json         separate --pmc FETCH_SIZE / WRITE_SIZE passes of the one-stream run -> HBM bytes of the vocoder family
#   pmc_util.md              two --pmc passes (SQ / GRBM / TA counters): MFMA, LDS, TA busy, sustained clock, wave-time split
#   mfma_ceiling.txt         tools/micro/mfma_peak: what the matrix cores sustain with zero vs random operands
set -u
TAG=${1:-final}
R=$(pwd)
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/bench.json 2> $OUT/bench.err
B="python $R/bench.py --no-cpu-baseline --no-side"
cd /tmp
summ() {   # $1 = rocprof dir, $2 = bench JSON of the profiled run, $3 = title
  DB=$(find $1 -name "*.db" | head -1)
  python - "$2" "$3" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r, a = d["roofline"], d["config"]["all_forwards"]
print(f"# rocprofv3 --kernel-trace --stats, {sys.argv[2]}\n")
print(f"MI355X, default workload (4 rotating Biaobei batches of 60, resident dictionary, vocoder {d['config']['vocoder_dtype']}); {a['n']} forwards in the "
      f"trace ({d['warmup']} warm-up + {d['steps']} timed), {a['mel_frames']} valid mel frames.")
print(f"bench.py JSON of the SAME (profiled) run: {d['value']:.0f} mel-frames/s, {d['ms_per_step']:.2f} ms/step; HifiGAN conv family by hipEvent inside the "
      f"timed region: {r['kernel_ms_per_step']:.2f} ms/step over {r['launches']} launches = {r['avg_launch_ms']:.4f} ms per launch = {r['achieved']:.0f} TFLOP/s = "
      f"{100 * r['frac']:.1f} % of the 2.5 PF roof.  The rocprofv3 sum over the same family (all forwards, warm-up included) is printed below the table.\n")
PY
  python $R/tools/rocpd_summary.py $DB --detail "kernel<" --voc-family
}
rm -rf /tmp/pr_*; 
rocprofv3 --kernel-trace --stats -d /tmp/pr_trace -o p -- $B > $OUT/trace_bench.json 2> /tmp/pr_trace.err
summ /tmp/pr_trace $OUT/trace_bench.json "two streams (the benched arrangement): bench.py --no-cpu-baseline --no-side" > $OUT/kernel_trace.md
rocprofv3 --kernel-trace --stats -d /tmp/pr_serial -o p -- $B --no-pipeline > $OUT/trace_serial_bench.json 2> /tmp/pr_serial.err
summ /tmp/pr_serial $OUT/trace_serial_bench.json "ONE stream: bench.py --no-cpu-baseline --no-side --no-pipeline" > $OUT/kernel_trace_serial.md
S="$B --no-pipeline --steps 3 --warmup 1"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pr_fetch -o p -- $S > $OUT/pmc_fetch_bench.json 2> /tmp/pr_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pr_write -o p -- $S > $OUT/pmc_write_bench.json 2> /tmp/pr_write.err
read N F <<< $(python -c "import json; d=json.load(open('$OUT/pmc_fetch_bench.json'))['config']['all_forwards']; print(d['n'], d['mel_frames'])")
# both S2PA paths on resident tensors (tensor API kernel s2pa_kernel<3,.> and the table kernel), same two counters
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pr_s2f -o p -- python $R/tools/s2pa_probe.py > $OUT/s2pa_probe.json 2> /tmp/pr_s2f.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pr_s2w -o p -- python $R/tools/s2pa_probe.py > /dev/null 2> /tmp/pr_s2w.err
python $R/tools/pmc_traffic.py $(find /tmp/pr_fetch -name "*.db" | head -1) $(find /tmp/pr_write -name "*.db" | head -1) $N $F f16 \
    $(find /tmp/pr_s2f -name "*.db" | head -1) $(find /tmp/pr_s2w -name "*.db" | head -1) $OUT/s2pa_probe.json > $OUT/pmc_traffic.json 2> $OUT/pmc_traffic.err
cd $R
bash tools/prof_pmc.sh ${TAG}_u dtts:: -- $S > /dev/null 2>&1
mv $R/gpurun_out/${TAG}_u_pmc_util.md $OUT/pmc_util.md 2>/dev/null
rm -f $R/gpurun_out/${TAG}_u_*
[ -x tools/micro/mfma_peak ] && ./tools/micro/mfma_peak 20000 > $OUT/mfma_ceiling.txt 2>&1
ls -la $OUT
