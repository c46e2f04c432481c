#!/bin/bash
# Round-end measurement set, run ON THE GPU BOX from the repo root (gpurun -- 'bash tools/profile_round.sh <tag>'):
#   bench.json          default bench.py line (with cpu_baseline)
#   trace/              rocprofv3 --kernel-trace --stats of the same command (no cpu_baseline leg)
#   pmc_fetch/, pmc_write/   separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of a short bench run
# Everything lands under gpurun_out/<tag>/ ; tools/rocpd_summary.py and tools/pmc_traffic.py turn it into profiles/.
set -u
TAG=${1:-final}
R=$(pwd)
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o p -- python $R/bench.py --no-cpu-baseline > $OUT/trace_bench.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- python $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- python $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $OUT/pmc_write.json 2> $OUT/pmc_write.err
# the databases are large: keep only what the summaries need
for d in trace pmc_fetch pmc_write; do ls -la $OUT/$d | tail -3; done
