#!/bin/bash
# rocprofv3 kernel trace of an arbitrary command -> gpurun_out/<tag>_trace.md (+ stdout in <tag>_out.txt)   (run on the GPU box)
#   bash tools/prof_cmd.sh <tag> [--detail PATTERN] -- <command ...>
TAG=$1; shift
DETAIL="kernel<"
if [ "$1" == "--detail" ]; then DETAIL=$2; shift 2; fi
[ "$1" == "--" ] && shift
R=$(pwd); OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o p -- "$@" > $OUT/${TAG}_out.txt 2> /tmp/prof_$TAG.err
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB --detail "$DETAIL" > $OUT/${TAG}_trace.md 2>> /tmp/prof_$TAG.err
tail -3 /tmp/prof_$TAG.err
