#!/bin/bash
# Package power and shader clock while a command runs (rocm-smi sampled every 0.25 s): is the vocoder at the board's power limit?
#   bash tools/power_sample.sh <outfile> -- <command ...>        (run on the GPU box)
OUT=$1; shift; [ "$1" == "--" ] && shift
"$@" > /dev/null 2>&1 &
PID=$!
sleep ${WARM:-6}
: > $OUT
while kill -0 $PID 2>/dev/null; do
  rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -i "power\|sclk" | tr -s " " | tr "\n" ";" >> $OUT; echo >> $OUT
  sleep 0.25
done
python3 - $OUT <<'PY'
import re, sys
pw, ck, mx = [], [], None
for ln in open(sys.argv[1]):
    m = re.search(r"(?:Average|Current Socket) Graphics Package Power \(W\): ([\d.]+)", ln) or re.search(r"Package Power \(W\): ([\d.]+)", ln)
    if m: pw.append(float(m.group(1)))
    m = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", ln)
    if m: ck.append(int(m.group(1)))
    m = re.search(r"Max Graphics Package Power \(W\): ([\d.]+)", ln)
    if m: mx = float(m.group(1))
if pw:
    pw.sort(); print(f"samples {len(pw)}  power W: median {pw[len(pw)//2]:.0f}  p90 {pw[int(len(pw)*0.9)]:.0f}  max {pw[-1]:.0f}   cap {mx}")
if ck:
    ck.sort(); print(f"sclk MHz: median {ck[len(ck)//2]}  min {ck[0]}  max {ck[-1]}")
PY
