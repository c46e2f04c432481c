#!/usr/bin/env python3
"""HifiGAN family per stage from a one-stream kernel trace summary (tools/rocpd_summary.py --detail "kernel<" output, e.g.
gpurun_out/<tag>/kernel_trace_serial.md + its bench JSON): ms per forward, TF/s, fraction of the 2.5 PF dense roof and of the 1.67 PF the matrix cores
sustain with fp16 data.  The range-guard forwards (GUARD instantiations: template argument `true` in the guard slot) are left out.
usage: per_stage.py kernel_trace_serial.md trace_serial_bench.json [prev: "3.18,6.62,3.87,2.58,1.83"]"""
import json
import re
import sys

md, bj = sys.argv[1], json.load(open(sys.argv[2]))
prev = [float(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else None
n_fw = bj["config"]["all_forwards"]["n"]
frames = bj["config"]["all_forwards"]["mel_frames"] / n_fw
guard_fw = (bj.get("vocoder_modes", {}).get("f16", {}).get("range_guard") or {}).get("guarded_forwards", 4)
FLOPS = {1: 132.1e6, 2: 264.2e6, 3: 132.1e6, 4: 66.1e6, 0: 19.6e6}       # per mel frame (ResBlocks of stage 1..4; 0 = conv_pre + 4 upsamplers)
agg, detail, in_detail = {}, [], False
for ln in open(md):
    if ln.startswith("### dispatches matching"):
        in_detail = True
        continue
    m = re.match(r"\| `(.+?)` \| (.+) \|$", ln.strip())
    if not m:
        continue
    cols = [c.strip() for c in m.group(2).split("|")]
    if in_detail:
        detail.append((m.group(1), cols[0], int(cols[4]), float(cols[6])))          # template args, grid, calls, total ms
    else:
        agg[m.group(1)] = (int(cols[0]), float(cols[1]))                             # calls, total ms
stage_ms = {0: 0.0, 1: 0.0, 2: 0.0, 3: 0.0, 4: 0.0}
unguarded = n_fw - guard_fw
for name, (calls, ms) in agg.items():
    m = re.match(r"void dtts::(vpair|rblock)_kernel<(\d+), ([^>]*)>", name)
    if not m:
        continue
    args = [a.strip() for a in m.group(3).split(",")]
    guard = args[2] == "true" if m.group(1) == "vpair" else args[6] == "true"
    if guard:
        continue
    stage_ms[{256: 1, 128: 2, 64: 3, 32: 4}[int(m.group(2))]] += ms / unguarded
# serial convolutions: the vocoder's vconv launches are the ones with >= 1,000 workgroups per utterance-row of the grid (conv_pre: x2 co-groups)
for targs, grid, calls, ms in detail:
    if not re.match(r"<\d, \d, \d, \d, \d+, true, false>", targs):
        continue
    gx, gy, gz = (int(v) for v in grid.split("x"))
    voc = (targs.startswith("<4, 2, 1, 4, 128") or targs.startswith("<2, 2, 1, 4, 128") or
           (targs.startswith("<2, 1, 1, 4, 128") and gx > 50000) or (targs.startswith("<2, 1, 2, 2, 64") and gx > 50000))
    if voc:
        stage_ms[0] += ms / n_fw
rows = [(2, "stage 2, C = 128 (`vpair<128,256|192>`, `rblock<128>`)"), (3, "stage 3, C = 64 (`rblock<64,5|4,...>` x 3)"),
        (1, "stage 1, C = 256 (`vpair<256,96|128>`, `rblock<256>`)"), (4, "stage 4, C = 32 (`rblock<32,...>` x 3 + fused conv_post)"),
        (0, "conv_pre + 4 upsamplers (`vconv<...,X3>`)")]
pidx = {1: 0, 2: 1, 3: 2, 4: 3, 0: 4}
print(f"| stage (kernels) | ms / forward{' (previous -> now)' if prev else ''} | TF/s | of 2.5 PF | of the 1.67 PF data ceiling |\n|---|---|---|---|---|")
tot = 0.0
for st, label in rows:
    ms = stage_ms[st]
    tot += ms
    tf = FLOPS[st] * frames / (ms * 1e-3) / 1e12
    p = f"{prev[pidx[st]]:.2f} -> " if prev else ""
    print(f"| {label} | {p}**{ms:.2f}** | {tf:.0f} | {100 * tf / 2500:.1f} % | {100 * tf / 1670:.0f} % |")
tf = 614.105088e6 * frames / (tot * 1e-3) / 1e12
print(f"| **sum of the rows above** | {f'{sum(prev):.2f} -> ' if prev else ''}**{tot:.2f}** | {tf:.0f} | {100 * tf / 2500:.1f} % | {100 * tf / 1670:.0f} % |")
print(f"\n({unguarded} unguarded forwards of {frames:.0f} valid mel frames on average; one stream)")
