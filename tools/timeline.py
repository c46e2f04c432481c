#!/usr/bin/env python3
"""Kernel timeline of the LAST n dispatches of a rocprofv3 result database (rocpd sqlite): start offset, duration and the gap to the
previous kernel's end, in dispatch order — what a launch-bound sequence (text->mel) spends between kernels.
usage: timeline.py results.db [n=200]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
s_col = "start" if "start" in cols else next((x for x in cols if "start" in x.lower()), None)
e_col = "end" if "end" in cols else next((x for x in cols if x.lower().startswith("end")), None)
if not s_col or not e_col:
    print("columns:", cols)
    sys.exit(1)
rows = c.execute(f'select name, "{s_col}", "{e_col}", grid_x, grid_y, grid_z from kernels order by "{s_col}" desc limit {n}').fetchall()[::-1]
t0 = rows[0][1]
prev_end = None
busy = gaps = 0.0
print("| # | start us | dur us | gap us | kernel | grid |")
print("|---|---|---|---|---|---|")
for i, (name, s, e, gx, gy, gz) in enumerate(rows):
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    short = name[name.find("::") + 2:][:70] if "::" in name else name[:70]
    print(f"| {i} | {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {gap:.1f} | `{short}` | {gx}x{gy}x{gz} |")
    busy += (e - s) / 1e3
    if prev_end is not None and gap > 0:
        gaps += gap
    prev_end = max(e, prev_end) if prev_end is not None else e
print(f"\nspan {(prev_end - t0) / 1e3:.1f} us, kernel time {busy:.1f} us, positive gaps {gaps:.1f} us")
