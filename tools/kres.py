#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage remarks (tools/build_variant.sh writes them to build/x/<name>_<src>.log):
kernel, VGPRs, AGPRs, spilled VGPRs, scratch bytes, waves / SIMD.   usage: kres.py build/x/base_vpair.log [filter]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = []
for blk in txt.split("Function Name: ")[1:]:
    name = blk.split()[0]
    g = lambda k: (re.search(k + r": (\d+)", blk) or [0, "?"])[1]
    rows.append((name, g("VGPRs"), g("AGPRs"), g("VGPRs Spill"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]")))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
for r, n in zip(rows, names):
    n = n.replace("dtts::", "").split("(")[0]
    if flt in n:
        print(f"{n:60s} vgpr {r[1]:>3} agpr {r[2]:>3} spill {r[3]:>3} scratch {r[4]:>4} occ {r[5]}")
