#!/usr/bin/env python3
"""Utilisation summary of a rocprofv3 --kernel-trace --pmc <counters> pass (rocpd sqlite database): per kernel name the
summed counters and the figures derived from them.

    GRBM_GUI_ACTIVE / 8 XCDs / duration           = shader clock the chip sustained in that kernel
    SQ_VALU_MFMA_BUSY_CYCLES / (256 CUs * 4 SIMDs * GRBM_GUI_ACTIVE / 8)   = MFMA pipe busy fraction (a 32x32x16 16-bit MFMA
                                                      holds the pipe 32 cycles: MI355X_MICROARCH.md, s_memtime row)
    SQ_LDS_IDX_ACTIVE / (256 CUs * GRBM_GUI_ACTIVE / 8)                    = LDS array busy fraction (+ SQ_LDS_BANK_CONFLICT share)
    TA_TA_BUSY / (instances * GRBM_GUI_ACTIVE / 8)                          = texture-addresser (vector memory path) busy fraction
    SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES = where resident waves spend their time
usage: pmc_summary.py results.db [more.db ...] [--match SUBSTR]   (one database per --pmc pass; counters are merged by kernel name)"""
import sqlite3
import sys

N_XCD, N_CU, N_SIMD = 8, 256, 4


def load(db, match):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info('counters_collection')")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    out = {}
    for n, cn, v, k in c.execute(f"select {name_col}, counter_name, sum(value), count(distinct dispatch_id) from counters_collection "
                                 f"group by {name_col}, counter_name"):
        if match and match not in n:
            continue
        out.setdefault(n, {})[cn] = (v, k)
    dur = {}
    for n, k, s in c.execute("select name, count(*), sum(duration) from kernels group by name"):
        dur[n] = (k, s)
    return out, dur


def main():
    args = [a for a in sys.argv[1:]]
    match = None
    if "--match" in args:
        i = args.index("--match")
        match = args[i + 1]
        del args[i:i + 2]
    merged, durs = {}, {}
    for db in args:
        o, d = load(db, match)
        for n, cs in o.items():
            merged.setdefault(n, {}).update({k: v[0] for k, v in cs.items()})
            if "GRBM_GUI_ACTIVE" in cs and n in d:
                durs[n] = d[n]            # durations of the pass that carries the clock counter
            durs.setdefault(n, d.get(n, (0, 0)))
    print("| kernel | calls | total ms | clock GHz | MFMA busy | LDS busy (conflict share) | TA busy | wave time: wait / issue-stall / issuing | MFMA insts (1e6) |")
    print("|---|---|---|---|---|---|---|---|---|")
    for n in sorted(merged, key=lambda x: -durs.get(x, (0, 0))[1]):
        cs = merged[n]
        k, ns = durs.get(n, (0, 0))
        gui = cs.get("GRBM_GUI_ACTIVE", 0.0) / N_XCD          # cycles, per XCD average
        f = lambda x: "n/a" if x is None else f"{100 * x:.1f} %"
        clock = gui / ns if ns else None                       # cycles per ns = GHz
        mfma = cs["SQ_VALU_MFMA_BUSY_CYCLES"] / (N_CU * N_SIMD * gui) if gui and "SQ_VALU_MFMA_BUSY_CYCLES" in cs else None
        lds = cs["SQ_LDS_IDX_ACTIVE"] / (N_CU * gui) if gui and "SQ_LDS_IDX_ACTIVE" in cs else None
        conf = cs["SQ_LDS_BANK_CONFLICT"] / cs["SQ_LDS_IDX_ACTIVE"] if cs.get("SQ_LDS_IDX_ACTIVE") and "SQ_LDS_BANK_CONFLICT" in cs else None
        ta = None
        for key in ("TA_TA_BUSY", "TA_BUSY_"):
            if key in cs and gui:
                ta = cs[key] / (N_CU * gui)                    # one TA per CU
                break
        wc = cs.get("SQ_WAVE_CYCLES")
        waves = "n/a"
        if wc:
            waves = " / ".join(f(cs.get(x, 0) / wc) for x in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"))
        mi = cs.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0) + cs.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0) + cs.get("SQ_INSTS_MFMA", 0) * 0
        print(f"| `{n[:110]}` | {k} | {ns / 1e6:.3f} | {'n/a' if clock is None else f'{clock:.2f}'} | {f(mfma)} | {f(lds)} ({f(conf)}) | {f(ta)} | {waves} | "
              f"{mi / 1e6:.1f} |")
    print("\nraw counter sums per kernel:")
    for n in sorted(merged):
        print(f"- `{n[:110]}`: " + ", ".join(f"{k}={v:.4g}" for k, v in sorted(merged[n].items())))


if __name__ == "__main__":
    main()
