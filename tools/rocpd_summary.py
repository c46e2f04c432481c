#!/usr/bin/env python3
"""Summarise a rocprofv3 result database (rocpd sqlite, the default output of ROCm 7.2's rocprofv3) into the
per-kernel table `rocprofv3 --stats` would print: calls, total / average / min / max duration, share.
usage: rocpd_summary.py results.db [--detail PATTERN] [--voc-family] > profiles/xxx.md
--voc-family adds the sum over the HifiGAN convolution family (vpair<*>, rblock<*> and the vocoder's vconv configurations; the FVAE
decoder's WaveNet layers run on vconv<4,1,1,4,64,true> / <4,1,2,2,64,true> with three channel blocks, grid.y == 3, and are excluded),
the figure bench.py's hipEvent timer must agree with."""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    detail = sys.argv[sys.argv.index("--detail") + 1] if "--detail" in sys.argv else None
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                     "group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for n, k, s, a, mn, mx in rows:
        print(f"| `{n[:150]}` | {k} | {s / 1e6:.3f} | {a / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100 * s / tot:.2f} |")
    print(f"\ntotal kernel time {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    if "--voc-family" in sys.argv:
        ms, n = voc_family(c)
        print(f"\nHifiGAN convolution family in this trace: {ms:.3f} ms over {n} launches = {ms / max(n, 1):.4f} ms per launch")
    if detail:
        print(f"\n### dispatches matching `{detail}` (grid, lds, vgpr, agpr, duration us)\n")
        q = ("select name, grid_x, grid_y, grid_z, lds_size, vgpr_count, accum_vgpr_count, count(*), avg(duration), sum(duration) "
             "from kernels where name like ? group by name, grid_x, grid_y, grid_z, lds_size order by sum(duration) desc limit 250")
        print("| kernel | grid | lds | vgpr | agpr | calls | avg us | total ms |")
        print("|---|---|---|---|---|---|---|---|")
        for n, gx, gy, gz, lds, v, a, k, avg, s in c.execute(q, (f"%{detail}%",)):
            short = n[n.find("<"):n.find(">") + 1] if "<" in n else n[:40]
            print(f"| `{short}` | {gx}x{gy}x{gz} | {lds} | {v} | {a} | {k} | {avg / 1e3:.1f} | {s / 1e6:.3f} |")


def voc_family(c):
    tot, n = 0.0, 0
    for name, gy, k, s in c.execute("select name, grid_y, count(*), sum(duration) from kernels group by name, grid_y"):
        fam = "vpair_kernel" in name or "rblock_kernel" in name
        if "vconv_kernel" in name:
            dec = "<4, 1, 1, 4, 64, true>" in name or ("<4, 1, 2, 2, 64, true>" in name and gy == 3)
            fam = not dec
        if fam:
            tot += s
            n += k
    return tot / 1e6, n


if __name__ == "__main__":
    main()
