#!/usr/bin/env python3
"""HBM traffic of the HifiGAN convolution family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; rocpd
databases) of the same `bench.py` command.  FETCH_SIZE is doubled as MI355X_MICROARCH.md (HBM section) prescribes for
wide coalesced reads on gfx950 (calibrated here with tools/voc_bench.py DTTS_CALIB=1: a 256 MiB read reports 128 MiB,
a 256 MiB fill reports 256 MiB); both counters are in KB.
usage: pmc_traffic.py fetch.db write.db <vocoder forwards in the profiled run> <valid mel frames over ALL those forwards> [precision]"""
import json
import sqlite3
import sys

FAMILY = ("vconv_kernel", "vpair_kernel", "rblock_kernel")


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info('counters_collection')")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    out = {}
    # one row per (dispatch, counter instance): launches = distinct dispatches
    disp_col = "dispatch_id" if "dispatch_id" in cols else None
    for n, v in c.execute(f"select {name_col}, sum(value) from counters_collection where counter_name = ? group by {name_col}", (counter,)):
        out[n] = [v, None]
    if disp_col:
        for n, k in c.execute(f"select {name_col}, count(distinct {disp_col}) from counters_collection where counter_name = ? group by {name_col}", (counter,)):
            out[n][1] = k
    return out


def main():
    fdb, wdb, forwards, frames_all = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    prec = sys.argv[5] if len(sys.argv) > 5 else "f16"
    frames = frames_all / forwards
    f, w = per_kernel(fdb, "FETCH_SIZE"), per_kernel(wdb, "WRITE_SIZE")
    rows, rd, wr = [], 0.0, 0.0
    for n in sorted(f):
        if not any(k in n for k in FAMILY):
            continue
        r = 2.0 * f[n][0] * 1024.0 / forwards
        x = w.get(n, [0.0, None])[0] * 1024.0 / forwards
        rd += r
        wr += x
        rows.append({"kernel": n, "read_bytes_per_step": r, "write_bytes_per_step": x,
                     "launches_per_step": (f[n][1] / forwards) if f[n][1] else None})
    # the S2PA dictionary-attention kernel (its FETCH_SIZE settles what the kernel really moves): one launch per forward
    s2 = None
    for n in sorted(f):
        if "s2pa_kernel" in n:
            k = f[n][1] or forwards
            s2 = {"kernel": n, "launches": k, "read_bytes_per_launch": 2.0 * f[n][0] * 1024.0 / k,
                  "write_bytes_per_launch": w.get(n, [0.0, None])[0] * 1024.0 / k,
                  "note": "FETCH_SIZE x 2 (MI355X_MICROARCH.md HBM section) = L2-miss traffic incl. Infinity-Cache hits; compare with "
                          "stages.s2pa_roofline.bytes_per_launch of bench.py (1,536 B x live gloss rows of the batch)"}
    print(json.dumps({
        "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on bench.py, MI355X; FETCH_SIZE doubled "
                  "per MI355X_MICROARCH.md HBM section; units KB; produced by tools/pmc_traffic.py",
        "kernels": "dtts::vconv_kernel<*> + dtts::vpair_kernel<*> + dtts::rblock_kernel<*> (the HifiGAN convolution family)",
        "vocoder_precision": prec, "vocoder_forwards_in_profile": forwards, "mel_frames_per_step": frames,
        "hbm_read_bytes_per_step": rd, "hbm_write_bytes_per_step": wr,
        "hbm_bytes_per_mel_frame": (rd + wr) / frames, "s2pa": s2, "per_kernel": rows}, indent=1))


if __name__ == "__main__":
    main()
