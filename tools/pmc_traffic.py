#!/usr/bin/env python3
"""HBM traffic of the HifiGAN convolution family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; rocpd
databases) of the same `bench.py` command.  FETCH_SIZE is doubled as MI355X_MICROARCH.md (HBM section) prescribes for
wide coalesced reads on gfx950 (calibrated here with tools/voc_bench.py DTTS_CALIB=1: a 256 MiB read reports 128 MiB,
a 256 MiB fill reports 256 MiB); both counters are in KB.
usage: pmc_traffic.py fetch.db write.db <vocoder forwards in the profiled run> <valid mel frames over ALL those forwards> [precision
                       [s2pa_fetch.db s2pa_write.db s2pa_probe.json]]
The optional last three come from the same two passes over tools/s2pa_probe.py (both S2PA paths on resident tensors): the counters of the
tensor-API kernel s2pa_kernel<3,.> and of the table kernel s2pa_kernel<1,.> are listed beside the probe's algorithmic bytes."""
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
    # the S2PA dictionary-attention kernels: FETCH_SIZE settles what each really moves (one launch per encode)
    def s2pa_rows(ff, ww, probe=None):
        rows_ = []
        for n in sorted(ff):
            if "s2pa_kernel" not in n:
                continue
            k = ff[n][1] or forwards
            path = "tensor_api" if "s2pa_kernel<3" in n else "resident_table"
            row = {"kernel": n, "path": path, "launches": k, "read_bytes_per_launch": 2.0 * ff[n][0] * 1024.0 / k,
                   "write_bytes_per_launch": ww.get(n, [0.0, None])[0] * 1024.0 / k}
            if probe and path in probe:
                row["algorithmic_bytes_per_launch"] = probe[path]["algorithmic_bytes_per_launch"]
                row["us_per_launch_hipevents"] = probe[path]["us_per_launch"]
                row["fetched_over_algorithmic"] = row["read_bytes_per_launch"] / max(probe[path]["algorithmic_bytes_per_launch"], 1)
            rows_.append(row)
        return rows_
    s2 = {"from_bench_run": s2pa_rows(f, w),
          "note": "FETCH_SIZE x 2 (MI355X_MICROARCH.md HBM section) = L2-miss traffic incl. Infinity-Cache hits; algorithmic bytes: 1,536 B "
                  "(resident table of pre-projected rows) or 6,144 B (tensor API: raw 768-wide key + value) x live gloss rows"}
    if len(sys.argv) > 8:
        with open(sys.argv[8]) as fh:
            probe = json.loads([ln for ln in fh.read().splitlines() if ln.startswith("{")][-1])
        s2["from_s2pa_probe"] = s2pa_rows(per_kernel(sys.argv[6], "FETCH_SIZE"), per_kernel(sys.argv[7], "WRITE_SIZE"), probe)
    print(json.dumps({
        "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on bench.py, MI355X; FETCH_SIZE doubled "
                  "per MI355X_MICROARCH.md HBM section; units KB; produced by tools/pmc_traffic.py",
        "kernels": "dtts::vconv_kernel<*> + dtts::vpair_kernel<*> + dtts::rblock_kernel<*> (the HifiGAN convolution family)",
        "vocoder_precision": prec, "vocoder_forwards_in_profile": forwards, "mel_frames_per_step": frames,
        "hbm_read_bytes_per_step": rd, "hbm_write_bytes_per_step": wr,
        "hbm_bytes_per_mel_frame": (rd + wr) / frames, "s2pa": s2, "per_kernel": rows}, indent=1))


if __name__ == "__main__":
    main()
