#!/usr/bin/env python3
"""Per-phase clock stamps of every s2pa_kernel workgroup (a variant build with -DS2PA_STAMP: tools/build_variant.sh s2st "-DS2PA_STAMP" ops.hip).
usage: python tools/s2pa_stamps.py build/x/s2st.so [--tensor]   -> when each workgroup started (dispatch ramp) and how long its phases took (100 MHz ticks)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from dict_tts_amd import abi, model, synth

lib = abi.load_library(os.path.abspath(sys.argv[1]))
T = lambda x: torch.from_numpy(np.ascontiguousarray(x))
sd = synth.dict_tts_state_dict(1234)
sd["dur_predictor.linear.0.bias"] = np.array([3.09], np.float32)
m = model.PortaSpeech_dict(hparams={})
m.load_state_dict({k: T(v) for k, v in sd.items()})
st = synth.biaobei_struct()
table = synth.dict_table(1234)
m.upload_dict_table(table)
B = 60
ib = synth.make_id_batch(st["sentences"][:B], table)
d = {k: T(ib[k]).cuda() for k in ("word_tokens", "entry_ids", "pron_modified")}
s = torch.cuda.current_stream().cuda_stream
tensor = "--tensor" in sys.argv   # the tensor API's kernel (collated keys / values resident in HBM) instead of the resident table's
if tensor:
    tb = {k: T(v).cuda() for k, v in synth.make_batch(st["sentences"][:B], 1234).items()}
for _ in range(3):
    if tensor:
        m.ctx.text2mel_encode(tb["word_tokens"].data_ptr(), tb["keys"].data_ptr(), tb["values"].data_ptr(), tb["key_map"].data_ptr(), tb["pinyin"].data_ptr(),
                              tb["pinyin_map"].data_ptr(), tb["pron_modified"].data_ptr(), None, B, tb["word_tokens"].shape[1], tb["keys"].shape[2], tb["pinyin"].shape[2], s)
    else:
        m.ctx.text2mel_encode_ids(d["word_tokens"].data_ptr(), d["entry_ids"].data_ptr(), d["pron_modified"].data_ptr(), None, B, ib["word_tokens"].shape[1], ib["L_k"], ib["P"], s)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (8 * 4096))()
lib.dtts_debug_s2pa_stamps.argtypes = [C.c_void_p]
assert lib.dtts_debug_s2pa_stamps(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.int64)
n = B * ib["word_tokens"].shape[1]
a = a[:n]
a = np.where(a == 0, a[:, :1], a)   # (a workgroup that returned early never wrote its later stamps)
live = (ib["entry_ids"].reshape(-1) >= 0)
t0 = a[:, 0].min()
print(f"workgroups {n} (live words {int(live.sum())}); s_memrealtime ticks are 10 ns")
print("start of workgroup, us after the first: percentiles 0/25/50/75/100:", np.round(np.percentile((a[:, 0] - t0) * 0.01, [0, 25, 50, 75, 100]), 2))
print("end   of workgroup, us after the first start:                     ", np.round(np.percentile((a[:, 6] - t0) * 0.01, [0, 25, 50, 75, 100]), 2))
names = ["prologue (entry, q, first rows in flight, key_map -> LDS, barrier)", "row stream + running sums", "wait at the merge barrier", "softmax over L",
         "context merge + store", "tail (dict_attn row, sense merge, pinyin mix)"]
for sel, lab in ((live, "live words"), (~live, "BOS / padding words")):
    print(f"-- {lab}: mean / p50 / p95 us per phase")
    for i, nm in enumerate(names):
        dlt = (a[sel, i + 1] - a[sel, i]) * 0.01
        print(f"   {nm:70s} {dlt.mean():6.2f} {np.percentile(dlt, 50):6.2f} {np.percentile(dlt, 95):6.2f}")
    tot = (a[sel, 6] - a[sel, 0]) * 0.01
    print(f"   {'whole workgroup':70s} {tot.mean():6.2f} {np.percentile(tot, 50):6.2f} {np.percentile(tot, 95):6.2f}")
# the slowest workgroups: who they are and where their time went
tot = (a[:, 6] - a[:, 0]) * 0.01
order = np.argsort(-(a[:, 6] - t0))[:8]
print("-- the 8 workgroups that finish last (block, start us, end us, phases us):")
for i in order:
    ph = " ".join(f"{(a[i, k + 1] - a[i, k]) * 0.01:5.1f}" for k in range(6))
    print(f"   block {i:5d}  start {(a[i, 0] - t0) * 0.01:6.2f}  end {(a[i, 6] - t0) * 0.01:6.2f}   {ph}")
