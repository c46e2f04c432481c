#!/usr/bin/env python3
"""Both S2PA paths on tensors RESIDENT in HBM, for rocprofv3 --pmc passes (tools/profile_round.sh): the first 60 Biaobei sentences,
`--n` encodes through the tensor API (collated keys / values [B,T_w,L_k,768] -> s2pa_kernel<3,.>) and through the resident table of
pre-projected rows (entry ids -> s2pa_kernel<1,.>).  Prints one JSON line: hipEvent time per launch and the algorithmic bytes of each."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from dict_tts_amd import abi, model, synth

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=4)
ap.add_argument("--lib", default=None, help="path of the library build to load instead of the in-tree release library (A/B runs)")
a = ap.parse_args()
if a.lib:
    abi.load_library(os.path.abspath(a.lib))
T = lambda x: torch.from_numpy(np.ascontiguousarray(x))
dev = torch.device("cuda", 0)
m = model.PortaSpeech_dict(hparams={})
m.load_state_dict({k: T(v) for k, v in synth.dict_tts_state_dict(1234).items()})
st = synth.biaobei_struct()
sent = st["sentences"][:60]
table = synth.dict_table(1234)
m.upload_dict_table(table)
ib = synth.make_id_batch(sent, table)
tb = {k: T(v).to(dev) for k, v in synth.make_batch(sent, 1234).items()}
d = {k: T(ib[k]).to(dev) for k in ("word_tokens", "entry_ids", "pron_modified")}
live_rows_of_entry = np.add.reduceat((table["key_map"] != 0).astype(np.int64), table["tok_off"][:-1])
e = ib["entry_ids"]
live_table = int(live_rows_of_entry[e[e >= 0]].sum())
live_tensor = int((tb["key_map"] != 0).sum().item())
ptr = lambda t: t.data_ptr()
stream = torch.cuda.current_stream().cuda_stream
B, T_w = ib["word_tokens"].shape
m.ctx.timer_enable(abi.TIMER_S2PA)
out = {}
for name in ("tensor_api", "resident_table"):
    def once():
        if name == "tensor_api":
            m.ctx.text2mel_encode(ptr(tb["word_tokens"]), ptr(tb["keys"]), ptr(tb["values"]), ptr(tb["key_map"]), ptr(tb["pinyin"]),
                                  ptr(tb["pinyin_map"]), ptr(tb["pron_modified"]), None, B, T_w, tb["keys"].shape[2], tb["pinyin"].shape[2], stream)
        else:
            m.ctx.text2mel_encode_ids(ptr(d["word_tokens"]), ptr(d["entry_ids"]), ptr(d["pron_modified"]), None, B, T_w, int(ib["L_k"]), int(ib["P"]), stream)
    once()
    torch.cuda.synchronize()
    m.ctx.timer_reset()
    for _ in range(a.n):
        once()
    torch.cuda.synchronize()
    ms, n = m.ctx.timer_read(abi.TIMER_S2PA)
    row_bytes, live = (6144, live_tensor) if name == "tensor_api" else (1536, live_table)
    out[name] = {"launches": n + 1, "us_per_launch": ms / max(n, 1) * 1e3, "live_gloss_rows": live, "algorithmic_bytes_per_launch": row_bytes * live,
                 "GBps": row_bytes * live / (ms / max(n, 1) * 1e-3) / 1e9}
print(json.dumps(out))
