#!/usr/bin/env python3
"""Text->mel micro-benchmark used while tuning the acoustic model (not the headline bench): the first Biaobei batch of B
utterances through dtts_text2mel_encode_ids + dtts_text2mel_decode on one stream, nothing else on the GPU.
usage: python tools/t2m_bench.py [--B 60] [--iters 20] [--lib build/x/variant.so]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from dict_tts_amd import abi, model, synth

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=60)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--lib", default=None, help="path of the library build to load instead of the in-tree release library (A/B runs)")
a = ap.parse_args()
if a.lib:
    abi.load_library(os.path.abspath(a.lib))
T = lambda x: torch.from_numpy(np.ascontiguousarray(x))
sd = synth.dict_tts_state_dict(1234)
sd["dur_predictor.linear.0.bias"] = np.array([3.09], np.float32)
m = model.PortaSpeech_dict(hparams={})
m.load_state_dict({k: T(v) for k, v in sd.items()})
st = synth.biaobei_struct()
table = synth.dict_table(1234)
m.upload_dict_table(table)
ib = synth.make_id_batch(st["sentences"][:a.B], table)
dev = torch.device("cuda")
d = {k: T(ib[k]).to(dev) for k in ("word_tokens", "entry_ids", "pron_modified")}
ptr = lambda t: t.data_ptr()
s = torch.cuda.current_stream().cuda_stream
m.ctx.timer_enable(abi.TIMER_S2PA)


def once(ev=None):
    if ev:
        ev[0].record()
    T_mel = m.ctx.text2mel_encode_ids(ptr(d["word_tokens"]), ptr(d["entry_ids"]), ptr(d["pron_modified"]), None, a.B, ib["word_tokens"].shape[1],
                                      ib["L_k"], ib["P"], s)
    if ev:
        ev[1].record()
    mel = torch.empty(a.B, T_mel, 80, device=dev)
    m.ctx.text2mel_decode(None, mel.data_ptr(), s)
    if ev:
        ev[2].record()
    return T_mel, mel


for _ in range(3):
    once()
torch.cuda.synchronize()
m.ctx.timer_reset()
enc = dec = 0.0
for _ in range(a.iters):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    T_mel, mel = once(ev)
    torch.cuda.synchronize()
    enc += ev[0].elapsed_time(ev[1]) / a.iters
    dec += ev[1].elapsed_time(ev[2]) / a.iters
s2_ms, s2_n = m.ctx.timer_read(abi.TIMER_S2PA)
print(f"B={a.B} T_w={ib['word_tokens'].shape[1]} T_mel={T_mel}: encode {enc:.3f} ms  decode {dec:.3f} ms  text->mel {enc + dec:.3f} ms  "
      f"s2pa {1e3 * s2_ms / max(s2_n, 1):.1f} us  mel checksum {float(mel.double().sum()):.6f}")
