#!/usr/bin/env python3
"""BASELINE configs[0]: one Biaobei sentence, B=1, text -> mel -> wav, one stream; wall time per utterance and (under rocprofv3) the
kernel-time share of it.  usage: python tools/b1_bench.py [reps] [--lib build/x/variant.so]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from dict_tts_amd import abi, model, synth, vocoder

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
if "--lib" in sys.argv:   # A/B runs: another build of the library, selected by path (nothing is copied over the release library)
    i = sys.argv.index("--lib")
    abi.load_library(os.path.abspath(sys.argv[i + 1]))
    del sys.argv[i:i + 2]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
sd = synth.dict_tts_state_dict(1234)
sd["dur_predictor.linear.0.bias"] = np.array([3.09], np.float32)
m = model.PortaSpeech_dict(hparams={})
m.load_state_dict({k: T(v) for k, v in sd.items()})
voc = vocoder.HifiGAN(state_dict={k: T(v) for k, v in synth.hifigan_state_dict(1234).items()}, config=synth.hifigan_config(), precision="f16")
st = synth.biaobei_struct()
table = synth.dict_table(1234)
m.upload_dict_table(table)
ib = synth.make_id_batch([st["sentences"][0]], table)
dev = torch.device("cuda")
d = {k: T(ib[k]).to(dev) for k in ("word_tokens", "entry_ids", "pron_modified")}
ptr = lambda t: t.data_ptr()
s = torch.cuda.current_stream().cuda_stream


def once():
    T_mel = m.ctx.text2mel_encode_ids(ptr(d["word_tokens"]), ptr(d["entry_ids"]), ptr(d["pron_modified"]), None, 1, ib["word_tokens"].shape[1],
                                      ib["L_k"], ib["P"], s)
    mel = torch.empty(1, T_mel, 80, device=dev)
    m.ctx.text2mel_decode(None, mel.data_ptr(), s)
    lens = torch.empty(1, dtype=torch.int32, device=dev)
    m.ctx.fetch(abi.OUT_MEL_LENS, lens.data_ptr(), s)
    wav = voc.forward_batch(mel, lens)
    return T_mel, wav


for _ in range(3):
    once()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    T_mel, wav = once()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"B=1, {T_mel} frames ({T_mel * 256 / 22050:.2f} s of audio): {dt * 1e3:.3f} ms per utterance end to end, RTF {dt / (T_mel * 256 / 22050):.2e}, reps {reps}")
