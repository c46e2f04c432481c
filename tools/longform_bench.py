#!/usr/bin/env python3
"""BASELINE configs[3] alone (1000 characters, ~5k mel frames, B=1, teacher-forced 5 frames/char): text->mel and vocoder times, for
`rocprofv3 --kernel-trace` + tools/timeline.py.  usage: python tools/longform_bench.py [reps] [--lib build/x/variant.so]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from dict_tts_amd import abi, model, synth, vocoder

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
if "--lib" in sys.argv:   # A/B runs: another build of the library, selected by path
    i = sys.argv.index("--lib")
    abi.load_library(os.path.abspath(sys.argv[i + 1]))
    del sys.argv[i:i + 2]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
sd = synth.dict_tts_state_dict(1234)
m = model.PortaSpeech_dict(hparams={})
m.load_state_dict({k: T(v) for k, v in sd.items()})
voc = vocoder.HifiGAN(state_dict={k: T(v) for k, v in synth.hifigan_state_dict(1234).items()}, config=synth.hifigan_config(), precision="f16")
st = synth.biaobei_struct()
dev = torch.device("cuda")
ids1000 = [w for s in st["sentences"] for w in s][:1000]
batch = synth.make_batch([ids1000], 1234)
m2w = T(synth.teacher_mel2word(batch["word_tokens"], 5, 5)).to(dev)
b = {k: T(v).to(dev) for k, v in batch.items()}
z = torch.randn(1, 16, 4096, device=dev)
T_mel = int(m2w.shape[1])
zp = z[:, :, : (T_mel + 3) // 4].contiguous()


def once(ev=None):
    if ev:
        ev[0].record()
    out = m((b["word_tokens"], None), b["pron_modified"], (None,) * 3, None, None,
            (b["keys"], b["values"], b["key_map"], b["pinyin"], b["pinyin_map"]), infer=True, mel2word=m2w, z_p=zp)
    if ev:
        ev[1].record()
    wav = voc.forward_batch(out["mel_out"], out["mel_lens"])
    if ev:
        ev[2].record()
    return out, wav


for _ in range(2):
    once()
torch.cuda.synchronize()
t0 = time.perf_counter()
a = c = 0.0
for _ in range(reps):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    out, wav = once(ev)
    torch.cuda.synchronize()
    a += ev[0].elapsed_time(ev[1]) / reps
    c += ev[1].elapsed_time(ev[2]) / reps
dt = (time.perf_counter() - t0) / reps
fr = int(out["mel_lens"].sum())
print(f"long form T_w={batch['word_tokens'].shape[1]} T_mel={fr}: {dt * 1e3:.2f} ms wall (text->mel {a:.2f} ms, vocoder {c:.2f} ms), {fr / dt:.0f} frames/s")
