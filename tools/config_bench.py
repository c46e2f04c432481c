#!/usr/bin/env python3
"""Timings of the other BASELINE.json configurations (parity for them lives in tests/test_gpu_parity.py; these are not
bench.py lines): configs[0] single sentence B=1, configs[3] long form (1000 chars, ~5k mel frames, B=1), configs[4]
dictionary stress (one GPU's share: B=32 mixed-length utterances, resident dictionary table vs the collated tensors)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from dict_tts_amd import model, synth, vocoder

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
sd = synth.dict_tts_state_dict(1234)
sd["dur_predictor.linear.0.bias"] = np.array([3.09], np.float32)     # ~22 frames per word, as bench.py
m = model.PortaSpeech_dict(hparams={})
m.load_state_dict({k: T(v) for k, v in sd.items()})
voc = vocoder.HifiGAN(state_dict={k: T(v) for k, v in synth.hifigan_state_dict(1234).items()}, config=synth.hifigan_config(), precision="f16")
st = synth.biaobei_struct()
dev = torch.device("cuda")


def run(batch, mel2word=None, reps=10, ids=None):
    b = {k: T(v).to(dev) for k, v in batch.items()} if ids is None else None
    z_all = torch.randn(len(batch["word_tokens"]) if ids is None else ids[0].shape[0], 16, 4096, device=dev)
    m2w = None if mel2word is None else T(mel2word).to(dev)

    def once():
        if ids is None:
            out = m((b["word_tokens"], None), b["pron_modified"], (None,) * 3, None, None,
                    (b["keys"], b["values"], b["key_map"], b["pinyin"], b["pinyin_map"]), infer=True, mel2word=m2w, z_p=Z[0])
        else:
            out = m.forward_ids(ids[0], ids[1], ids[2], ids[3], ids[4], z_p=Z[0])
        wav = voc.forward_batch(out["mel_out"], out["mel_lens"])
        return out, wav
    Z = [None]
    # the prior noise must match T_mel/4: encode once through the ABI to learn T_mel
    if ids is None:
        ptr = lambda t: t.data_ptr()
        T_mel = m.ctx.text2mel_encode(ptr(b["word_tokens"]), ptr(b["keys"]), ptr(b["values"]), ptr(b["key_map"]), ptr(b["pinyin"]),
                                      ptr(b["pinyin_map"]), ptr(b["pron_modified"]), (m2w.data_ptr(), m2w.shape[1]) if m2w is not None else None,
                                      *b["word_tokens"].shape, b["keys"].shape[2], b["pinyin"].shape[2], torch.cuda.current_stream().cuda_stream)
    else:
        T_mel = m.ctx.text2mel_encode_ids(ids[0].data_ptr(), ids[1].data_ptr(), ids[2].data_ptr(), None, *ids[0].shape, ids[3], ids[4],
                                          torch.cuda.current_stream().cuda_stream)
    Z[0] = z_all[:, :, : T_mel // 4].contiguous()
    for _ in range(2):
        out, wav = once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out, wav = once()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    frames = int(out["mel_lens"].sum())
    return dt, frames, T_mel


# configs[0]: sentence #1, B=1
dt, fr, tm = run(synth.make_batch([st["sentences"][0]], 1234), reps=30)
print(f"configs[0] single sentence B=1: {dt * 1e3:.2f} ms for {fr} frames ({fr * 256 / 22050:.2f} s of audio), RTF {dt / (fr * 256 / 22050):.2e}")
# configs[3]: 1000 chars, 5 frames/char teacher-forced
ids1000 = [w for s in st["sentences"] for w in s][:1000]
batch = synth.make_batch([ids1000], 1234)
m2w = synth.teacher_mel2word(batch["word_tokens"], 5, 5)
dt, fr, tm = run(batch, mel2word=m2w, reps=10)
print(f"configs[3] long form T_w=1002 -> T_mel={tm}: {dt * 1e3:.2f} ms for {fr * 256 / 22050:.1f} s of audio, RTF {dt / (fr * 256 / 22050):.2e}, {fr / dt:.0f} frames/s")
# configs[4]: B=32 mixed lengths drawn from ALL 7,030 zh-dict.json entries, heteronyms x5 (one GPU's share of B=128)
rng = np.random.default_rng(5)
full = synth.zh_dict_struct()["entries"]
eids = np.array(sorted(full.keys()))
wts = np.array([5.0 if len(full[i]) > 1 else 1.0 for i in eids])
wts /= wts.sum()
sents = [rng.choice(eids, size=int(rng.integers(6, 61)), p=wts).tolist() for _ in range(32)]
batch = synth.make_batch(sents, 1234, full, pron_every=3)
dt, fr, tm = run(batch, reps=10)
print(f"configs[4] dictionary stress B=32 (collated tensors, L_k={batch['keys'].shape[2]}): {dt * 1e3:.2f} ms/batch, {fr / dt:.0f} frames/s")
t0 = time.perf_counter()
table = synth.dict_table(1234, full)
t1 = time.perf_counter()
m.upload_dict_table(table)
torch.cuda.synchronize()
print(f"full dictionary table: {len(table['L'])} entries, {table['keys'].shape[0]} gloss rows, {table['keys'].nbytes / 2**20:.0f} MiB of raw gloss rows on the host, "
      f"{table['keys'].shape[0] * 2 * 192 * 4 / 2**20:.0f} MiB resident (projected K + V, 192 wide each); "
      f"host build {t1 - t0:.1f} s, upload {time.perf_counter() - t1:.2f} s (once)")
ib = synth.make_id_batch(sents, table, pron_every=3)
ids = (T(ib["word_tokens"]).to(dev), T(ib["entry_ids"]).to(dev).to(torch.int32), T(ib["pron_modified"]).to(dev), ib["L_k"], ib["P"])
dt, fr, tm = run(batch, reps=10, ids=ids)
print(f"configs[4] dictionary stress B=32 (resident table + ids): {dt * 1e3:.2f} ms/batch, {fr / dt:.0f} frames/s")
