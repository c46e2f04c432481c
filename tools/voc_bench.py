#!/usr/bin/env python3
"""Vocoder-only micro-benchmark used while tuning the HifiGAN kernels (not the headline bench): B utterances of
random mel with Biaobei-like lengths, reports ms / forward and the algorithmic TFLOP/s of the conv kernels."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from dict_tts_amd import abi, synth, vocoder

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=60)
ap.add_argument("--T", type=int, default=740)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--precision", default="f16")
ap.add_argument("--uniform", action="store_true", help="every utterance T frames long (tile-count experiments)")
ap.add_argument("--tune", type=int, default=0, help="dtts_config.tune_flags (A/B switches, include/dicttts_hip.h)")
ap.add_argument("--lib", default=None, help="path of the library build to load instead of the in-tree release library (A/B runs: nothing is copied over it)")
a = ap.parse_args()
if a.lib:
    abi.load_library(os.path.abspath(a.lib))
T_ = lambda x: torch.from_numpy(np.ascontiguousarray(x))
voc = vocoder.HifiGAN(state_dict={k: T_(v) for k, v in synth.hifigan_state_dict(1234).items()}, config={**synth.hifigan_config(), "dtts_tune_flags": a.tune},
                      precision=abi.VOC_PRECISIONS[a.precision])
voc.ctx.timer_enable(abi.TIMER_VOC_CONV)
rng = np.random.default_rng(0)
lens = np.clip(rng.normal(364, 110, a.B), 120, a.T).astype(np.int32)
lens[0] = a.T
if a.uniform:
    lens[:] = a.T
mel = torch.from_numpy(np.clip(rng.normal(-3, 1.2, (a.B, a.T, 80)), -6, 1.5).astype(np.float32)).cuda()
lens_d = torch.from_numpy(lens).cuda()
for _ in range(2):
    wav = voc.forward_batch(mel, lens_d)
torch.cuda.synchronize()
voc.ctx.timer_reset()
t0 = time.perf_counter()
for _ in range(a.iters):
    wav = voc.forward_batch(mel, lens_d)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.iters
ms, n = voc.ctx.timer_read(abi.TIMER_VOC_CONV)
frames = int(lens.sum())
import hashlib
print(f"wav md5 {hashlib.md5(wav.cpu().numpy().tobytes()).hexdigest()[:12]}  sum|wav| {float(wav.abs().double().sum()):.6f}  nonfinite-counter {int(voc.ctx.vocoder_nonfinite())}")   # bit-identity across builds
print(f"frames {frames}  wall {dt * 1e3:.2f} ms/forward  conv-kernel {ms / a.iters:.2f} ms/forward  "
      f"{614105088 * frames / (ms / a.iters * 1e-3) / 1e12:.1f} TFLOP/s  ({frames / dt:.0f} frames/s vocoder-only)")
if os.environ.get("DTTS_CALIB"):
    # known-byte-count kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE (MI355X_MICROARCH.md §HBM)
    big = torch.empty(256 * 1024 * 1024 // 4, device="cuda")   # 256 MiB
    big.fill_(1.0)
    torch.cuda.synchronize()
    c = big.clone()                                             # reads 256 MiB, writes 256 MiB
    torch.cuda.synchronize()
    s = big.sum()                                               # reads 256 MiB
    torch.cuda.synchronize()
