#!/usr/bin/env python3
"""FFT-block stack micro-benchmark (not the headline bench): the FastspeechDecoder configuration of the
Dict-TTS config chain (hidden 192, 2 heads, 4 layers, k = 9) over a Biaobei-like batch of mel-frame sequences.
Reports ms / forward and algorithmic TFLOP/s against the fp32 MFMA roof (157.3 TFLOP/s)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from dict_tts_amd import fft, synth

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=60)
ap.add_argument("--T", type=int, default=740)
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
C, L, K, H = 192, 4, 9, 2
T_ = lambda x: torch.from_numpy(np.ascontiguousarray(x))
m = fft.FFTBlocks(C, L, ffn_kernel_size=K, num_heads=H, hparams={})
m.load_state_dict({k: T_(v) for k, v in synth.fft_blocks_state_dict(1234, C, L, K).items()})
rng = np.random.default_rng(0)
lens = np.clip(rng.normal(364, 110, a.B), 120, a.T).astype(np.int64)
lens[0] = a.T
x = rng.normal(0, 1, (a.B, a.T, C)).astype(np.float32)
for b, n in enumerate(lens):
    x[b, n:] = 0
xd = T_(x).cuda()
for _ in range(2):
    y = m(xd)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    y = m(xd)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.iters
tok = int(lens.sum())
# 2*MAC per VALID token and layer: qkv + out projections, attention over the utterance's own frames, FFN
flop = sum(L * n * (2 * C * 3 * C + 2 * C * C + 4 * n * C + 2 * C * 4 * C * K + 2 * 4 * C * C) for n in lens.tolist())
print(f"tokens {tok} (padded {a.B * a.T})  {dt * 1e3:.2f} ms/forward  {flop / dt / 1e12:.1f} TFLOP/s algorithmic "
      f"= {100 * flop / dt / 157.3e12:.1f} % of the fp32 MFMA roof  ({tok / dt:.0f} frames/s)")
