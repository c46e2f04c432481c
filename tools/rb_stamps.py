#!/usr/bin/env python3
"""Per-phase clock stamps of the whole-ResBlock kernels (a variant build with -DRB_STAMP: tools/build_variant.sh rbst "-DRB_STAMP" rblock.hip).
usage: python tools/rb_stamps.py build/x/rbst.so [tune_flags]  -> per (C, k): microseconds per tile and phase, wave 0 and the workgroup's last wave"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from dict_tts_amd import abi, synth, vocoder

lib = abi.load_library(os.path.abspath(sys.argv[1]))
tune = int(sys.argv[2]) if len(sys.argv) > 2 else 0   # dtts_config.tune_flags of the vocoder context
T_ = lambda x: torch.from_numpy(np.ascontiguousarray(x))
voc = vocoder.HifiGAN(state_dict={k: T_(v) for k, v in synth.hifigan_state_dict(1234).items()}, config={**synth.hifigan_config(), "dtts_tune_flags": tune}, precision="f16")
rng = np.random.default_rng(0)
B, T = 60, 740
lens = np.clip(rng.normal(364, 110, B), 120, T).astype(np.int32)
lens[0] = T
mel = torch.from_numpy(np.clip(rng.normal(-3, 1.2, (B, T, 80)), -6, 1.5).astype(np.float32)).cuda()
lens_d = torch.from_numpy(lens).cuda()
buf = (C.c_ulonglong * (2 * 12 * 16))()
lib.dtts_debug_rb_stamps.argtypes = [C.c_void_p, C.c_int]
for _ in range(2):
    voc.forward_batch(mel, lens_d)
torch.cuda.synchronize()
assert lib.dtts_debug_rb_stamps(buf, 1) == 0
N = 5
for _ in range(N):
    voc.forward_batch(mel, lens_d)
torch.cuda.synchronize()
assert lib.dtts_debug_rb_stamps(buf, 0) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(2, 12, 16).astype(np.float64)
names = ["rewrite x", "barrier (x written)", "conv1", "barrier (A read)", "rewrite xt", "barrier (xt written)", "conv2", "barrier (xt read)", "  epilogue: rest (conv_post, guard)", "-", "tile bookkeeping / wait for x", "tiles",
         "  epilogue: plan, stage-sum loads issued", "  epilogue: slabs -> staging rows, next x issued", "  epilogue: rows read back + sum, stored", "-"]
for si, (c, k) in enumerate((c, k) for c in (32, 64, 128, 256) for k in (3, 7, 11)):
    if a[0, si, 11] == 0:
        continue
    print(f"== C = {c}, k = {k}: {int(a[0, si, 11] / N)} tiles per forward; us per tile (wave 0 | last wave)")
    tot = [0.0, 0.0]
    for ph in (10, 0, 1, 2, 3, 4, 5, 6, 7, 12, 13, 14, 8):
        v = [a[w, si, ph] * 0.01 / max(a[w, si, 11], 1) for w in (0, 1)]
        tot = [tot[0] + v[0], tot[1] + v[1]]
        print(f"   {names[ph]:32s} {v[0]:7.2f} | {v[1]:7.2f}")
    print(f"   {'sum':32s} {tot[0]:7.2f} | {tot[1]:7.2f}")
