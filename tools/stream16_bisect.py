"""Which hop of the 16-bit inter-iteration stream changes the waveform, and by how much?  Compares DTTS_VOC_F16 with the fp16 stream (tune 0) against the fp32
stream (tune_flags bit 15) on the same mels; with the ablation library (LIB=dict_tts_amd/libdicttts_abl.so) DTTS_S16 masks single hops (bit 2 i: iteration 0 -> 1 of
stage i, bit 2 i + 1: iteration 1 -> 2).  usage: [LIB=...] [DTTS_S16=mask] [B=2 T=48] python tools/stream16_bisect.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from dict_tts_amd import abi, synth, vocoder
lib = os.environ.get("LIB")
if lib: abi.load_library(os.path.abspath(lib))
T_ = lambda x: torch.from_numpy(np.ascontiguousarray(x))
sd = {k: T_(v) for k, v in synth.hifigan_state_dict(1234).items()}
B, T = int(os.environ.get("B", 2)), int(os.environ.get("T", 48))
rng = np.random.default_rng(0)
mel = torch.from_numpy(np.clip(rng.normal(-3, 1.2, (B, T, 80)), -6, 1.5).astype(np.float32)).cuda()
lens = torch.full((B,), T, dtype=torch.int32).cuda()
out = {}
for t in (32768, 0):
    v = vocoder.HifiGAN(state_dict=sd, config={**synth.hifigan_config(), "dtts_tune_flags": t}, precision=abi.VOC_F16)
    w = v.forward_batch(mel, lens); torch.cuda.synchronize()
    out[t] = w.cpu().numpy().astype(np.float64)
d = out[0] - out[32768]
print(f"S16={os.environ.get('DTTS_S16')} B={B} T={T}: rms(ref) {np.sqrt((out[32768]**2).mean()):.4f} rms(diff) {np.sqrt((d**2).mean()):.3e} max {np.abs(d).max():.3e} finite {np.isfinite(out[0]).all()}")
