#!/usr/bin/env python3
"""Which vocoder mode does THIS HifiGAN checkpoint get, and is it within the waveform gate?  (test tooling: imports the oracle)

Every number in this repo is measured on seeded synthetic weights: the reference's pretrained vocoder (readme.md:70,101) sits behind
links the build container cannot reach.  The one decision that depends on the real weights is whether the fp16 ResBlock operands of
DTTS_VOC_F16 are valid for them (the reference computes in fp32, modules/hifigan/hifigan.py:51-58).  This tool closes that in one
command on a machine that has the checkpoint and a GPU:

    python tools/validate_checkpoint.py /path/to/vocoder_ckpt_dir            # config.yaml + model_ckpt_steps_*.ckpt, or config.json + generator_v1
    python tools/validate_checkpoint.py /path/to/dir --mel-npy a.npy b.npy   # real mels [T,80] instead of synthetic ones
    python tools/validate_checkpoint.py --synthetic                           # the repo's seeded generator (what CI can run)

It prints, in this order:
  1. the static decision at load time (dict_tts_amd.vocoder.HifiGAN, AUTO): worst-case bound and propagated-RMS estimate of the fp16
     operands for |mel| <= 6 (dtts_vocoder_fp16_bound), the margin rule `rms_estimate x EST_SIGMAS < 65504`, and the resulting status
     'proven' | 'checked' | 'rejected';
  2. the MEASURED crest of the operands on the CPU oracle: max |operand| over every ResBlock conversion point and mel, against the static
     RMS estimate — the number EST_SIGMAS stands for (a checkpoint whose measured peak / estimate exceeds EST_SIGMAS needs a larger margin);
  3. the census range guard over all mels (dtts_vocoder_range_guard: every one of the 72 conversion points counted) and the always-on
     detector's count (dtts_vocoder_nonfinite);
  4. the waveform gate of BASELINE.json (RMS(gpu - ref) <= 1e-4, |RMS(gpu) - RMS(ref)| <= 1e-4) of fp16 and of bf16x3 against the oracle on
     the first --gate mels;
  5. the verdict: the mode HifiGAN(precision=None) runs this checkpoint in, and whether that mode met the gate.
Exit code 0 = the mode AUTO chose meets the gate with a clean guard; 1 otherwise.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import torch.nn.functional as F

from dict_tts_amd import abi, synth, vocoder
from oracle import hifigan_ref as href


VOC_NAMES = {v: f"DTTS_VOC_{k.upper()}" for k, v in abi.VOC_PRECISIONS.items()}


def rms(a):
    return float(np.sqrt(np.mean(np.square(np.asarray(a, np.float64)))))


def operand_peaks(fsd, cfg, mel_T80):
    """max |x| of every fp16 conversion point of DTTS_VOC_F16 (the ResBlock inputs x of each iteration and the intermediates xt) in the exact
    fp32 forward of the oracle's arithmetic (oracle/hifigan_ref.py:generator_forward, restated here with the taps exposed)"""
    L = href.LRELU_SLOPE
    peak = 0.0
    with torch.no_grad():
        x = torch.as_tensor(mel_T80, dtype=torch.float32).unsqueeze(0).transpose(2, 1)
        x = F.conv1d(x, fsd["conv_pre.weight"], fsd["conv_pre.bias"], padding=3)
        nk = len(cfg["resblock_kernel_sizes"])
        for i, (u, k) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
            x = F.conv_transpose1d(F.leaky_relu(x, L), fsd[f"ups.{i}.weight"], fsd[f"ups.{i}.bias"], stride=u, padding=(k - u) // 2)
            xs = None
            for j, (rk, rd) in enumerate(zip(cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"])):
                p, r = f"resblocks.{i * nk + j}", x
                for m, d in enumerate(rd):
                    peak = max(peak, float(r.abs().max()))
                    xt = F.conv1d(F.leaky_relu(r, L), fsd[f"{p}.convs1.{m}.weight"], fsd[f"{p}.convs1.{m}.bias"], padding=href._pad(rk, d), dilation=d)
                    peak = max(peak, float(xt.abs().max()))
                    r = F.conv1d(F.leaky_relu(xt, L), fsd[f"{p}.convs2.{m}.weight"], fsd[f"{p}.convs2.{m}.bias"], padding=href._pad(rk, 1)) + r
                xs = r if xs is None else xs + r
            x = xs / nk
    return peak


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("ckpt_dir", nargs="?", help="vocoder checkpoint directory (vocoders/hifigan.py:16-52 discovery rule)")
    ap.add_argument("--synthetic", action="store_true", help="the repo's seeded synthetic generator instead of a checkpoint")
    ap.add_argument("--scale-resblocks", type=float, default=1.0, help="(with --synthetic) multiply every ResBlock weight_g: moves the checkpoint across the margin")
    ap.add_argument("--mels", type=int, default=16, help="synthetic mels to run when no --mel-npy is given")
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--mel-npy", nargs="*", default=[], help="real mels, each [T,80] float32 (.npy)")
    ap.add_argument("--gate", type=int, default=3, help="mels compared with the CPU oracle (fp32 torch; ~1 s per 100 frames)")
    a = ap.parse_args()
    if not torch.cuda.is_available():
        sys.exit("validate_checkpoint.py needs a ROCm GPU (the HIP path has no CPU fallback)")
    T_ = lambda x: torch.from_numpy(np.ascontiguousarray(x))
    if a.synthetic:
        cfg = synth.hifigan_config()
        sd = {k: T_(v) for k, v in synth.hifigan_state_dict(1234).items()}
        if a.scale_resblocks != 1.0:
            sd = {k: (v * a.scale_resblocks if k.startswith("resblocks.") and k.endswith("weight_g") else v) for k, v in sd.items()}
        src = f"synthetic generator (seed 1234, ResBlock gains x{a.scale_resblocks:g})"
    elif a.ckpt_dir:
        cfg, sd = vocoder.find_vocoder_checkpoint(a.ckpt_dir)
        src = a.ckpt_dir
    else:
        ap.error("give a checkpoint directory or --synthetic")
    full_cfg = {**vocoder.HIFIGAN_DEFAULTS, **(cfg or {})}
    mels = [np.load(f).astype(np.float32) for f in a.mel_npy] or [synth.random_mel(100 + i, a.frames, f"val{i}") for i in range(a.mels)]
    print(f"checkpoint: {src}\n  upsample_rates {full_cfg['upsample_rates']}  initial channels {full_cfg['upsample_initial_channel']}  "
          f"resblock kernels {full_cfg['resblock_kernel_sizes']}\n  {len(mels)} mels, {sum(m.shape[0] for m in mels)} frames, "
          f"|mel| max {max(float(np.abs(m).max()) for m in mels):.2f}")

    import warnings
    with warnings.catch_warnings(record=True) as ws:
        warnings.simplefilter("always")
        auto = vocoder.HifiGAN(state_dict=sd, config=cfg)          # what a user gets
    for w in ws:
        print(f"  [warning] {w.message}")
    H = vocoder.HifiGAN
    print(f"\n1. static decision (dtts_finalize_weights): status = {auto.fp16_status!r}, mode = {VOC_NAMES.get(auto.precision, auto.precision)}")
    probe = auto if auto.precision == abi.VOC_F16 else None
    if probe is None:
        try:
            probe = vocoder.HifiGAN(state_dict=sd, config=cfg, precision="f16")
        except abi.DttsError as e:
            print(f"   explicit fp16 is not available for this generator: {e}")
    wc = est = None
    if probe is not None:
        wc, est = probe.ctx.vocoder_fp16_bound(H.MEL_ABS_MAX)
        print(f"   worst-case bound of an fp16 operand for |mel| <= {H.MEL_ABS_MAX:g}: {wc:.4g}   ({'<' if wc < H.FP16_MAX else '>='} {H.FP16_MAX:.0f}: "
              f"{'PROVEN safe' if wc < H.FP16_MAX else 'not provable (every trained-size generator)'})")
        print(f"   propagated RMS estimate: {est:.4g};  margin rule: estimate x EST_SIGMAS ({H.EST_SIGMAS:g}) = {est * H.EST_SIGMAS:.4g} "
              f"{'<' if est * H.EST_SIGMAS <= H.FP16_MAX else '>'} {H.FP16_MAX:.0f}  ->  {'fp16 under the detector' if est * H.EST_SIGMAS <= H.FP16_MAX else 'REJECTED statically (bf16x3)'}")

    fsd = href.fold_weight_norm(sd)
    peak = max(operand_peaks(fsd, full_cfg, m) for m in mels[:max(a.gate, 1)])
    print(f"\n2. measured on the CPU oracle ({min(len(mels), max(a.gate, 1))} mels): max |fp16 operand| = {peak:.4g}  "
          f"({peak / H.FP16_MAX:.2e} of the fp16 range)")
    if est:
        print(f"   measured peak / static RMS estimate = {peak / est:.2f}   (EST_SIGMAS = {H.EST_SIGMAS:g} is the crest this ratio must stay below for the "
              f"static rule to be conservative)")

    print("\n3. census range guard + always-on detector over all mels")
    clamped = bad = None
    if probe is not None:
        g = vocoder.HifiGAN(state_dict=sd, config=cfg, precision="f16")
        g.ctx.vocoder_range_guard(True)
        clamped = 0
        s = torch.cuda.current_stream().cuda_stream
        for m in mels:
            g.forward_batch(T_(m[None]).cuda())
            clamped += g.ctx.vocoder_clamped(s)
        torch.cuda.synchronize()
        bad = int(g.ctx.vocoder_nonfinite())
        print(f"   fp16: {clamped} activations beyond the fp16 range (72 conversion points x every output row), detector count {bad}")

    print(f"\n4. waveform gate against the oracle (first {a.gate} mels): RMS(gpu - ref) <= 1e-4 and |RMS(gpu) - RMS(ref)| <= 1e-4")
    gate = {}
    modes = [("f16", probe)] if probe is not None else []
    modes.append(("bf16x3", vocoder.HifiGAN(state_dict=sd, config=cfg, precision="bf16x3")))
    for name, v in modes:
        worst = (0.0, 0.0)
        ok = True
        for m in mels[:a.gate]:
            ref = href.spec2wav(fsd, full_cfg, m).numpy()
            try:
                w = v.forward_batch(T_(m[None]).cuda()).view(-1).cpu().numpy()
            except abi.DttsError as e:
                print(f"   {name}: {e}")
                ok = False
                break
            if not np.isfinite(w).all():
                ok = False
                worst = (float("inf"), float("inf"))
                break
            worst = (max(worst[0], rms(w - ref)), max(worst[1], abs(rms(w) - rms(ref))))
        ok = ok and worst[0] <= 1e-4 and worst[1] <= 1e-4
        gate[name] = ok
        print(f"   {name:7s} rms(diff) {worst[0]:.3e}  |drms| {worst[1]:.2e}  ->  {'PASS' if ok else 'FAIL'}")

    chosen = VOC_NAMES.get(auto.precision, str(auto.precision))
    chosen_key = "f16" if auto.precision == abi.VOC_F16 else "bf16x3"
    clean = chosen_key != "f16" or (clamped == 0 and bad == 0)
    good = gate.get(chosen_key, False) and clean
    print(f"\n5. verdict: HifiGAN(precision=None) runs this checkpoint in {chosen} (status {auto.fp16_status!r}); gate {'met' if gate.get(chosen_key) else 'NOT met'}, "
          f"guard {'clean' if clean else f'NOT clean ({clamped} clamped, {bad} non-finite)'}.")
    if chosen_key == "f16" and not clean:
        print("   -> fp16 overflows on these mels although the static rule admitted it: use precision='bf16x3' and raise EST_SIGMAS "
              "(dict_tts_amd/vocoder.py) to above the ratio of section 2.")
    sys.exit(0 if good else 1)


if __name__ == "__main__":
    main()
