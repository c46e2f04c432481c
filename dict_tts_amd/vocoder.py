"""Drop-in for the reference vocoder plugin ``vocoders.hifigan.HifiGAN`` (vocoders/hifigan.py:38-62), running the
HifiGAN generator as HIP kernels through libdicttts_hip.so.

Select it the way the reference selects any vocoder (vocoders/base_vocoder.py:15-23):
``--hparams=vocoder=dict_tts_amd.vocoder.HifiGAN``.  The no-argument constructor reads ``hparams['vocoder_ckpt']``
exactly as the reference does (``config.yaml`` + newest ``model_ckpt_steps_*.ckpt`` -> ``state_dict.model_gen``,
or ``config.json`` + ``generator_v1`` -> ``generator``); weight norm is folded inside the library.
"""
import glob
import json
import os
import re

import numpy as np
import torch

from . import abi
from . import hparams as hparams_mod
from .hparams import HIFIGAN_DEFAULTS, fill_abi_config, load_config_chain

VOCODERS = {}


def register_vocoder(cls):
    """vocoders/base_vocoder.py:9-12"""
    VOCODERS[cls.__name__.lower()] = cls
    VOCODERS[cls.__name__] = cls
    return cls


def get_vocoder_cls(hp):
    """vocoders/base_vocoder.py:15-23: registry key or dotted path"""
    import importlib
    name = hp["vocoder"]
    if name in VOCODERS:
        return VOCODERS[name]
    pkg, cls_name = ".".join(name.split(".")[:-1]), name.split(".")[-1]
    return getattr(importlib.import_module(pkg), cls_name)


def find_vocoder_checkpoint(base_dir):
    """-> (config dict, state dict) following vocoders/hifigan.py:16-32,41-52"""
    config_path = f"{base_dir}/config.yaml"
    if os.path.exists(config_path):
        ckpts = sorted(glob.glob(f"{base_dir}/model_ckpt_steps_*.ckpt"),
                       key=lambda x: int(re.findall(r"model_ckpt_steps_(\d+)\.ckpt", x)[0]))
        if not ckpts:
            raise FileNotFoundError(f"no model_ckpt_steps_*.ckpt under {base_dir}")
        ckpt = torch.load(ckpts[-1], map_location="cpu", weights_only=False)
        return load_config_chain(config_path), ckpt["state_dict"]["model_gen"]
    config_path = f"{base_dir}/config.json"
    if os.path.exists(config_path):
        with open(config_path) as f:
            config = json.load(f)
        ckpt = torch.load(f"{base_dir}/generator_v1", map_location="cpu", weights_only=False)
        return config, ckpt["generator"]
    raise FileNotFoundError(f"neither config.yaml nor config.json under vocoder_ckpt={base_dir!r}")


@register_vocoder
class HifiGAN:
    """Same contract as the reference class: ``spec2wav(mel[T,80], **ignored) -> np.float32[T*hop]``."""

    FP16_MAX = 65504.0
    MEL_ABS_MAX = 6.0     # the reference's log10-mel lies in [-6, 1.5] (egs/egs_bases/tts/base.yaml:59-60): range of the static bound
    EST_SIGMAS = 16.0     # an RMS estimate within FP16_MAX / EST_SIGMAS of the limit: do not even try fp16

    def __init__(self, state_dict=None, config=None, precision=None, ctx=None, unfused=False, range_guard=None):
        """precision: None (AUTO) | 'f16' | 'bf16' | 'bf16x3' | abi.VOC_*.  Nothing is read from the process environment.

        fp16 operands have a narrower range than the reference's fp32 arithmetic (modules/hifigan/hifigan.py:51-58).  Whether
        DTTS_VOC_F16 is valid is DECIDED, not sampled (include/dicttts_hip.h: dtts_vocoder_fp16_bound / dtts_vocoder_nonfinite):
          * at construction, from the folded weights: ``fp16_status`` is 'proven' (worst-case bound of every fp16 operand < 65504 for
            |mel| <= MEL_ABS_MAX: no call in that range can overflow), 'checked' (not provable — every trained generator — fp16 under
            the always-on detector) or, AUTO only, 'rejected' (the propagated RMS estimate already nears the fp16 limit, or the fused
            kernels do not cover the generator's shape: DTTS_VOC_BF16X3 from the start);
          * on EVERY call, by the conv_post epilogue's detector: a forward in which any fp16 operand overflowed delivers non-finite
            pre-tanh values, which are counted and poisoned with NaN.  spec2wav / spec2wav_batch / forward_batch(check=True) read the
            count behind their synchronisation: AUTO redoes the call in DTTS_VOC_BF16X3 and keeps that mode, an explicit 'f16' raises.
            A pipelined caller (forward_batch without check) calls ``overflowed()`` after it has synchronised for the waveform.
        No call can return garbage silently.  range_guard=True additionally runs the CENSUS instantiations on every call (all 72
        conversion points counted, a few % slower) and raises on a clamp: the parity tests' mode."""
        if state_dict is None:
            config, state_dict = find_vocoder_checkpoint(hparams_mod.hparams["vocoder_ckpt"])   # looked up at call time: the
            # INTEGRATION.md hook may rebind dict_tts_amd.hparams.hparams after this module was imported
        self.config = {**HIFIGAN_DEFAULTS, **(config or {})}
        if not torch.cuda.is_available():
            raise abi.DttsError("dict_tts_amd.vocoder.HifiGAN needs a ROCm GPU: the HIP path has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device())
        auto = precision is None
        if auto:
            precision = abi.VOC_F16   # the waveform-exact default
        elif isinstance(precision, str):
            precision = abi.VOC_PRECISIONS[precision]
        self._unfused = unfused
        self._auto = auto
        self._census = False      # range_guard=True: the census instantiations on every call, a clamp raises
        self._state_dict = None   # AUTO in fp16: kept for the fallback (host tensors the caller handed over)
        self._seen_bad = 0
        self.fp16_status = None   # 'proven' | 'checked' | 'rejected' | None (not an fp16 mode)
        self.fp16_bound = None    # (worst_case, rms_estimate) at MEL_ABS_MAX
        if ctx is not None:
            # a shared context computes in ITS OWN mode (dtts_config.vocoder_precision), whatever was asked for here: an explicit precision
            # that disagrees is an error, AUTO takes the context's (it used to assume fp16 — a bf16 / bf16x3 context then claimed
            # 'proven' and synchronised for the detector for nothing)
            ctx_precision = int(ctx.cfg.vocoder_precision)
            if not auto and precision != ctx_precision:
                raise abi.DttsError(f"HifiGAN(ctx=..., precision={precision}): the shared context was created with vocoder_precision={ctx_precision}")
            self.precision = ctx_precision
            self._shared_ctx = True
            self.ctx = ctx
            self.ctx.load_state_dict("vocoder", state_dict)
            self.ctx.finalize(abi.PART_VOCODER)
        else:
            census = precision == abi.VOC_F16 and bool(range_guard)
            try:
                self._build(state_dict, precision, census)
            except abi.DttsError as e:
                if not (auto and "DTTS_VOC_BF16X3" in str(e)):
                    raise
                import warnings
                warnings.warn(f"HifiGAN: the fused fp16 kernels do not cover this generator ({e}); using DTTS_VOC_BF16X3")
                self._build(state_dict, abi.VOC_BF16X3, False)
                self.fp16_status = "rejected"
                census = False
            self._census = census
        if self.precision == abi.VOC_F16:
            wc, est = self.ctx.vocoder_fp16_bound(self.MEL_ABS_MAX)
            self.fp16_bound = (wc, est)
            self.fp16_status = "proven" if wc < self.FP16_MAX else "checked"
            if auto and ctx is None and est * self.EST_SIGMAS > self.FP16_MAX:
                import warnings
                warnings.warn(f"HifiGAN: the propagated RMS of an fp16 operand is {est:.3g} (limit {self.FP16_MAX:.0f}): "
                              f"DTTS_VOC_F16 is not valid for this checkpoint, using DTTS_VOC_BF16X3")
                self._build(state_dict, abi.VOC_BF16X3, False)
                self.fp16_status = "rejected"
            elif auto and ctx is None:
                self._state_dict = state_dict
            self._seen_bad = self.ctx.vocoder_nonfinite()
        self.hop = self.ctx.hop()

    def _build(self, state_dict, precision, guard):
        cfg = fill_abi_config(abi.default_config(), None, self.config, vocoder_precision=precision)
        cfg.vocoder_unfused = 1 if self._unfused else 0   # testing aid (bf16 mode): one kernel per convolution
        cfg.vocoder_range_guard = 1 if guard else 0
        ctx = abi.Context(cfg)
        ctx.load_state_dict("vocoder", state_dict)
        ctx.finalize(abi.PART_VOCODER)
        self.ctx, self.precision = ctx, precision
        self._seen_bad = 0   # (a new context counts from zero)

    # -- reference API -------------------------------------------------------------------------------------
    def spec2wav(self, mel, **kwargs):
        """vocoders/hifigan.py:54-62; one utterance, numpy in / numpy out"""
        c = torch.as_tensor(np.asarray(mel), dtype=torch.float32).unsqueeze(0).to(self.device)
        return self.forward_batch(c, check=True).view(-1).cpu().numpy()

    def overflowed(self):
        """True when a forward since the last call of this method (or construction) delivered non-finite pre-tanh samples, i.e. an fp16
        operand overflowed.  Only meaningful once the caller has synchronised with the stream(s) of those forwards."""
        n = self.ctx.vocoder_nonfinite()
        bad = n != self._seen_bad
        self._seen_bad = n
        return bad

    # -- batched fast path (the reference calls spec2wav once per utterance, tasks/tts/dict_tts.py:255) ------
    def forward_batch(self, mel, lens=None, check=False):
        """mel [B,T,80] float32 cuda tensor, lens [B] int32 cuda tensor or None -> wav [B, T*hop] cuda tensor.
        check=True: synchronise the stream behind the forward and act on the overflow detector (AUTO: redo in DTTS_VOC_BF16X3 and keep
        it; explicit fp16: raise).  check=False (pipelined callers): nothing synchronises here — call overflowed() after the waveform's
        own synchronisation and redo the batch with check=True when it says so (dict_tts_amd/infer.py does)."""
        assert mel.is_cuda and mel.dtype == torch.float32 and mel.dim() == 3
        mel = mel.contiguous()
        B, T, _ = mel.shape
        wav = torch.empty(B, T * self.hop, dtype=torch.float32, device=mel.device)
        if lens is not None:
            lens = lens.to(device=mel.device, dtype=torch.int32).contiguous()
        stream = torch.cuda.current_stream().cuda_stream
        lens_p = lens.data_ptr() if lens is not None else None
        self.ctx.hifigan_forward(mel.data_ptr(), lens_p, B, T, wav.data_ptr(), stream)
        if self._census:
            n = self.ctx.vocoder_clamped(stream)   # (synchronises the stream)
            if n:
                raise abi.DttsError(f"DTTS_VOC_F16: {n} activations exceeded the fp16 range (the reference computes in fp32, "
                                    f"modules/hifigan/hifigan.py:51-58); use precision='bf16x3'")
        if check and self.precision == abi.VOC_F16:
            torch.cuda.current_stream().synchronize()
            if self.overflowed():
                if not (self._auto and self._state_dict is not None):
                    how = ("create the shared context with vocoder_precision=DTTS_VOC_BF16X3 (a shared context cannot be rebuilt from here)"
                           if getattr(self, "_shared_ctx", False) else "use precision='bf16x3' or leave the precision to AUTO")
                    raise abi.DttsError("DTTS_VOC_F16: an fp16 operand overflowed in this call (non-finite pre-tanh samples; the reference "
                                        f"computes in fp32, modules/hifigan/hifigan.py:51-58); {how}")
                import warnings
                warnings.warn("HifiGAN: an fp16 operand overflowed in this call; switching to DTTS_VOC_BF16X3 and redoing it")
                self._build(self._state_dict, abi.VOC_BF16X3, False)
                self._state_dict = None
                self.fp16_status = "rejected"
                self.ctx.hifigan_forward(mel.data_ptr(), lens_p, B, T, wav.data_ptr(), stream)
        return wav

    def to_int16(self, wav, lens=None, norm=False):
        """save_wav's sample conversion (utils/audio.py:11-16) on the device: wav [B, T*hop] float32 cuda tensor as
        forward_batch returned it, lens [B] valid frames -> int16 cuda tensor [B, T*hop] (zeros past each utterance)"""
        assert wav.is_cuda and wav.dtype == torch.float32 and wav.dim() == 2 and wav.shape[1] % self.hop == 0
        wav = wav.contiguous()
        B, N = wav.shape
        out = torch.empty(B, N, dtype=torch.int16, device=wav.device)
        if lens is not None:
            lens = lens.to(device=wav.device, dtype=torch.int32).contiguous()
        self.ctx.wav_to_int16(wav.data_ptr(), lens.data_ptr() if lens is not None else None, B, N // self.hop, norm,
                              out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        return out

    def spec2wav_batch(self, mels):
        """list of [T_i,80] arrays -> list of float32 arrays [T_i*hop]; each equals spec2wav(mel_i) of the reference"""
        lens = [int(np.asarray(m).shape[0]) for m in mels]
        T = max(lens)
        batch = torch.zeros(len(mels), T, self.config.get("audio_num_mel_bins", 80), dtype=torch.float32)
        for i, m in enumerate(mels):
            batch[i, :lens[i]] = torch.as_tensor(np.asarray(m), dtype=torch.float32)
        wav = self.forward_batch(batch.to(self.device), torch.tensor(lens, dtype=torch.int32), check=True).cpu().numpy()
        return [wav[i, :lens[i] * self.hop] for i in range(len(mels))]
