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

    GUARD_CALLS = 4    # precision not chosen explicitly: the first calls run with the fp16 range guard on ...
    GUARD_EVERY = 16   # ... and afterwards every GUARD_EVERY-th call does (overflow depends on the input, not only on the checkpoint)

    def __init__(self, state_dict=None, config=None, precision=None, ctx=None, unfused=False, range_guard=None):
        """precision: None (= env DTTS_VOCODER_PRECISION, else AUTO) | 'f16' | 'bf16' | 'bf16x3' | abi.VOC_*.
        AUTO = DTTS_VOC_F16 (the waveform-exact default) with two safety nets, because fp16 operands have a narrower range than
        the reference's fp32 arithmetic: (1) a generator shape the fused fp16 kernels do not cover falls back to DTTS_VOC_BF16X3
        at construction; (2) the first GUARD_CALLS forward calls, and every GUARD_EVERY-th call after them, run with the library's
        range guard on (dtts_vocoder_range_guard; a guarded call synchronises its stream to read the count) and a call that
        saturated / overflowed an fp16 activation is REDONE in DTTS_VOC_BF16X3, which the object then keeps.
        An explicit precision is taken literally; range_guard=True then keeps the guard on for every call and raises on a clamp."""
        if state_dict is None:
            config, state_dict = find_vocoder_checkpoint(hparams_mod.hparams["vocoder_ckpt"])   # looked up at call time: the
            # INTEGRATION.md hook may rebind dict_tts_amd.hparams.hparams after this module was imported
        self.config = {**HIFIGAN_DEFAULTS, **(config or {})}
        if not torch.cuda.is_available():
            raise abi.DttsError("dict_tts_amd.vocoder.HifiGAN needs a ROCm GPU: the HIP path has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device())
        if precision is None and os.environ.get("DTTS_VOCODER_PRECISION"):
            precision = os.environ["DTTS_VOCODER_PRECISION"]
        auto = precision is None
        if auto:
            precision = abi.VOC_F16   # the waveform-exact default
        elif isinstance(precision, str):
            precision = abi.VOC_PRECISIONS[precision]
        self._unfused = unfused
        self._guard_left = 0      # -1: every call guarded (range_guard=True); > 0: the initial guarded calls left; 0 + _auto_guard: sampled
        self._guard_raise = False
        self._auto_guard = False
        self._calls = 0
        self._state_dict = None
        if ctx is not None:
            self.precision = precision
            self.ctx = ctx
            self.ctx.load_state_dict("vocoder", state_dict)
            self.ctx.finalize(abi.PART_VOCODER)
        else:
            guard = precision == abi.VOC_F16 and (auto or bool(range_guard))
            try:
                self._build(state_dict, precision, guard)
            except abi.DttsError as e:
                if not (auto and "DTTS_VOC_BF16X3" in str(e)):
                    raise
                import warnings
                warnings.warn(f"HifiGAN: the fused fp16 kernels do not cover this generator ({e}); using DTTS_VOC_BF16X3")
                self._build(state_dict, abi.VOC_BF16X3, False)
                guard = False
            if guard:
                self._guard_left = -1 if range_guard else self.GUARD_CALLS
                self._guard_raise = bool(range_guard) and not auto
                self._auto_guard = auto
                self._state_dict = state_dict if auto else None   # kept for the fallback (host tensors the caller handed over)
        self.hop = self.ctx.hop()

    def _build(self, state_dict, precision, guard):
        cfg = fill_abi_config(abi.default_config(), None, self.config, vocoder_precision=precision)
        cfg.vocoder_unfused = 1 if self._unfused else 0   # testing aid (bf16 mode): one kernel per convolution
        cfg.vocoder_range_guard = 1 if guard else 0
        ctx = abi.Context(cfg)
        ctx.load_state_dict("vocoder", state_dict)
        ctx.finalize(abi.PART_VOCODER)
        self.ctx, self.precision = ctx, precision

    # -- reference API -------------------------------------------------------------------------------------
    def spec2wav(self, mel, **kwargs):
        """vocoders/hifigan.py:54-62; one utterance, numpy in / numpy out"""
        c = torch.as_tensor(np.asarray(mel), dtype=torch.float32).unsqueeze(0).to(self.device)
        return self.forward_batch(c).view(-1).cpu().numpy()

    # -- batched fast path (the reference calls spec2wav once per utterance, tasks/tts/dict_tts.py:255) ------
    def forward_batch(self, mel, lens=None):
        """mel [B,T,80] float32 cuda tensor, lens [B] int32 cuda tensor or None -> wav [B, T*hop] cuda tensor"""
        assert mel.is_cuda and mel.dtype == torch.float32 and mel.dim() == 3
        mel = mel.contiguous()
        B, T, _ = mel.shape
        wav = torch.empty(B, T * self.hop, dtype=torch.float32, device=mel.device)
        if lens is not None:
            lens = lens.to(device=mel.device, dtype=torch.int32).contiguous()
        stream = torch.cuda.current_stream().cuda_stream
        sampled = False
        if self._auto_guard and self._guard_left == 0:   # past the initial guarded calls: every GUARD_EVERY-th call is guarded again
            self._calls += 1
            sampled = self._calls % self.GUARD_EVERY == 0
            if sampled:
                self.ctx.vocoder_range_guard(True)
        self.ctx.hifigan_forward(mel.data_ptr(), lens.data_ptr() if lens is not None else None, B, T, wav.data_ptr(), stream)
        if self._guard_left or sampled:
            n = self.ctx.vocoder_clamped(stream)   # (synchronises the stream: only during the guarded calls)
            if n:
                if self._guard_raise or self._state_dict is None:
                    raise abi.DttsError(f"DTTS_VOC_F16: {n} activations exceeded the fp16 range (the reference computes in fp32, "
                                        f"modules/hifigan/hifigan.py:51-58); use precision='bf16x3'")
                import warnings
                warnings.warn(f"HifiGAN: {n} activations exceeded the fp16 range; switching to DTTS_VOC_BF16X3 and redoing this call")
                self._guard_left = 0
                self._auto_guard = False
                self._build(self._state_dict, abi.VOC_BF16X3, False)
                self._state_dict = None
                self.ctx.hifigan_forward(mel.data_ptr(), lens.data_ptr() if lens is not None else None, B, T, wav.data_ptr(), stream)
            elif sampled:
                self.ctx.vocoder_range_guard(False)
            elif self._guard_left > 0:
                self._guard_left -= 1
                if self._guard_left == 0:
                    self.ctx.vocoder_range_guard(False)
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
        wav = self.forward_batch(batch.to(self.device), torch.tensor(lens, dtype=torch.int32)).cpu().numpy()
        return [wav[i, :lens[i] * self.hop] for i in range(len(mels))]
