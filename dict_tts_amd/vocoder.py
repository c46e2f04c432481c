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

    def __init__(self, state_dict=None, config=None, precision=None, ctx=None, unfused=False):
        if state_dict is None:
            config, state_dict = find_vocoder_checkpoint(hparams_mod.hparams["vocoder_ckpt"])   # looked up at call time: the
            # INTEGRATION.md hook may rebind dict_tts_amd.hparams.hparams after this module was imported
        self.config = {**HIFIGAN_DEFAULTS, **(config or {})}
        if not torch.cuda.is_available():
            raise abi.DttsError("dict_tts_amd.vocoder.HifiGAN needs a ROCm GPU: the HIP path has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device())
        if precision is None:
            precision = abi.VOC_PRECISIONS[os.environ.get("DTTS_VOCODER_PRECISION", "f16")]   # f16 = the waveform-exact default
        elif isinstance(precision, str):
            precision = abi.VOC_PRECISIONS[precision]
        self.precision = precision
        if ctx is None:
            cfg = fill_abi_config(abi.default_config(), None, self.config, vocoder_precision=precision)
            cfg.vocoder_unfused = 1 if unfused else 0   # testing aid (bf16 mode): one kernel per convolution
            ctx = abi.Context(cfg)
        self.ctx = ctx
        self.ctx.load_state_dict("vocoder", state_dict)
        self.ctx.finalize(abi.PART_VOCODER)
        self.hop = self.ctx.hop()

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
        self.ctx.hifigan_forward(mel.data_ptr(), lens.data_ptr() if lens is not None else None, B, T, wav.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream)
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
