"""Drop-in for the reference's FastSpeech FFT block stack ``modules.fastspeech.tts_modules.FFTBlocks`` at inference
(tts_modules.py:458-523; the ``FastspeechDecoder`` / encoder body of the fs2 / PortaSpeech tasks of the same repo,
SURVEY.md 8f-2): same constructor arguments, ``load_state_dict`` with the reference's key names, ``forward(x [B,T,C],
padding_mask=None) -> [B,T,C]``, every tensor op executed by libdicttts_hip.so (``dtts_fft_blocks_forward``).

The sinusoidal table is built on the host exactly as ``SinusoidalPositionalEmbedding.get_embedding`` builds it
(modules/commons/common_layers.py:110-127: it is not part of the state dict) and handed to the library as a device
tensor.  ``norm='ln'`` only; ``attn_mask`` / ``return_hiddens`` are not implemented (no caller of the inference path
uses them)."""
import math

import torch

from . import abi
from .hparams import BIAOBEI_DEFAULTS, fill_abi_config

DEFAULT_MAX_TARGET_POSITIONS = 2000   # tts_modules.py:13-14


def sinusoid_table(num_embeddings, embedding_dim, padding_idx=0):
    """the torch expressions of common_layers.py:110-127, so that the table is bit-identical to the reference's"""
    half_dim = embedding_dim // 2
    emb = math.log(10000) / (half_dim - 1)
    emb = torch.exp(torch.arange(half_dim, dtype=torch.float) * -emb)
    emb = torch.arange(num_embeddings, dtype=torch.float).unsqueeze(1) * emb.unsqueeze(0)
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=1).view(num_embeddings, -1)
    if embedding_dim % 2 == 1:
        emb = torch.cat([emb, torch.zeros(num_embeddings, 1)], dim=1)
    if padding_idx is not None:
        emb[padding_idx, :] = 0
    return emb


class FFTBlocks(torch.nn.Module):
    def __init__(self, hidden_size, num_layers, ffn_kernel_size=9, dropout=None, num_heads=2, use_pos_embed=True,
                 use_last_norm=True, norm="ln", use_pos_embed_alpha=True, hparams=None):
        super().__init__()
        if not torch.cuda.is_available():
            raise abi.DttsError("dict_tts_amd.fft.FFTBlocks needs a ROCm GPU: the HIP path has no CPU fallback")
        if norm != "ln":
            raise NotImplementedError("only norm='ln' FFT blocks are implemented")
        hp = {**BIAOBEI_DEFAULTS, **(hparams or {})}
        if hp.get("ffn_act", "gelu") != "gelu" or hp.get("ffn_padding", "SAME") != "SAME":
            raise NotImplementedError("only ffn_act=gelu / ffn_padding=SAME are implemented (egs/egs_bases/tts/base.yaml:73-74)")
        cfg = fill_abi_config(abi.default_config(), hp)
        cfg.hidden_size, cfg.num_heads = int(hidden_size), int(num_heads)
        cfg.fft_layers, cfg.fft_kernel_size = int(num_layers), int(ffn_kernel_size)
        cfg.fft_use_pos_embed, cfg.fft_use_last_norm = int(bool(use_pos_embed)), int(bool(use_last_norm))
        self.ctx = abi.Context(cfg)
        self.hidden_size, self.num_layers = int(hidden_size), int(num_layers)
        self.use_pos_embed, self.use_pos_embed_alpha = bool(use_pos_embed), bool(use_pos_embed_alpha)
        self.device = torch.device("cuda", torch.cuda.current_device())
        self._table = None
        self._state = {}
        self._ready = False

    def load_state_dict(self, state_dict, strict=True):
        used = {k: v for k, v in state_dict.items() if k != "embed_positions._float_tensor"}   # a dtype/device marker
        if not self.use_pos_embed_alpha:
            used.pop("pos_embed_alpha", None)
        self._state = dict(state_dict)
        self.ctx.load_state_dict("fft", used)
        try:
            self.ctx.finalize(abi.PART_FFT)          # names the first missing tensor
        except abi.DttsError as e:
            raise RuntimeError(f"Error(s) in loading state_dict for FFTBlocks: {e}") from e
        self._ready = True
        return torch.nn.modules.module._IncompatibleKeys([], [])

    def state_dict(self, *args, **kwargs):
        return dict(self._state)

    def forward(self, x, padding_mask=None, attn_mask=None, return_hiddens=False):
        if attn_mask is not None or return_hiddens:
            raise NotImplementedError("attn_mask / return_hiddens are not implemented")
        if not self._ready:
            raise RuntimeError("load_state_dict() must be called first")
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        B, T, C = x.shape
        assert C == self.hidden_size
        lens = None
        if padding_mask is not None:                 # [B,T], True = padding (must be a suffix, as every caller produces it)
            pm = padding_mask.to(self.device).bool()
            lens = (~pm).sum(1).to(torch.int32)
            assert bool((pm == (torch.arange(T, device=self.device)[None] >= lens[:, None])).all()), "padding must be a suffix"
        table, n_pos = None, 0
        if self.use_pos_embed:
            if self._table is None or self._table.shape[0] <= T:
                self._table = sinusoid_table(max(DEFAULT_MAX_TARGET_POSITIONS, T + 1), C, 0).to(self.device).contiguous()
            table, n_pos = self._table, self._table.shape[0]
        y = torch.empty_like(x)
        self.ctx.fft_blocks_forward(x.data_ptr(), lens.data_ptr() if lens is not None else None,
                                    table.data_ptr() if table is not None else None, n_pos, B, T, y.data_ptr(),
                                    torch.cuda.current_stream().cuda_stream)
        return y
