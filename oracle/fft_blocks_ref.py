"""TEST INFRASTRUCTURE ONLY (imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product).

CPU fp32 restatement of the reference's FastSpeech FFT block stack at inference:
``FFTBlocks.forward`` (modules/fastspeech/tts_modules.py:458-523) over ``TransformerEncoderLayer`` -> ``EncSALayer``
(tts_modules.py:17-32, modules/commons/common_layers.py:624-673), fairseq-style ``MultiheadAttention`` on the torch
fast path (common_layers.py:171-290: ``F.multi_head_attention_forward``, bias-free in/out projections),
``TransformerFFNLayer`` (common_layers.py:541-581: Conv1d k SAME, * k**-0.5, GELU, Linear) and
``SinusoidalPositionalEmbedding`` (common_layers.py:93-148) with ``make_positions`` (utils/tts_utils.py:6-18).
Pinned by tests/golden/g8_fft_blocks.npz, generated from the reference itself by oracle/make_golden.py.
"""
import math

import torch
import torch.nn.functional as F


def sinusoid_table(num_embeddings, embedding_dim, padding_idx=0):
    """common_layers.py:110-127 (the tensor2tensor flavour: sin block then cos block; row padding_idx zeroed)"""
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


def make_positions(tensor, padding_idx=0):
    """utils/tts_utils.py:6-18: non-padding symbols -> 1, 2, 3, ...; padding -> padding_idx.  FFTBlocks passes
    x[..., 0] (tts_modules.py:505), so a frame whose FIRST channel is exactly 0 counts as padding here."""
    mask = tensor.ne(padding_idx).int()
    return (torch.cumsum(mask, dim=1).type_as(mask) * mask).long() + padding_idx


def enc_sa_layer(sd, p, x, padding_mask, num_heads, kernel_size):
    """EncSALayer.forward, x [B,T,C] (the reference works on [T,B,C]; per-token math is layout independent).
    common_layers.py:649-673"""
    B, T, C = x.shape
    keep = (1.0 - padding_mask.float())[..., None]                                     # [B,T,1]
    residual = x
    h = F.layer_norm(x, (C,), sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"], 1e-5)
    # F.multi_head_attention_forward with in_proj_bias = out_proj.bias = None (bias=False at :637)
    qkv = F.linear(h, sd[p + "self_attn.in_proj_weight"])
    q, k, v = qkv.split(C, dim=-1)
    dk = C // num_heads
    q = q.view(B, T, num_heads, dk).transpose(1, 2) * dk ** -0.5
    k = k.view(B, T, num_heads, dk).transpose(1, 2)
    v = v.view(B, T, num_heads, dk).transpose(1, 2)
    scores = q @ k.transpose(-1, -2)                                                   # [B,h,T,T]
    scores = scores.masked_fill(padding_mask[:, None, None, :], float("-inf"))         # key_padding_mask
    o = (torch.softmax(scores, dim=-1) @ v).transpose(1, 2).reshape(B, T, C)
    h = F.linear(o, sd[p + "self_attn.out_proj.weight"])
    x = (residual + h) * keep
    residual = x
    h = F.layer_norm(x, (C,), sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"], 1e-5)
    # TransformerFFNLayer: the LayerNorm output at padded frames is its bias, and the SAME-padded conv sees it
    h = F.conv1d(h.transpose(1, 2), sd[p + "ffn.ffn_1.weight"], sd[p + "ffn.ffn_1.bias"], padding=kernel_size // 2).transpose(1, 2)
    h = F.gelu(h * kernel_size ** -0.5)
    h = F.linear(h, sd[p + "ffn.ffn_2.weight"], sd[p + "ffn.ffn_2.bias"])
    return (residual + h) * keep


def fft_blocks(sd, x, padding_mask=None, num_heads=2, kernel_size=9, use_pos_embed=True, use_last_norm=True, prefix=""):
    """FFTBlocks.forward(x [B,T,C]) -> [B,T,C]   (tts_modules.py:495-523)"""
    B, T, C = x.shape
    if padding_mask is None:
        padding_mask = x.abs().sum(-1).eq(0)
    keep = (1.0 - padding_mask.float())[..., None]
    if use_pos_embed:
        table = sinusoid_table(max(2000, T + 1), C, 0)                                 # init_size = DEFAULT_MAX_TARGET_POSITIONS
        alpha = sd.get(prefix + "pos_embed_alpha", torch.ones(1))
        x = x + alpha * table[make_positions(x[..., 0], 0)]
    x = x * keep
    n_layers = 1 + max(int(k[len(prefix):].split(".")[1]) for k in sd if k.startswith(prefix + "layers."))
    for i in range(n_layers):
        x = enc_sa_layer(sd, f"{prefix}layers.{i}.op.", x, padding_mask, num_heads, kernel_size) * keep
    if use_last_norm:
        x = F.layer_norm(x, (C,), sd[prefix + "layer_norm.weight"], sd[prefix + "layer_norm.bias"], 1e-5) * keep
    return x
