"""Deterministic synthetic weights and inputs for the Dict-TTS inference path.

No trained checkpoint, Biaobei audio or roformer gloss embeddings exist offline, so tests,
``__graft_entry__.smoke()`` and ``bench.py`` run on tensors produced here: a counter-based generator (numpy
Philox keyed by (seed, crc32(tensor name))) so that any tensor can be regenerated independently, in any
order, on any machine.  The *shapes and key names* are exactly those of the reference state dicts
(``PortaSpeech_dict`` — modules/dict_tts/model.py:14-33 — and ``HifiGanGenerator`` —
modules/hifigan/hifigan.py:100-122), including the weight-norm ``weight_g/weight_v`` pairs and the tensors
that are loaded but unused at inference (SURVEY.md §8a "Parameter inventory").

The *structure* of the batches (sentences, senses per character, gloss lengths, pinyin ids) comes from
``data/biaobei_struct.json`` (derived from the reference's data files by oracle/make_biaobei_struct.py);
batch collation follows tasks/tts/dataset_utils.py:264-330.
"""
import json
import os
import zlib

import numpy as np

HIDDEN = 192
GLOSS_DIM = 768
N_MEL = 80
N_PINYIN = 185
WORD_SIZE = 8000

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def _rng(seed, name):
    return np.random.Generator(np.random.Philox(key=[int(seed) & 0xFFFFFFFFFFFFFFFF, zlib.crc32(name.encode())]))


def randn(seed, name, shape, scale=1.0):
    return (_rng(seed, name).standard_normal(size=shape, dtype=np.float32) * np.float32(scale)).astype(np.float32)


# ----------------------------------------------------------------------------------------------------------
# state dicts
# ----------------------------------------------------------------------------------------------------------
def _conv(sd, seed, name, cout, cin, k, gain=1.0, bias=0.05, wn=False, transposed=False):
    shape = (cin, cout, k) if transposed else (cout, cin, k)
    fan_in = cin * k if not transposed else cin * max(1, k // 2)
    w = randn(seed, name + ".w", shape, gain / np.sqrt(fan_in))
    if wn:
        # weight_norm(dim=0): g has shape [shape[0],1,1]; make g != ||v|| so that folding is exercised
        nrm = np.sqrt((w.reshape(shape[0], -1) ** 2).sum(1)).reshape(shape[0], 1, 1)
        sd[name + ".weight_g"] = (nrm * (1.0 + 0.1 * randn(seed, name + ".g", (shape[0], 1, 1)))).astype(np.float32)
        sd[name + ".weight_v"] = (w * np.float32(1.7)).astype(np.float32)
    else:
        sd[name + ".weight"] = w
    sd[name + ".bias"] = randn(seed, name + ".b", (cout,), bias)


def _linear(sd, seed, name, cout, cin, gain=1.0, bias=None):
    sd[name + ".weight"] = randn(seed, name + ".w", (cout, cin), gain / np.sqrt(cin))
    if bias is not None:
        sd[name + ".bias"] = randn(seed, name + ".b", (cout,), bias)


def _ln(sd, seed, name, c, g="gamma", b="beta"):
    sd[f"{name}.{g}"] = (1.0 + 0.1 * randn(seed, name + ".g", (c,))).astype(np.float32)
    sd[f"{name}.{b}"] = randn(seed, name + ".b", (c,), 0.1)


def _rel_encoder(sd, seed, p, layers=4, h=HIDDEN, f=4 * HIDDEN, k=5):
    for i in range(layers):
        for n in "qkvo":
            _conv(sd, seed, f"{p}.attn_layers.{i}.conv_{n}", h, h, 1, gain=1.0 if n in "qk" else 0.7)
        _ln(sd, seed, f"{p}.norm_layers_1.{i}", h)
        _conv(sd, seed, f"{p}.ffn_layers.{i}.conv_1", f, h, k, gain=1.2)
        _conv(sd, seed, f"{p}.ffn_layers.{i}.conv_2", h, f, 1, gain=0.7)
        _ln(sd, seed, f"{p}.norm_layers_2.{i}", h)
    _ln(sd, seed, f"{p}.last_ln", h)


def _wn(sd, seed, p, hidden, k, layers, gin):
    for i in range(layers):
        _conv(sd, seed, f"{p}.in_layers.{i}", 2 * hidden, hidden, k, gain=1.0, wn=True)
        rs = 2 * hidden if i < layers - 1 else hidden
        _conv(sd, seed, f"{p}.res_skip_layers.{i}", rs, hidden, 1, gain=0.8, wn=True)
    _conv(sd, seed, f"{p}.cond_layer", 2 * hidden * layers, gin, 1, gain=0.7, wn=True)


def dict_tts_state_dict(seed=1234, n_phone=6, word_size=WORD_SIZE):
    """numpy state dict with the key names / shapes of ``state_dict['model']``."""
    sd = {}
    h = HIDDEN
    # PortaSpeech leftovers: loaded, never used by PortaSpeech_dict (modules/portaspeech/model.py:153-158)
    for n in ("enc_pos_proj", "dec_query_proj", "dec_res_proj"):
        _linear(sd, seed, n, h, 2 * h, bias=0.02)
    sd["attn.in_proj_weight"] = randn(seed, "attn.in_proj_weight", (3 * h, h), h ** -0.5)
    _linear(sd, seed, "attn.out_proj", h, h)
    # duration predictor (modules/portaspeech/model.py:38-66)
    for i in range(3):
        _conv(sd, seed, f"dur_predictor.conv.{i}.1", 128, h if i == 0 else 128, 5, gain=1.3)
        _ln(sd, seed, f"dur_predictor.conv.{i}.3", 128, "weight", "bias")
    sd["dur_predictor.linear.0.weight"] = randn(seed, "dur.lin.w", (1, 128), 0.6 / np.sqrt(128))
    sd["dur_predictor.linear.0.bias"] = np.array([2.2], np.float32)  # ~ exp(2.3)-1 = 9 frames / word
    # FVAE (modules/dict_tts/fvae_semantics.py:61-82)
    _conv(sd, seed, "fvae.g_pre_net.0", h, h, 8, gain=1.0)
    _conv(sd, seed, "fvae.encoder.pre_net.0", h, N_MEL, 8)
    _wn(sd, seed, "fvae.encoder.wn", h, 5, 8, h)
    _conv(sd, seed, "fvae.encoder.out_proj", 32, h, 1)
    for f in (0, 2, 4, 6):
        p = f"fvae.prior_flow.flows.{f}"
        _conv(sd, seed, p + ".pre", 64, 8, 1, gain=1.0)
        _wn(sd, seed, p + ".enc", 64, 3, 4, h)
        _conv(sd, seed, p + ".post", 8, 64, 1, gain=0.5)
    _conv(sd, seed, "fvae.decoder.pre_net.0", h, 16, 4, gain=1.0, transposed=True)
    # ConvTranspose1d keeps a [cout] bias
    _wn(sd, seed, "fvae.decoder.wn", h, 5, 4, h)
    _conv(sd, seed, "fvae.decoder.out_proj", N_MEL, h, 1, gain=1.5)
    sd["fvae.decoder.out_proj.bias"] = (sd["fvae.decoder.out_proj.bias"] - 2.5).astype(np.float32)  # log-mel range
    # dictionary encoder (modules/dict_tts/layers/dict_encoder.py:69-128)
    p = "dict_encoder.S2PA_module"
    sd[p + ".emb.weight"] = randn(seed, p + ".emb", (n_phone, h), h ** -0.5)
    sd[p + ".word_emb.weight"] = randn(seed, p + ".word_emb", (word_size, h), h ** -0.5)
    sd[p + ".emb.weight"][0] = 0
    sd[p + ".word_emb.weight"][0] = 0
    _rel_encoder(sd, seed, p + ".semantic_encoder")
    a = p + ".s2pa_attention"
    _linear(sd, seed, a + ".q_transform", h, h, gain=6.0)
    _linear(sd, seed, a + ".k_transform", h, GLOSS_DIM, gain=6.0)
    _linear(sd, seed, a + ".v_transform", h, GLOSS_DIM, gain=2.0)
    _linear(sd, seed, a + ".output_transform", h, h)
    sd[a + ".pinyin_embedding.weight"] = randn(seed, a + ".pinyin", (N_PINYIN, h), 1.0)
    sd[a + ".pinyin_embedding.weight"][0] = 0
    _rel_encoder(sd, seed, p + ".linguistic_encoder")
    return sd


def fft_blocks_state_dict(seed=1234, hidden=HIDDEN, layers=4, kernel_size=9, use_pos_embed=True, use_last_norm=True):
    """numpy state dict of the reference's ``FFTBlocks`` (modules/fastspeech/tts_modules.py:458-493): per layer
    ``layers.i.op.{layer_norm1, self_attn.in_proj_weight, self_attn.out_proj, layer_norm2, ffn.ffn_1, ffn.ffn_2}``"""
    sd = {}
    for i in range(layers):
        p = f"layers.{i}.op"
        _ln(sd, seed, f"fft.{p}.layer_norm1", hidden, "weight", "bias")
        sd[f"{p}.layer_norm1.weight"], sd[f"{p}.layer_norm1.bias"] = sd.pop(f"fft.{p}.layer_norm1.weight"), sd.pop(f"fft.{p}.layer_norm1.bias")
        sd[f"{p}.self_attn.in_proj_weight"] = randn(seed, f"fft.{p}.in_proj", (3 * hidden, hidden), 1.0 / np.sqrt(hidden))
        sd[f"{p}.self_attn.out_proj.weight"] = randn(seed, f"fft.{p}.out_proj", (hidden, hidden), 0.7 / np.sqrt(hidden))
        _ln(sd, seed, f"fft.{p}.layer_norm2", hidden, "weight", "bias")
        sd[f"{p}.layer_norm2.weight"], sd[f"{p}.layer_norm2.bias"] = sd.pop(f"fft.{p}.layer_norm2.weight"), sd.pop(f"fft.{p}.layer_norm2.bias")
        sd[f"{p}.ffn.ffn_1.weight"] = randn(seed, f"fft.{p}.ffn_1.w", (4 * hidden, hidden, kernel_size), 1.6 / np.sqrt(hidden * kernel_size) * np.sqrt(kernel_size))
        sd[f"{p}.ffn.ffn_1.bias"] = randn(seed, f"fft.{p}.ffn_1.b", (4 * hidden,), 0.05)
        sd[f"{p}.ffn.ffn_2.weight"] = randn(seed, f"fft.{p}.ffn_2.w", (hidden, 4 * hidden), 0.7 / np.sqrt(4 * hidden))
        sd[f"{p}.ffn.ffn_2.bias"] = randn(seed, f"fft.{p}.ffn_2.b", (hidden,), 0.05)
    if use_last_norm:
        _ln(sd, seed, "fft.layer_norm", hidden, "weight", "bias")
        sd["layer_norm.weight"], sd["layer_norm.bias"] = sd.pop("fft.layer_norm.weight"), sd.pop("fft.layer_norm.bias")
    if use_pos_embed:
        sd["pos_embed_alpha"] = np.array([0.8], np.float32)
        sd["embed_positions._float_tensor"] = np.zeros(1, np.float32)
    return sd


def hifigan_config():
    """egs/egs_bases/tts/vocoder/hifigan.yaml:3-10"""
    return {"resblock": "1", "upsample_rates": [8, 8, 2, 2], "upsample_kernel_sizes": [16, 16, 4, 4],
            "upsample_initial_channel": 512, "resblock_kernel_sizes": [3, 7, 11],
            "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]]}


def hifigan_state_dict(seed=1234, weight_norm=True, cfg=None):
    """numpy state dict of ``HifiGanGenerator`` (``state_dict.model_gen``), weight-norm form by default."""
    cfg = cfg or hifigan_config()
    sd = {}
    c0 = cfg["upsample_initial_channel"]
    _conv(sd, seed, "conv_pre", c0, N_MEL, 7, gain=0.35, wn=weight_norm)
    ch = c0
    for i, (u, k) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
        ch = c0 // (2 ** (i + 1))
        _conv(sd, seed, f"ups.{i}", ch, 2 * ch, k, gain=1.4, wn=weight_norm, transposed=True)
        # fan-in of a stride-u transposed conv is cin*k/u
        for key in (f"ups.{i}.weight_v", f"ups.{i}.weight"):
            if key in sd:
                sd[key] = (sd[key] * np.float32(np.sqrt(max(1, k // 2) / (k / u)))).astype(np.float32)
        if weight_norm:
            v = sd[f"ups.{i}.weight_v"]
            nrm = np.sqrt((v.reshape(v.shape[0], -1) ** 2).sum(1)).reshape(-1, 1, 1)
            sd[f"ups.{i}.weight_g"] = (nrm / 1.7 * (1.0 + 0.1 * randn(seed, f"ups.{i}.g", (v.shape[0], 1, 1)))
                                        ).astype(np.float32)
        for j, kk in enumerate(cfg["resblock_kernel_sizes"]):
            r = f"resblocks.{i * 3 + j}"
            for m in range(3):
                _conv(sd, seed, f"{r}.convs1.{m}", ch, ch, kk, gain=1.0, wn=weight_norm)
                _conv(sd, seed, f"{r}.convs2.{m}", ch, ch, kk, gain=0.6, wn=weight_norm)
    _conv(sd, seed, "conv_post", 1, ch, 7, gain=0.25, wn=weight_norm)
    return sd


# ----------------------------------------------------------------------------------------------------------
# inputs
# ----------------------------------------------------------------------------------------------------------
_STRUCT = None


def biaobei_struct():
    global _STRUCT
    if _STRUCT is None:
        with open(os.path.join(_DATA, "biaobei_struct.json")) as f:
            d = json.load(f)
        d["entries"] = {int(k): v for k, v in d["entries"].items()}
        _STRUCT = d
    return _STRUCT


_FULL = None


def zh_dict_struct():
    """sense structure of ALL 7,030 zh-dict.json characters (oracle/make_biaobei_struct.py:write_full_dictionary):
    {'n_entries', 'entries': word id (3 + rank) -> [[gloss_tokens, pinyin_initial_id, pinyin_final_id], ...]} — the table
    of BASELINE.json configs[4] ("word_size=8000, full zh-dict.json entry set in HBM")"""
    global _FULL
    if _FULL is None:
        with open(os.path.join(_DATA, "zh_dict_struct.json")) as f:
            d = json.load(f)
        d["entries"] = {int(k): v for k, v in d["entries"].items()}
        _FULL = d
    return _FULL


BOS_ID, EOS_ID = 1203, 1204  # any ids >= 3 outside the char range; '<BOS>'/'<EOS>' are ordinary vocabulary words


def config5_sentences(n=128, seed=55, entries=None):
    """BASELINE.json configs[4] / SURVEY §8d 'Config 5': n mixed-length utterances, T_w ~ U{6..60} characters drawn from the WHOLE
    dictionary with the heteronyms (>= 2 pronunciations) over-sampled x5; the first / last table rows appear.  -> list of id lists"""
    ent = entries if entries is not None else zh_dict_struct()["entries"]
    rng = np.random.default_rng(seed)
    ids = np.array(sorted(ent))
    wts = np.array([5.0 if len(ent[i]) > 1 else 1.0 for i in ids])
    wts /= wts.sum()
    sents = [rng.choice(ids, size=int(rng.integers(6, 61)), p=wts).tolist() for _ in range(n)]
    sents[0][0], sents[1][-1] = int(ids[0]), int(ids[-1])
    return sents


def dict_entry(word_id, seed=1234, entries=None):
    """One ``dict_embed`` item (binarizer_zh.py:301-309): key/value [L,768] (same content), key_map [L],
    pinyin [P], pinyin_map [P]."""
    entries = biaobei_struct()["entries"] if entries is None else entries
    senses = entries[word_id]
    if senses[0][2] < 0:  # char absent from the dictionary: zero entry (binarizer_zh.py:250-259)
        return (np.zeros((3, GLOSS_DIM), np.float32), np.array([0, 1, 0], np.float32),
                np.array([0], np.int64), np.array([1], np.int64))
    L = sum(s[0] for s in senses)
    emb = randn(seed, f"gloss.{word_id}", (L, GLOSS_DIM), 0.5)
    key_map, pinyin, pinyin_map = [], [], []
    for i, (n, ini, fin) in enumerate(senses):
        key_map += [0] + [i + 1] * (n - 2) + [0]
        pinyin += [ini, fin]
        pinyin_map += [i + 1, i + 1]
    return emb, np.array(key_map, np.float32), np.array(pinyin, np.int64), np.array(pinyin_map, np.int64)


def make_batch(sentences, seed=1234, entries=None, pron_every=7):
    """Collate sentences (lists of word ids, without BOS/EOS) the way DictTTSDataset.collater does
    (tasks/tts/dataset_utils.py:264-302).  Returns a dict of numpy arrays:
    word_tokens [B,T_w] i64, keys/values [B,T_w,L_k,768] f32, key_map [B,T_w,L_k] f32, pinyin [B,T_w,P] i64,
    pinyin_map [B,T_w,P] i64, pron_modified [B,T_w] i64."""
    B = len(sentences)
    items = [[dict_entry(w, seed, entries) for w in s] for s in sentences]
    Tw = max(len(s) for s in sentences) + 2
    Lk = max(e[0].shape[0] for it in items for e in it)
    P = max(e[2].shape[0] for it in items for e in it)
    word_tokens = np.zeros((B, Tw), np.int64)
    keys = np.zeros((B, Tw, Lk, GLOSS_DIM), np.float32)
    key_map = np.zeros((B, Tw, Lk), np.float32)
    pinyin = np.zeros((B, Tw, P), np.int64)
    pinyin_map = np.zeros((B, Tw, P), np.int64)
    pron_modified = np.zeros((B, Tw), np.int64)
    n_multi = 0
    for b, (s, it) in enumerate(zip(sentences, items)):
        word_tokens[b, :len(s) + 2] = [BOS_ID] + list(s) + [EOS_ID]
        for t, (emb, km, py, pm) in enumerate(it):
            keys[b, t + 1, :emb.shape[0]] = emb
            key_map[b, t + 1, :km.shape[0]] = km
            pinyin[b, t + 1, :py.shape[0]] = py
            pinyin_map[b, t + 1, :pm.shape[0]] = pm
            if pm.max() >= 2:
                n_multi += 1
                if pron_every and n_multi % pron_every == 0:
                    pron_modified[b, t + 1] = 2  # sandhi-forced sense (sandhi_processor.py:447-483)
    # F.pad(..., value=1) on the word axis: first and LAST row of the padded batch (dataset_utils.py:288-300)
    key_map[:, 0, :] = 1
    key_map[:, -1, :] = 1
    pinyin_map[:, 0, :] = 1
    pinyin_map[:, -1, :] = 1
    return {"word_tokens": word_tokens, "keys": keys, "values": keys.copy(), "key_map": key_map, "pinyin": pinyin,
            "pinyin_map": pinyin_map, "pron_modified": pron_modified}


def biaobei_batch(first=0, count=60, seed=1234):
    st = biaobei_struct()
    return make_batch(st["sentences"][first:first + count], seed)


def teacher_mel2word(word_tokens, frames_per_char=22, frames_edge=11):
    """Teacher-forced mel2word: 22 frames / char, 11 for BOS/EOS; [B,T] i64, 0 = padding."""
    rows = []
    for wt in word_tokens:
        n = int((wt > 0).sum())
        d = [frames_edge] + [frames_per_char] * (n - 2) + [frames_edge]
        rows.append(np.repeat(np.arange(1, n + 1), d))
    T = max(len(r) for r in rows)
    out = np.zeros((len(rows), T), np.int64)
    for b, r in enumerate(rows):
        out[b, :len(r)] = r
    return out


def noise(seed, B, T4, name="z_p"):
    """prior sample z_p [B,16,T_mel/4] (fvae_semantics.py:110-111), injected explicitly."""
    return randn(seed, name, (B, 16, T4))


def random_mel(seed, T, name="mel"):
    """a log-mel-like [T,80] array in roughly [-6, 1.5] (base.yaml:59-60)"""
    m = randn(seed, name, (T, N_MEL), 1.2) - 3.0
    # smooth along time so that it resembles a spectrogram rather than white noise
    m[1:] = 0.6 * m[1:] + 0.4 * m[:-1]
    return np.clip(m, -6.0, 1.5).astype(np.float32)


# ----------------------------------------------------------------------------------------------------------
# resident dictionary table: the ragged form of the reference's ``dict_embed`` indexed dataset
# ----------------------------------------------------------------------------------------------------------
def dict_table(seed=1234, entries=None):
    """-> dict(ids, tok_off, keys, key_map, pin_off, pinyin, pinyin_map, L, P): every entry of the structure file, in
    ascending word-id order; ``ids[word_id]`` is the table row of a word."""
    entries = biaobei_struct()["entries"] if entries is None else entries
    order = sorted(entries)
    ids = {w: i for i, w in enumerate(order)}
    items = [dict_entry(w, seed, entries) for w in order]
    tok_off = np.zeros(len(order) + 1, np.int32)
    pin_off = np.zeros(len(order) + 1, np.int32)
    for i, (emb, km, py, pm) in enumerate(items):
        tok_off[i + 1] = tok_off[i] + emb.shape[0]
        pin_off[i + 1] = pin_off[i] + py.shape[0]
    return {"ids": ids, "tok_off": tok_off, "pin_off": pin_off,
            "keys": np.concatenate([it[0] for it in items]), "key_map": np.concatenate([it[1] for it in items]),
            "pinyin": np.concatenate([it[2] for it in items]), "pinyin_map": np.concatenate([it[3] for it in items]),
            "L": np.diff(tok_off), "P": np.diff(pin_off)}


def make_id_batch(sentences, table, pron_every=7):
    """the id-only form of make_batch(): word_tokens [B,T_w] i64, entry_ids [B,T_w] i32 (-1 BOS / last row, -2 batch
    padding), pron_modified [B,T_w] i64 and the batch maxima L_k, P the collated tensors would have"""
    B = len(sentences)
    Tw = max(len(s) for s in sentences) + 2
    word_tokens = np.zeros((B, Tw), np.int64)
    entry = np.full((B, Tw), -2, np.int32)
    pron_modified = np.zeros((B, Tw), np.int64)
    L_k, P, n_multi = 1, 1, 0
    for b, s in enumerate(sentences):
        word_tokens[b, :len(s) + 2] = [BOS_ID] + list(s) + [EOS_ID]
        for t, w in enumerate(s):
            e = table["ids"][w]
            entry[b, t + 1] = e
            L_k, P = max(L_k, int(table["L"][e])), max(P, int(table["P"][e]))
            if table["pinyin_map"][table["pin_off"][e]:table["pin_off"][e + 1]].max() >= 2:
                n_multi += 1
                if pron_every and n_multi % pron_every == 0:
                    pron_modified[b, t + 1] = 2
    entry[:, 0] = -1
    entry[:, -1] = -1
    return {"word_tokens": word_tokens, "entry_ids": entry, "pron_modified": pron_modified, "L_k": L_k, "P": P}
