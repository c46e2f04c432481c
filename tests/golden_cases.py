"""Seeded input builders shared by oracle/make_golden.py (which feeds them to the REFERENCE implementation in
the build container) and by the parity tests (which feed them to the oracle and to the HIP path).

Everything is regenerated from the counter-based generator in dict_tts_amd/synth.py, so the committed
fixtures under tests/golden/ hold only the reference's OUTPUTS."""
import numpy as np

from dict_tts_amd import synth

SEED = 1234


def g1_inputs():
    """G1: one 4-layer Encoder; x [2,192,11], lengths (11, 7)"""
    x = synth.randn(SEED, "g1.x", (2, 192, 11))
    lengths = np.array([11, 7], np.int64)
    return x, lengths


def g2_inputs():
    """G2: S2PA attention, B=2, T_w=6, L_k=12, P=4: a two-sense word, a forced pronunciation
    (pron_modified != 0), a fully padded word row (all key_map == 0), BOS/last rows all 1."""
    B, T, L, P = 2, 6, 12, 4
    x = synth.randn(SEED, "g2.x", (B, 192, T))
    keys = synth.randn(SEED, "g2.keys", (B, T, L, 768), 0.5)
    values = synth.randn(SEED, "g2.values", (B, T, L, 768), 0.5)
    key_map = np.zeros((B, T, L), np.float32)
    pinyin = np.zeros((B, T, P), np.int64)
    pinyin_map = np.zeros((B, T, P), np.int64)
    pron_modified = np.zeros((B, T), np.int64)
    for b in range(B):
        for t in range(1, T - 1):
            if (b, t) == (1, 4):
                continue  # fully padded word row: every logit masked -> uniform softmax
            if (b + t) % 2 == 0:  # two senses: [0 1 1 1 0 | 0 2 2 2 2 0]
                key_map[b, t, :11] = [0, 1, 1, 1, 0, 0, 2, 2, 2, 2, 0]
                pinyin[b, t] = [3 + t, 40 + t, 7 + b, 90 + t]
                pinyin_map[b, t] = [1, 1, 2, 2]
            else:  # single sense
                key_map[b, t, :7] = [0, 1, 1, 1, 1, 1, 0]
                pinyin[b, t, :2] = [10 + t, 60 + b]
                pinyin_map[b, t, :2] = [1, 1]
    keys[key_map == 0] *= 0.0  # padded gloss tokens are zero vectors in the real data too
    values[key_map == 0] *= 0.0
    keys[:, 0] = 0
    keys[:, -1] = 0
    values[:, 0] = 0
    values[:, -1] = 0
    key_map[:, 0] = 1
    key_map[:, -1] = 1
    pinyin_map[:, 0] = 1
    pinyin_map[:, -1] = 1
    pron_modified[0, 2] = 2  # (0,2) has two senses: force sense 2
    pron_modified[1, 3] = 1  # (1,3) has two senses: force sense 1
    return x, keys, values, key_map, pinyin, pinyin_map, pron_modified


def g3_inputs():
    """G3: duration predictor input [3,9,192]; utterance 1 has 6 valid words, utterance 2 has 4 (padding rows
    are exactly zero, which is how the reference derives src_padding, model.py:73)."""
    x = synth.randn(SEED, "g3.x", (3, 9, 192), 0.8)
    x[1, 6:] = 0
    x[2, 4:] = 0
    return x


def g3_int_durations():
    """integer durations for the length regulator alone: an all-zero row (tts_modules.py:248-250), zeros in
    the middle, a long word"""
    dur = np.array([[2, 0, 3, 1, 0, 4, 0, 0],
                    [0, 0, 0, 0, 0, 0, 0, 0],
                    [1, 1, 1, 1, 1, 1, 1, 9],
                    [0, 5, 0, 0, 0, 0, 0, 0]], np.int64)
    ilens = np.array([8, 5, 8, 3], np.int64)
    return dur, ilens


def g4_inputs():
    """G4: FVAE decode; g [2,192,24], second item valid for 16 frames only (unmasked-padding behaviour)"""
    g = synth.randn(SEED, "g4.g", (2, 192, 24), 0.7)
    g[1, :, 16:] = 0
    z = synth.noise(SEED, 2, 6, "g4.z")
    return g, z


def g5_sentences():
    st = synth.biaobei_struct()
    return st["sentences"][:3]


def g5_batch(which):
    """which: 0,1,2 -> that sentence alone (B=1); 'all' -> the three as one batch"""
    s = g5_sentences()
    return synth.make_batch(s if which == "all" else [s[which]], SEED, pron_every=2)


def g5_noise(B, T4, which):
    return synth.noise(SEED, B, T4, f"g5.z.{which}")


def g6_mel():
    return synth.random_mel(SEED, 32, "g6.mel")


def g8_inputs(which):
    """FFTBlocks cases.  'dec': hidden 192, 4 layers, k=9, positional embedding, last norm, three
    utterances of 45 / 30 / 1 frames in a 45-frame batch, one VALID frame whose first channel is exactly 0 (the
    reference's make_positions(x[..., 0]) treats it as padding: it gets no position and does not advance the count).
    'enc': 2 layers, k=5, no positional embedding (how FastspeechEncoder runs the stack), lengths 20 / 13."""
    from dict_tts_amd import synth
    if which == "dec":
        x = synth.randn(SEED, "g8.dec.x", (3, 45, 192), 1.0)
        lens = np.array([45, 30, 1], np.int64)
        x[0, 7, 0] = 0.0
    else:
        x = synth.randn(SEED, "g8.enc.x", (2, 20, 192), 1.0)
        lens = np.array([20, 13], np.int64)
    for b, n in enumerate(lens):
        x[b, n:] = 0.0
    return x, lens


G8_CASES = {"dec": dict(layers=4, kernel_size=9, use_pos_embed=True, use_last_norm=True),
            "enc": dict(layers=2, kernel_size=5, use_pos_embed=False, use_last_norm=True)}


def g9_wavs():
    """G9 (output side, utils/audio.py:11-16): three float32 waveforms in (-1, 1) like the vocoder's tanh output, ragged
    lengths (frames 3 / 7 / 1, hop 256), values scaled so that w * 32767 has fractions on both sides of .5, one exact 0
    and one sample pair at +-0.999969 (the int16 edge)."""
    lens = [3, 7, 1]
    out = []
    for i, n in enumerate(lens):
        w = np.tanh(synth.randn(SEED, f"g9.wav{i}", (n * 256,), 0.6)).astype(np.float32)
        w[0] = 0.0
        w[1], w[2] = np.float32(0.999969), np.float32(-0.999969)
        out.append(w)
    return lens, out


G10_WORDS = None


def g10_entries():
    """G10 (dict_embed fixture, utils/indexed_datasets.py:41-54 + binarizer_zh.py:250-259,301-309): four ``dict_embed`` items in
    the reference's item layout — an absent character (zero entry), a one-sense word, a two-sense heteronym and a
    three-sense one (gloss lengths cut to <= 8 tokens to keep the fixture small) — plus the pinyin encoder list."""
    st = synth.biaobei_struct()
    by_senses = {}
    for w, senses in sorted(st["entries"].items()):
        n = 0 if senses[0][2] < 0 else len(senses)
        by_senses.setdefault(n, w)
    words = [2, by_senses[1], by_senses[2], by_senses[3]]   # 2 = '<UNK>': what the binarizer writes for a word outside zh-dict.json
    pinyin_encoder = ["<UNK>"] + [f"py{i}" for i in range(1, 185)]
    items = []
    for w in words:
        senses = st["entries"].get(w, [[3, 0, -1]])
        if senses[0][2] < 0:
            items.append({"tokens_gloss": ["O"], "key": np.zeros((3, 768), np.float32), "key_map": [0, 1, 0],
                          "pinyin": ["<UNK>"], "pinyin_map": [1]})
            continue
        short = {w: [[min(n, 8), a, b] for n, a, b in senses]}
        emb, km, py, pm = synth.dict_entry(w, SEED, short)
        items.append({"tokens_gloss": ["<sos>"] + ["t"] * (emb.shape[0] - 2) + ["<eos>"], "key": emb,
                      "key_map": [int(v) for v in km], "pinyin": [pinyin_encoder[int(i)] for i in py],
                      "pinyin_map": [int(v) for v in pm]})
    return words, items, pinyin_encoder
