"""Pin oracle/ (our CPU restatement) against outputs of the REFERENCE implementation itself
(tests/golden/*.npz, produced by oracle/make_golden.py in the build container).  CPU only."""
import os

import numpy as np
import pytest
import torch

import golden_cases as gc
from dict_tts_amd import synth
from oracle import dict_tts_ref as ref
from oracle import hifigan_ref as href

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))


@pytest.fixture(scope="module")
def sd():
    raw = {k: T(v) for k, v in synth.dict_tts_state_dict(gc.SEED, n_phone=6).items()}
    return href.fold_weight_norm(raw)


def _close(a, b, tol):
    a = a.numpy() if isinstance(a, torch.Tensor) else a
    assert a.shape == b.shape, (a.shape, b.shape)
    err = float(np.abs(a - b).max())
    assert err <= tol, err


def test_g1_encoder(sd, golden_dir):
    g = np.load(os.path.join(golden_dir, "g1_encoder.npz"))
    x, lengths = gc.g1_inputs()
    x_mask = (torch.arange(x.shape[2])[None] < T(lengths)[:, None]).unsqueeze(1).float()
    y = ref.rel_encoder(sd, "dict_encoder.S2PA_module.semantic_encoder", T(x), x_mask)
    _close(y, g["out"], 2e-6)


def test_g2_s2pa(sd, golden_dir):
    g = np.load(os.path.join(golden_dir, "g2_s2pa.npz"))
    x, keys, values, key_map, pinyin, pinyin_map, pron_modified = gc.g2_inputs()
    c, a, p, w = ref.s2pa_attention(sd, "dict_encoder.S2PA_module.s2pa_attention", T(x), T(keys), T(values),
                                    T(key_map), T(pinyin), T(pinyin_map), T(pron_modified))
    _close(c, g["context"], 2e-6)
    _close(a, g["dict_attn"], 1e-6)
    _close(p, g["pron"], 2e-6)
    _close(w, g["pron_attn"], 1e-6)
    # the fully padded word row attends uniformly; the forced rows are one-hot over their sense's tokens
    assert np.allclose(g["dict_attn"][1, 0, :, 4], 1.0 / 12, atol=1e-7)
    assert np.allclose(g["pron_attn"][0, 2], [0, 0, 1, 1], atol=1e-6)
    assert np.allclose(g["pron_attn"][1, 3], [1, 1, 0, 0], atol=1e-6)


def test_g3_duration(sd, golden_dir):
    g = np.load(os.path.join(golden_dir, "g3_duration.npz"))
    xin = T(gc.g3_inputs())
    dur, mel2word = ref.add_dur(sd, xin, None)
    _close(dur, g["dur"], 2e-6)
    assert np.array_equal(mel2word.numpy(), g["mel2word"])
    d, il = gc.g3_int_durations()
    assert np.array_equal(ref.length_regulator(T(d), T(il)).numpy(), g["mel2word_int"])
    # the all-zero row was filled with ones (tts_modules.py:248-250)
    assert g["mel2word_int"][1].tolist()[:6] == [1, 2, 3, 4, 5, 0]


def test_g4_fvae(sd, golden_dir):
    g = np.load(os.path.join(golden_dir, "g4_fvae.npz"))
    gg, z = gc.g4_inputs()
    mel, z_out = ref.fvae_infer(sd, T(gg), T(z))
    _close(z_out, g["z_p"], 5e-6)
    _close(mel, g["mel"], 2e-5)


@pytest.mark.parametrize("which", [0, 1, 2, "all"])
def test_g5_end2end(sd, golden_dir, which):
    g = np.load(os.path.join(golden_dir, "g5_end2end.npz"))
    b = {k: T(v) for k, v in gc.g5_batch(which).items()}
    r = ref.forward_infer(sd, b["word_tokens"], (b["keys"], b["values"], b["key_map"], b["pinyin"], b["pinyin_map"]),
                          b["pron_modified"], z_p=lambda B, T4: T(gc.g5_noise(B, T4, which)))
    tag = f"b{which}"
    _close(r["dur"], g[tag + ".dur"], 5e-6)
    _close(r["x_mask"], g[tag + ".x_mask"], 0)           # identical integer durations / T_mel
    _close(r["word_encoder_out"], g[tag + ".word_encoder_out"], 2e-5)
    _close(r["pron_attn"], g[tag + ".pron_attn"], 2e-6)
    _close(r["mel_out"], g[tag + ".mel_out"], 1e-4)
    for u in range(b["word_tokens"].shape[0]):
        ids = ref.decode_pinyin(r["pron_attn"][u], b["pinyin"][u])
        want = g[tag + ".pinyin_ids"][u]
        assert ids == want[:len(ids)].tolist()


def test_g6_hifigan(golden_dir):
    g = np.load(os.path.join(golden_dir, "g6_hifigan.npz"))
    raw = {k: T(v) for k, v in synth.hifigan_state_dict(gc.SEED).items()}
    hsd = href.fold_weight_norm(raw)
    # weight-norm folding for Conv1d and ConvTranspose1d (norm over dims != 0 in both cases)
    _close(hsd["conv_pre.weight"][:8], g["folded.conv_pre.weight.head"], 1e-6)
    _close(hsd["ups.0.weight"][:8], g["folded.ups.0.weight.head"], 1e-6)
    mel = gc.g6_mel()
    with torch.no_grad():
        c = torch.as_tensor(mel).unsqueeze(0).transpose(2, 1)
        wav, stages = href.generator_forward(hsd, synth.hifigan_config(), c, return_stages=True)
    for i in range(4):
        _close(stages[f"ups.{i}"][0, :, :64], g[f"ups.{i}.head"], 2e-5)
    _close(wav.view(-1), g["wav"], 2e-5)
    _close(href.spec2wav(hsd, synth.hifigan_config(), mel), g["wav"], 2e-5)
    assert wav.shape[-1] == 32 * 256


@pytest.mark.parametrize("which", ["dec", "enc"])
def test_g8_fft_blocks_oracle_vs_reference_golden(golden_dir, which):
    """oracle/fft_blocks_ref.py vs the reference's FFTBlocks: positional embedding with the
    first-channel-zero quirk, bias-free attention projections, k**-0.5 GELU FFN, LayerNorm bias leaking through the
    SAME-padded conv at the sequence end, a 1-frame utterance"""
    from oracle import fft_blocks_ref as fref
    from dict_tts_amd import synth
    g = np.load(os.path.join(golden_dir, "g8_fft_blocks.npz"))
    cfg = gc.G8_CASES[which]
    sd = {k: torch.from_numpy(v) for k, v in synth.fft_blocks_state_dict(gc.SEED, 192, **cfg).items()}
    x, lens = gc.g8_inputs(which)
    y = fref.fft_blocks(sd, torch.from_numpy(x), num_heads=2, kernel_size=cfg["kernel_size"],
                        use_pos_embed=cfg["use_pos_embed"], use_last_norm=cfg["use_last_norm"])
    assert y.shape == g[which + ".out"].shape
    assert np.abs(y.numpy() - g[which + ".out"]).max() <= 2e-5
    for b, n in enumerate(lens):
        assert not bool((y[b, n:] != 0).any())          # padded frames are exactly zero
