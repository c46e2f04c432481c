"""GPU parity tests (run with `-m gpu` on an MI355X): the HIP path, called through the C ABI via the Python shims,
against (a) the golden fixtures produced by the REFERENCE implementation and (b) the CPU oracle on the same
seeded inputs.  Tolerances are BASELINE.json's: mel max-abs <= 1e-3, integer durations exact, waveform
RMS(gpu - ref) <= 1e-4 AND |RMS(gpu) - RMS(ref)| <= 1e-4 for the default vocoder mode (DTTS_VOC_F16, the one bench.py
measures), decoded pinyin identical.  The all-bf16 mode is held to its own, looser, documented bound."""
import os

import numpy as np
import pytest
import torch

import golden_cases as gc
from dict_tts_amd import abi, synth

pytestmark = pytest.mark.gpu
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
rms = lambda a: float(np.sqrt(np.mean(np.square(np.asarray(a, dtype=np.float64)))))


@pytest.fixture(scope="module")
def acoustic():
    from dict_tts_amd import model
    m = model.PortaSpeech_dict(hparams={})
    m.load_state_dict({k: T(v) for k, v in synth.dict_tts_state_dict(gc.SEED, n_phone=6).items()}, strict=True)
    return m


@pytest.fixture(scope="module")
def oracle_sd():
    from oracle import hifigan_ref as href
    return href.fold_weight_norm({k: T(v) for k, v in synth.dict_tts_state_dict(gc.SEED, n_phone=6).items()})


@pytest.fixture(scope="module")
def voc_sd():
    return {k: T(v) for k, v in synth.hifigan_state_dict(gc.SEED).items()}


@pytest.fixture(scope="module")
def oracle_voc_sd(voc_sd):
    from oracle import hifigan_ref as href
    return href.fold_weight_norm(voc_sd)


def _vocoder(voc_sd, precision):
    from dict_tts_amd import vocoder
    return vocoder.HifiGAN(state_dict=voc_sd, config=synth.hifigan_config(), precision=precision)


@pytest.fixture(scope="module")
def _voc_f16(voc_sd):
    from dict_tts_amd import vocoder
    # PINNED to the benched mode: explicit precision (no automatic fallback) with the range guard on for EVERY call, so a
    # clamped / overflowed fp16 activation raises inside the test instead of silently moving the suite to bf16x3
    return vocoder.HifiGAN(state_dict=voc_sd, config=synth.hifigan_config(), precision="f16", range_guard=True)


@pytest.fixture
def voc(_voc_f16):
    """the default (and benched) mode: fp16 ResBlock operands + split-operand serial convolutions.  The mode is asserted before AND
    after every test that uses it."""
    assert _voc_f16.precision == abi.VOC_F16 and _voc_f16._census and not _voc_f16._auto
    yield _voc_f16
    assert _voc_f16.precision == abi.VOC_F16 and _voc_f16._census, "the vocoder left DTTS_VOC_F16 during the test"
    torch.cuda.synchronize()
    assert not _voc_f16.overflowed(), "the always-on detector saw non-finite pre-tanh samples during the test"


@pytest.fixture(scope="module")
def voc_plain(voc_sd):
    """DTTS_VOC_F16 exactly as bench.py builds it: explicit precision, NO range guard = the release (GUARD = false) kernel instantiations"""
    return _vocoder(voc_sd, "f16")


@pytest.fixture(scope="module")
def voc_bf16(voc_sd):
    return _vocoder(voc_sd, abi.VOC_BF16)


@pytest.fixture(scope="module")
def voc_bf16_unfused(voc_sd):
    from dict_tts_amd import vocoder
    return vocoder.HifiGAN(state_dict=voc_sd, config=synth.hifigan_config(), precision=abi.VOC_BF16, unfused=True)


def wave_gate(w, ref):
    """BASELINE.md §4 / SURVEY §8d: RMS(gpu - ref) and |RMS(gpu) - RMS(ref)| both <= 1e-4"""
    assert w.shape == ref.shape
    assert rms(w - ref) <= 1e-4 and abs(rms(w) - rms(ref)) <= 1e-4, (rms(w - ref), rms(w), rms(ref))


@pytest.fixture(scope="module")
def voc_x3(voc_sd):
    return _vocoder(voc_sd, abi.VOC_BF16X3)


def _run(model, batch, z=None, mel2word=None):
    b = {k: T(v) for k, v in batch.items()}
    return model((b["word_tokens"], None), b["pron_modified"], (None, None, None), None, None,
                 (b["keys"], b["values"], b["key_map"], b["pinyin"], b["pinyin_map"]), infer=True, z_p=z, mel2word=mel2word)


# ------------------------------------------------------------------------------------------------ acoustic model
@pytest.mark.parametrize("which", [0, 1, 2, "all"])
def test_g5_end2end_vs_reference_golden(acoustic, golden_dir, which):
    g = np.load(os.path.join(golden_dir, "g5_end2end.npz"))
    tag = f"b{which}"
    batch = gc.g5_batch(which)
    B = batch["word_tokens"].shape[0]
    T_mel = g[tag + ".mel_out"].shape[1]
    r = _run(acoustic, batch, z=T(gc.g5_noise(B, T_mel // 4, which)))
    assert np.array_equal(r["x_mask"].cpu().numpy(), g[tag + ".x_mask"]), "durations / T_mel differ from the reference"
    assert np.abs(r["dur"].cpu().numpy() - g[tag + ".dur"]).max() <= 1e-4
    assert np.abs(r["word_encoder_out"].cpu().numpy() - g[tag + ".word_encoder_out"]).max() <= 1e-4
    assert np.abs(r["pron_attn"].cpu().numpy() - g[tag + ".pron_attn"]).max() <= 1e-5
    err = np.abs(r["mel_out"].cpu().numpy() - g[tag + ".mel_out"]).max()
    assert err <= 1e-3, err
    from dict_tts_amd.model import decode_pinyin_ids
    for u in range(B):
        ids = decode_pinyin_ids(r["pron_attn"][u], batch["pinyin"][u])
        assert ids == g[tag + ".pinyin_ids"][u][:len(ids)].tolist()


def test_s2pa_edge_cases_vs_oracle(acoustic, oracle_sd):
    """a fully padded word row (uniform attention), forced pronunciations, ragged lengths, an UNK word"""
    from oracle import dict_tts_ref as ref
    st = synth.biaobei_struct()
    sents = [st["sentences"][3], st["sentences"][7][:4], st["sentences"][11]]
    batch = synth.make_batch(sents, gc.SEED, pron_every=1)
    batch["key_map"][1, 2, :] = 0          # word row with every gloss token masked
    batch["pron_modified"][0, 1] = 6       # a sense index above pinyin_map.max(): the rule must not fire
    b = {k: T(v) for k, v in batch.items()}
    want = ref.forward_infer(oracle_sd, b["word_tokens"], (b["keys"], b["values"], b["key_map"], b["pinyin"], b["pinyin_map"]),
                             b["pron_modified"], z_p=lambda B, T4: T(synth.noise(7, B, T4)))
    T_mel = want["mel_out"].shape[1]
    got = _run(acoustic, batch, z=T(synth.noise(7, 3, T_mel // 4)))
    assert torch.equal(got["mel2word"].cpu(), want["mel2word"])
    assert (got["dict_attn"].cpu() - want["dict_attn"]).abs().max() <= 1e-5
    assert (got["pron_attn"].cpu() - want["pron_attn"]).abs().max() <= 1e-5
    L = batch["key_map"].shape[2]
    assert torch.allclose(got["dict_attn"][1, 0, :, 2].cpu(), torch.full((L,), 1.0 / L), atol=1e-6)
    assert (got["word_encoder_out"].cpu() - want["word_encoder_out"]).abs().max() <= 1e-4
    assert (got["mel_out"].cpu() - want["mel_out"]).abs().max() <= 1e-3


def test_teacher_forced_mel2word_and_batch_padding_semantics(acoustic, oracle_sd):
    """teacher-forced durations (model.py:77) and the unmasked-decoder behaviour on padded frames"""
    from oracle import dict_tts_ref as ref
    batch = synth.biaobei_batch(0, 4, gc.SEED)
    m2w = synth.teacher_mel2word(batch["word_tokens"], 7, 3)
    m2w = m2w[:, : m2w.shape[1] - (m2w.shape[1] % 4) + 1] if m2w.shape[1] % 4 == 0 else m2w  # force a ragged T
    b = {k: T(v) for k, v in batch.items()}
    want = ref.forward_infer(oracle_sd, b["word_tokens"], (b["keys"], b["values"], b["key_map"], b["pinyin"], b["pinyin_map"]),
                             b["pron_modified"], mel2word=T(m2w), z_p=lambda B, T4: T(synth.noise(11, B, T4)))
    T_mel = want["mel_out"].shape[1]
    assert T_mel % 4 == 0
    got = _run(acoustic, batch, z=T(synth.noise(11, 4, T_mel // 4)), mel2word=T(m2w))
    assert got["mel_out"].shape == want["mel_out"].shape
    assert torch.equal(got["mel2word"].cpu(), want["mel2word"])
    assert (got["mel_out"].cpu() - want["mel_out"]).abs().max() <= 1e-3   # ALL frames, padded ones included


def test_fused_prior_flow_equals_launch_by_launch(acoustic, monkeypatch):
    """the prior flow as one kernel (flowstack.hip: split-bf16 contractions, hardware exp2 / rcp in the gate) against the same flow launch by
    launch on the generic exact-fp32 kernels (dtts_config.tune_flags bit 8): same noise, same durations, mel within a
    tenth of the reference tolerance on every frame incl. the padded ones and the chunk seams (T_mel/4 > 96 rows: two chunks per utterance)"""
    from dict_tts_amd import model
    plain = model.PortaSpeech_dict(hparams={"dtts_tune_flags": 256})
    plain.load_state_dict({k: T(v) for k, v in synth.dict_tts_state_dict(gc.SEED, n_phone=6).items()}, strict=True)
    batch = synth.biaobei_batch(4, 6, gc.SEED)
    m2w = synth.teacher_mel2word(batch["word_tokens"], 45, 9)     # long utterances: 45 frames per word
    T_mel = m2w.shape[1] + (-m2w.shape[1]) % 4
    z = T(synth.noise(13, 6, T_mel // 4))
    a = _run(acoustic, batch, z=z, mel2word=T(m2w))
    b = _run(plain, batch, z=z, mel2word=T(m2w))
    assert a["mel_out"].shape == b["mel_out"].shape and a["mel_out"].shape[1] // 4 > 96
    assert torch.equal(a["mel2word"].cpu(), b["mel2word"].cpu())
    d = float((a["mel_out"] - b["mel_out"]).abs().max())
    assert 0.0 < d <= 1e-4, d       # different arithmetic (so not identical), far inside the 1e-3 gate


def test_length_regulator_device_vs_reference_golden(golden_dir):
    """G3 integer semantics on the device kernels through dtts_length_regulate: the reference's mel2word for crafted
    integer durations (zeros in the middle, an all-zero utterance -> ones, ilens shorter than T_w), and torch.round's
    half-to-even ties"""
    g = np.load(os.path.join(golden_dir, "g3_duration.npz"))
    d, il = gc.g3_int_durations()
    want = g["mel2word_int"]                                   # produced by the reference LengthRegulator
    ctx = abi.Context()
    dur = torch.log(T(d).double() + 1.0).float().cuda()        # exp(dur) - 1 rounds back to d exactly
    assert torch.equal(torch.clamp(torch.round(dur.cpu().exp() - 1), min=0).long(), T(d))
    ilens = T(il).to(torch.int32).cuda()
    cap = 64
    m2w = torch.full((d.shape[0], cap), -1, dtype=torch.int64, device="cuda")
    t_max = ctx.length_regulate(dur.data_ptr(), ilens.data_ptr(), d.shape[0], d.shape[1], m2w.data_ptr(), cap, None)
    assert t_max == want.shape[1]
    assert np.array_equal(m2w.cpu().numpy()[:, :t_max], want)
    assert int(m2w[:, t_max:].abs().max()) == 0
    # random log-durations, ragged ilens: device exp/round/scan vs the oracle (torch.round = half-to-even; an exact .5
    # cannot be forced through exp(), so ties are covered by the kernel using rintf, the same IEEE rounding)
    rng = np.random.default_rng(3)
    Bn, Tn = 16, 40
    dur = torch.from_numpy(rng.uniform(0.0, 3.2, (Bn, Tn)).astype(np.float32))
    il = torch.from_numpy(rng.integers(1, Tn + 1, Bn).astype(np.int64))
    from oracle import dict_tts_ref as ref_mod
    d_int = torch.clamp(torch.round(dur.exp() - 1), min=0).long()
    want = ref_mod.length_regulator(d_int, il)
    cap2 = int(want.shape[1]) + 7
    m2w = torch.full((Bn, cap2), -1, dtype=torch.int64, device="cuda")
    dur_d, il_d = dur.cuda(), il.to(torch.int32).cuda()      # keep the device tensors alive across the call
    t_max = ctx.length_regulate(dur_d.data_ptr(), il_d.data_ptr(), Bn, Tn, m2w.data_ptr(), cap2, None)
    assert t_max == want.shape[1] and np.array_equal(m2w.cpu().numpy()[:, :t_max], want.numpy())
    with pytest.raises(abi.DttsError, match="exceed the capacity"):
        ctx.length_regulate(dur_d.data_ptr(), il_d.data_ptr(), Bn, Tn, m2w.data_ptr(), 4, None)


# ------------------------------------------------------------------------------------------------ vocoder
def test_g6_hifigan_vs_reference_golden(voc, voc_bf16, voc_x3, golden_dir):
    g = np.load(os.path.join(golden_dir, "g6_hifigan.npz"))
    mel = gc.g6_mel()
    ref_wav = g["wav"]
    w = voc.spec2wav(mel)
    assert w.shape == (32 * 256,)
    wave_gate(w, ref_wav)                                   # the default mode meets the full waveform gate
    assert np.abs(w - ref_wav).max() <= 1e-3
    w3 = voc_x3.spec2wav(mel)
    assert w3.shape == ref_wav.shape == (32 * 256,)
    assert rms(w3 - ref_wav) <= 1e-4 and abs(rms(w3) - rms(ref_wav)) <= 1e-4, (rms(w3 - ref_wav), rms(w3), rms(ref_wav))
    assert np.abs(w3 - ref_wav).max() <= 1e-3
    w1 = voc_bf16.spec2wav(mel)
    assert abs(rms(w1) - rms(ref_wav)) <= 1e-4, (rms(w1), rms(ref_wav))
    assert rms(w1 - ref_wav) <= 0.02 * rms(ref_wav), rms(w1 - ref_wav) / rms(ref_wav)   # bf16 operands: ~1 % noise


def test_hifigan_ragged_batch_equals_per_utterance_oracle(voc, voc_x3, voc_bf16, oracle_voc_sd):
    """spec2wav_batch(list) == the reference's one-utterance-per-call spec2wav for every item (zero padding at the
    utterance end, not the batch end); lengths straddle the time-tile sizes"""
    from oracle import hifigan_ref as href
    lens = [5, 33, 17, 64, 1]
    mels = [synth.random_mel(100 + i, n, f"rag{i}") for i, n in enumerate(lens)]
    want = [href.spec2wav(oracle_voc_sd, synth.hifigan_config(), m).numpy() for m in mels]
    got3 = voc_x3.spec2wav_batch(mels)
    got1 = voc_bf16.spec2wav_batch(mels)
    got = voc.spec2wav_batch(mels)
    for w, a, b, g, n in zip(want, got3, got1, got, lens):
        assert a.shape == w.shape == (n * 256,)
        wave_gate(g, w)
        assert rms(a - w) <= 1e-4, rms(a - w)
        assert abs(rms(b) - rms(w)) <= 1e-4
    # samples past an utterance's end are zero in the batched output
    full = voc.forward_batch(torch.stack([T(np.pad(m, ((0, 64 - m.shape[0]), (0, 0)))) for m in mels]).cuda(),
                                torch.tensor(lens, dtype=torch.int32))
    assert float(full[0, 5 * 256:].abs().max()) == 0.0


def test_hifigan_many_utterances_persistent_tiles(voc, voc_plain, oracle_voc_sd):
    """the fused kernels' persistent workgroups walk the batch's valid tiles through a tile table (utterance -> tile count prefix sums):
    70 utterances, lengths 0..61 incl. empty and one-frame ones, enough tiles that every workgroup takes several — each row of the
    batched output is BIT-identical to the same utterance run alone (same tile origins), zero past its end, and within the waveform
    gate of the oracle"""
    assert voc.precision == abi.VOC_F16 and voc._census   # the benched mode with the census on: a clamp raises
    from oracle import hifigan_ref as href
    rng = np.random.RandomState(7)
    lens = [0, 1, 61, 2, 33] + [int(v) for v in rng.randint(0, 62, size=65)]
    B, Tm = len(lens), 64
    mel = np.zeros((B, Tm, 80), np.float32)
    for i, n in enumerate(lens):
        if n:
            mel[i, :n] = synth.random_mel(300 + i, n, f"many{i}")
    full = voc.forward_batch(T(mel).cuda(), torch.tensor(lens, dtype=torch.int32)).cpu().numpy()
    assert np.isfinite(full).all()
    # the unguarded instantiations (what bench.py times) compute the same bits as the guarded ones
    assert voc_plain.precision == abi.VOC_F16 and not voc_plain._census
    assert np.array_equal(full, voc_plain.forward_batch(T(mel).cuda(), torch.tensor(lens, dtype=torch.int32)).cpu().numpy())
    for i, n in enumerate(lens):
        assert float(np.abs(full[i, n * 256:]).max(initial=0.0)) == 0.0
        if n == 0:
            continue
        if i < 12 or n >= 60:
            alone = voc.spec2wav(mel[i, :n])
            assert np.array_equal(alone, full[i, :n * 256]), i
        # EVERY non-empty row against the oracle (the CPU oracle does ~3 k frames / s: 2 k frames here)
        wave_gate(full[i, :n * 256], href.spec2wav(oracle_voc_sd, synth.hifigan_config(), mel[i, :n]).numpy())


def test_hifigan_linearity_free_properties_long(voc):
    """size-independent properties at a realistic length (400 frames): determinism, finite output in (-1, 1),
    and shift-consistency — the middle of the utterance does not depend on what is 200 frames away"""
    mel = synth.random_mel(5, 400, "long")
    a = voc.spec2wav(mel)
    b = voc.spec2wav(mel)
    assert np.array_equal(a, b)
    assert np.isfinite(a).all() and np.abs(a).max() < 1.0
    mel2 = mel.copy()
    mel2[:100] = synth.random_mel(6, 100, "other")
    c = voc.spec2wav(mel2)
    mid = slice(300 * 256, 350 * 256)   # > receptive field away from the edited frames
    assert np.array_equal(a[mid], c[mid])


def test_missing_weight_fails_loudly(voc_sd):
    from dict_tts_amd import vocoder
    sd = {k: v for k, v in voc_sd.items() if not k.startswith("resblocks.7.convs2.1")}
    with pytest.raises(abi.DttsError, match="resblocks.7.convs2.1"):
        vocoder.HifiGAN(state_dict=sd, config=synth.hifigan_config())
    ctx = abi.Context()
    with pytest.raises(abi.DttsError, match="before a successful"):
        ctx.text2mel_decode(1, 1, None)


def test_fused_resblock_equals_unfused(voc_bf16, voc_bf16_unfused, oracle_voc_sd):
    """the fused kernels (rblock.hip: whole ResBlocks at C <= 64; vpair.hip: one ResBlock iteration at C = 128) and the
    per-convolution path have the same bf16
    rounding POINTS (conv inputs), fp32 accumulation and fp32 residual; their fp32 summation order differs (the fused
    kernel starts the accumulator at the bias and accumulates conv2 into the residual registers), which flips an
    occasional bf16 rounding.  So: they agree to well below the bf16 noise floor, and both sit at the same distance
    from the fp32 oracle.  Lengths straddle the fused kernel's time tiles."""
    from oracle import hifigan_ref as href
    lens = [3, 50, 97]
    mels = [synth.random_mel(300 + i, n, f"fuse{i}") for i, n in enumerate(lens)]
    a = voc_bf16.spec2wav_batch(mels)
    b = voc_bf16_unfused.spec2wav_batch(mels)   # dtts_config.vocoder_unfused = 1: one kernel per convolution
    for x, y, n, m in zip(a, b, lens, mels):
        w = href.spec2wav(oracle_voc_sd, synth.hifigan_config(), m).numpy()
        assert x.shape == y.shape == w.shape == (n * 256,)
        ex, ey = rms(x - w), rms(y - w)
        # two independent bf16-noise realisations would sit sqrt(2) * e apart; the paths share every rounding point
        assert rms(x - y) <= 0.75 * max(ex, ey), (rms(x - y), ex, ey)
        assert ex <= 1.15 * ey + 1e-5 and ey <= 1.15 * ex + 1e-5, (ex, ey)
        assert abs(rms(x) - rms(w)) <= 1e-4 and abs(rms(y) - rms(w)) <= 1e-4


# ------------------------------------------------------------------------------------------------ BASELINE configs 4, 5
def test_config4_long_form_1000_chars(acoustic, oracle_sd, voc, oracle_voc_sd):
    """BASELINE.json configs[3]: 1000-char input (T_w = 1002), teacher-forced 5 frames/char -> ~5k mel frames, B=1:
    attention over 1002 words, every conv tiled over 5k..1.28M time steps, mel and waveform vs the oracle"""
    assert voc.precision == abi.VOC_F16 and voc._census   # the benched mode with the census on: a clamp raises
    from oracle import dict_tts_ref as ref
    from oracle import hifigan_ref as href
    st = synth.biaobei_struct()
    ids = [w for s in st["sentences"] for w in s][:1000]
    batch = synth.make_batch([ids], gc.SEED)
    m2w = synth.teacher_mel2word(batch["word_tokens"], 5, 5)
    assert batch["word_tokens"].shape[1] == 1002 and m2w.shape[1] == 5010
    b = {k: T(v) for k, v in batch.items()}
    want = ref.forward_infer(oracle_sd, b["word_tokens"], (b["keys"], b["values"], b["key_map"], b["pinyin"], b["pinyin_map"]),
                             b["pron_modified"], mel2word=T(m2w), z_p=lambda B, T4: T(synth.noise(21, B, T4)))
    T_mel = want["mel_out"].shape[1]
    assert T_mel == 5012
    got = _run(acoustic, batch, z=T(synth.noise(21, 1, T_mel // 4)), mel2word=T(m2w))
    assert (got["word_encoder_out"].cpu() - want["word_encoder_out"]).abs().max() <= 2e-4
    assert (got["mel_out"].cpu() - want["mel_out"]).abs().max() <= 1e-3
    mel = want["mel_out"][0].numpy()
    wav = voc.spec2wav(mel)
    wref = href.spec2wav(oracle_voc_sd, synth.hifigan_config(), mel).numpy()
    assert wav.shape == wref.shape == (5012 * 256,)
    wave_gate(wav, wref)


def test_config5_dictionary_stress_mixed_lengths(acoustic, oracle_sd):
    """BASELINE.json configs[4] (one GPU's share): B=32 mixed-length utterances (6..60 chars) drawn from the whole
    dictionary structure with heteronyms over-sampled x5, word ids up to word_size=8000, every char forced to a sense"""
    from oracle import dict_tts_ref as ref
    st = synth.biaobei_struct()
    rng = np.random.default_rng(5)
    ids = np.array(sorted(st["entries"].keys()))
    wts = np.array([5.0 if len(st["entries"][i]) > 1 else 1.0 for i in ids])
    wts /= wts.sum()
    sents = [rng.choice(ids, size=int(rng.integers(6, 61)), p=wts).tolist() for _ in range(32)]
    batch = synth.make_batch(sents, gc.SEED, pron_every=3)
    batch["word_tokens"][batch["word_tokens"] == synth.BOS_ID] = 7999          # highest row of the 8000-word table
    b = {k: T(v) for k, v in batch.items()}
    want = ref.forward_infer(oracle_sd, b["word_tokens"], (b["keys"], b["values"], b["key_map"], b["pinyin"], b["pinyin_map"]),
                             b["pron_modified"], z_p=lambda B, T4: T(synth.noise(31, B, T4)))
    T_mel = want["mel_out"].shape[1]
    got = _run(acoustic, batch, z=T(synth.noise(31, 32, T_mel // 4)))
    assert torch.equal(got["mel2word"].cpu(), want["mel2word"])
    assert (got["pron_attn"].cpu() - want["pron_attn"]).abs().max() <= 1e-5
    assert (got["dict_attn"].cpu() - want["dict_attn"]).abs().max() <= 1e-5
    assert (got["mel_out"].cpu() - want["mel_out"]).abs().max() <= 1e-3
    from dict_tts_amd.model import decode_pinyin_ids
    for u in range(32):
        assert decode_pinyin_ids(got["pron_attn"][u], batch["pinyin"][u]) == ref.decode_pinyin(want["pron_attn"][u], b["pinyin"][u])


# ------------------------------------------------------------------------------------------------ resident dictionary
def test_resident_dictionary_ids_equal_collated_tensors(acoustic):
    """dtts_dict_table_upload + dtts_text2mel_encode_ids (batches carry only ids) must reproduce
    dtts_text2mel_encode on the tensors DictTTSDataset.collater would build from the same table: ragged lengths, heteronyms,
    forced senses, an UNK (zero) entry."""
    st = synth.biaobei_struct()
    entries = dict(st["entries"])
    unk = 7000
    entries[unk] = [[3, 0, -1]]                       # a char absent from zh-dict.json: the 3-token zero entry
    table = synth.dict_table(gc.SEED, entries)
    acoustic.upload_dict_table(table)
    sents = [st["sentences"][5], st["sentences"][9][:5] + [unk], st["sentences"][40], st["sentences"][41][:3]]
    tb = synth.make_batch(sents, gc.SEED, entries, pron_every=2)
    ib = synth.make_id_batch(sents, table, pron_every=2)
    assert np.array_equal(tb["word_tokens"], ib["word_tokens"]) and np.array_equal(tb["pron_modified"], ib["pron_modified"])
    assert tb["keys"].shape[2] == ib["L_k"] and tb["pinyin"].shape[2] == ib["P"]
    ra = acoustic((T(tb["word_tokens"]), None), T(tb["pron_modified"]), (None,) * 3, None, None,
                  (T(tb["keys"]), T(tb["values"]), T(tb["key_map"]), T(tb["pinyin"]), T(tb["pinyin_map"])), infer=True)
    zp = ra["z_p_in"]
    rb = acoustic.forward_ids(T(ib["word_tokens"]), T(ib["entry_ids"]), T(ib["pron_modified"]), ib["L_k"], ib["P"], z_p=zp)
    # round 3: the resident table holds the PROJECTED rows (K = key Wk^T, V = value Wv^T), so the id path computes k . q in the
    # reference's association order while the tensor API keeps the re-associated key . (Wk^T q): the same numbers up to fp32
    # rounding, no longer bit for bit — both within the oracle tolerances of each other, integer durations identical
    assert torch.equal(ra["mel2word"], rb["mel2word"])
    for k, tol in (("dur", 1e-5), ("pron_attn", 1e-5), ("dict_attn", 1e-5), ("word_encoder_out", 1e-4), ("mel_out", 1e-3)):
        assert (ra[k] - rb[k]).abs().max() <= tol, (k, float((ra[k] - rb[k]).abs().max()))
    ctx = abi.Context()
    with pytest.raises(abi.DttsError, match="not finalized|before dtts_dict_table_upload"):
        ctx.text2mel_encode_ids(1, 1, None, None, 1, 4, 8, 2, None)


@pytest.mark.gpu
def test_run_inference_two_stream_pipeline_equals_serial(acoustic, voc, tmp_path):
    """run_inference(pipeline=True) overlaps the vocoder of batch i with text->mel of batch i+1 on two streams; the
    files it writes must be byte-identical to the serial loop's (same kernels, same inputs, buffers never shared)"""
    from scipy.io import wavfile
    from dict_tts_amd import infer
    st = synth.biaobei_struct()
    batches = []
    for k, n in enumerate((3, 1, 4, 2)):               # different batch sizes and lengths: every arena is re-used/re-sized
        sent = st["sentences"][10 * k: 10 * k + n]
        b = {key: T(v) for key, v in synth.make_batch(sent, 100 + k).items()}
        b["item_name"] = [f"{k:02d}_{i:02d}" for i in range(n)]
        b["text"] = [f"utt {k} {i}" for i in range(n)]
        batches.append(b)
    enc = [f"p{i}" for i in range(185)]
    out = {}
    for mode in (False, True):
        torch.manual_seed(7)                            # z_p is drawn from the CPU generator, in batch order
        d = tmp_path / ("pipe" if mode else "serial")
        rows = infer.run_inference(acoustic, voc, batches, str(d), enc, pipeline=mode)
        assert len(rows) == 10
        out[mode] = (rows, [wavfile.read(os.path.join(d, "wavs", r["wav_fn_pred"] + ".wav"))[1] for r in rows])
    assert out[False][0] == out[True][0]
    for a, b in zip(out[False][1], out[True][1]):
        assert a.shape == b.shape and a.shape[0] > 0 and np.array_equal(a, b)


@pytest.mark.gpu
def test_two_stream_harness_redoes_an_overflowed_batch_and_the_one_in_flight(voc_sd):
    """the always-on fp16 overflow detector inside the PIPELINED harness (infer._iter_results, pipeline=True): batch 1 of four carries a mel
    far outside the stated range (x 3e6) and overflows the fp16 ResBlocks.  Its waveform comes back finite and equal to the bf16x3 vocoder's
    (AUTO redid it), batch 2 — already in flight in fp16 when the overflow was noticed — is redone too, batches 0 / 2 / 3 equal what the
    serial loop produces, and with an explicit 'f16' vocoder the same run raises instead of handing a poisoned waveform on."""
    from dict_tts_amd import infer, vocoder
    cfg = synth.hifigan_config()
    mels = [T(np.stack([synth.random_mel(40 + k, 48, f"pipe{k}")])).cuda() for k in range(4)]
    mels[1] = mels[1] * 3e6

    class FakeModel:   # the harness only needs mel_out / mel_lens / pron_attn from the acoustic model
        def __call__(self, *a, **kw):
            k = FakeModel.k
            FakeModel.k += 1
            return {"mel_out": mels[k], "mel_lens": torch.tensor([48], dtype=torch.int32, device="cuda"), "pron_attn": torch.zeros(1, 3, 2)}
    batches = [{"word_tokens": torch.ones(1, 3, dtype=torch.int64), "keys": None, "values": None, "key_map": None, "pinyin": torch.zeros(1, 3, 2, dtype=torch.int64),
                "pinyin_map": None} for _ in range(4)]
    x3 = vocoder.HifiGAN(state_dict=voc_sd, config=cfg, precision="bf16x3")
    f16 = vocoder.HifiGAN(state_dict=voc_sd, config=cfg, precision="f16")
    want = []
    for k in range(4):
        v = x3 if k == 1 else f16
        want.append(v.to_int16(v.forward_batch(mels[k], torch.tensor([48], dtype=torch.int32)), torch.tensor([48], dtype=torch.int32)).cpu().numpy()[0])
    torch.cuda.synchronize()
    assert not f16.overflowed()
    import warnings
    FakeModel.k = 0
    auto = vocoder.HifiGAN(state_dict=voc_sd, config=cfg)
    with warnings.catch_warnings(record=True) as ws:
        warnings.simplefilter("always")
        got = [w[0] for _, _, w in infer._iter_results(FakeModel(), auto, batches, pipeline=True)]
    assert auto.precision == abi.VOC_BF16X3 and any("overflowed" in str(w.message) for w in ws)
    assert len(got) == 4 and all(np.isfinite(g.astype(np.float64)).all() for g in got)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])        # fp16 before the overflow; the overflowed batch redone in bf16x3
    for k in (2, 3):                                                                  # after the switch: bf16x3, within the int16 step of the fp16 result
        assert np.abs(got[k].astype(np.int32) - want[k].astype(np.int32)).max() <= 16, k
    FakeModel.k = 0
    with pytest.raises(abi.DttsError, match="overflowed"):
        list(infer._iter_results(FakeModel(), f16, batches, pipeline=True))


@pytest.mark.gpu
@pytest.mark.parametrize("norm", [False, True])
def test_device_int16_conversion_equals_reference_rule(voc, norm):
    """dtts_wav_to_int16 == utils/audio.py:11-16 per utterance over its own valid samples (wav / max|wav| if norm;
    * 32767 in fp32; truncating astype(int16)), bit for bit; samples past an utterance's end are 0"""
    from dict_tts_amd import infer
    hop = voc.hop
    rng = np.random.default_rng(5)
    lens = np.array([7, 3, 0, 5], np.int32)
    wav = rng.uniform(-1, 1, (4, 7 * hop)).astype(np.float32)
    wav[0, :6] = [0.0, 0.5, -0.5, 0.99997, -1.0, 1.0]
    wav[1] *= 0.01                                        # quiet utterance: norm matters
    wav[1, 3 * hop:] = 0.9                                # beyond its end: must not enter its max
    got = voc.to_int16(T(wav).cuda(), T(lens).cuda(), norm=norm).cpu().numpy()
    assert got.dtype == np.int16 and got.shape == wav.shape
    for b in range(4):
        n = int(lens[b]) * hop
        if n:
            assert np.array_equal(got[b, :n], infer.wav_to_int16(wav[b, :n], norm)), b
        assert not got[b, n:].any()
    if not norm:
        assert got[0, :6].tolist() == [0, 16383, -16383, 32766, -32767, 32767]


def test_device_int16_conversion_vs_reference_save_wav_golden(voc, golden_dir):
    """G9: dtts_wav_to_int16 on a ragged batch == the samples the REFERENCE's save_wav wrote (utils/audio.py:11-16,
    oracle/make_golden_io.py), norm off and on, and == the oracle restatement; samples past an utterance's end are 0"""
    from oracle import audio_ref
    g = np.load(os.path.join(golden_dir, "g9_save_wav.npz"))
    lens, wavs = gc.g9_wavs()
    hop = voc.hop
    assert hop == 256
    batch = np.zeros((len(wavs), max(lens) * hop), np.float32)
    for i, w in enumerate(wavs):
        batch[i, :w.shape[0]] = w
    batch[0, lens[0] * hop:] = 0.97                      # junk past the end of utterance 0: must not enter its peak
    for norm in (False, True):
        got = voc.to_int16(T(batch).cuda(), T(np.array(lens, np.int32)).cuda(), norm=norm).cpu().numpy()
        for i, n in enumerate(lens):
            want = g[f"u{i}.norm{int(norm)}"]
            assert np.array_equal(got[i, :n * hop], want), (i, norm)
            assert np.array_equal(audio_ref.save_wav_pcm(wavs[i], norm), want)
            assert not got[i, n * hop:].any()


# ------------------------------------------------------------------------------------------------ BASELINE configs[1] at full size
def test_config2_batch60_full_size_vs_oracle(acoustic, oracle_sd, voc, voc_plain, oracle_voc_sd):
    """BASELINE.json configs[1] at its real size: the first 60 Biaobei sentences as ONE batch (T_w = 27, L_k = 148).
    (1) predicted durations: integer mel2word exact, pinyin strings identical, mel <= 1e-3 on every frame (the decoder
        runs unmasked over the padded batch, SURVEY 0.3, so padding leakage at B = 60 is part of the comparison);
    (2) teacher-forced 22 frames / char (T_mel = 740, the bench shape): mel <= 1e-3 on all 60 x 740 frames, and the
        waveform gates RMS(gpu - ref), |RMS(gpu) - RMS(ref)| <= 1e-4 end to end (GPU mel -> GPU vocoder against oracle mel ->
        oracle vocoder) on the shortest, a middle and the longest utterance."""
    assert voc.precision == abi.VOC_F16 and voc._census   # the benched mode with the census on: a clamp raises
    from dict_tts_amd.model import decode_pinyin_ids
    from oracle import dict_tts_ref as ref
    from oracle import hifigan_ref as href
    st = synth.biaobei_struct()
    batch = synth.make_batch(st["sentences"][:60], gc.SEED)
    B, T_w = batch["word_tokens"].shape
    assert (B, T_w, batch["keys"].shape[2]) == (60, 27, 148)
    b = {k: T(v) for k, v in batch.items()}
    dm = (b["keys"], b["values"], b["key_map"], b["pinyin"], b["pinyin_map"])
    # (1) predicted durations
    want = ref.forward_infer(oracle_sd, b["word_tokens"], dm, b["pron_modified"], z_p=lambda B_, T4: T(synth.noise(61, B_, T4)))
    T_mel = want["mel_out"].shape[1]
    got = _run(acoustic, batch, z=T(synth.noise(61, B, T_mel // 4)))
    assert (got["dur"].cpu() - want["dur"]).abs().max() <= 1e-4
    # integer durations: exact wherever the reference's own round() is numerically decided.  Among the 1,157 words of this
    # batch one sits ON a rounding boundary (utterance 49, word 14: exp(dur) - 1 = 7.499998 on the CPU, 7.500009 here, log
    # durations 8e-6 apart): fp32 implementations cannot agree there (neither do two CPU thread counts), so such numerical
    # ties (|frac - .5| < 5e-5 in the oracle) are listed and excluded, and everything else must be identical.
    v = want["dur"].exp() - 1
    tie = ((v - v.floor() - 0.5).abs() < 5e-5) & (b["word_tokens"] > 0)
    gi = (got["dur"].cpu().exp() - 1).round().clamp(min=0)
    wi = v.round().clamp(min=0)
    n_ties, n_flips = int(tie.sum()), int((gi != wi).sum())
    print(f"\n[config2 B=60] words {int((b['word_tokens'] > 0).sum())}  n_ties (|frac - .5| < 5e-5 in the oracle) {n_ties}  "
          f"n_flips (integer durations that differ from the oracle's) {n_flips}  flips outside ties {int(((gi != wi) & ~tie).sum())}")
    assert n_ties <= 2 and torch.equal(gi[~tie], wi[~tie]), (n_ties, n_flips)
    if not torch.equal(got["mel2word"].cpu(), want["mel2word"]):   # a tie went the other way: compare the rest on the SAME durations
        assert bool(((gi != wi) & ~tie).sum() == 0)
        want = ref.forward_infer(oracle_sd, b["word_tokens"], dm, b["pron_modified"], mel2word=got["mel2word"].cpu(),
                                 z_p=lambda B_, T4: T(synth.noise(61, B_, T4)))
    assert torch.equal(got["mel2word"].cpu(), want["mel2word"]) and got["mel_out"].shape == want["mel_out"].shape
    assert (got["mel_out"].cpu() - want["mel_out"]).abs().max() <= 1e-3
    for u in range(B):
        assert decode_pinyin_ids(got["pron_attn"][u], batch["pinyin"][u]) == ref.decode_pinyin(want["pron_attn"][u], b["pinyin"][u])
    # (2) the bench shape
    m2w = synth.teacher_mel2word(batch["word_tokens"])
    want = ref.forward_infer(oracle_sd, b["word_tokens"], dm, b["pron_modified"], mel2word=T(m2w),
                             z_p=lambda B_, T4: T(synth.noise(62, B_, T4)))
    T_mel = want["mel_out"].shape[1]
    assert T_mel % 4 == 0 and T_mel >= 500
    got = _run(acoustic, batch, z=T(synth.noise(62, B, T_mel // 4)), mel2word=T(m2w))
    assert got["mel_out"].shape == want["mel_out"].shape
    assert (got["mel_out"].cpu() - want["mel_out"]).abs().max() <= 1e-3
    lens = got["mel_lens"].cpu().numpy()
    assert np.array_equal(lens, (want["mel2word"] > 0).sum(1).numpy())
    wav = voc.forward_batch(got["mel_out"], got["mel_lens"]).cpu().numpy()
    assert np.array_equal(wav, voc_plain.forward_batch(got["mel_out"], got["mel_lens"]).cpu().numpy())   # guarded == release kernels
    for u in range(B):   # ALL 60 utterances through the waveform gate (~24 k frames of CPU oracle, a few seconds)
        n = int(lens[u])
        wref = href.spec2wav(oracle_voc_sd, synth.hifigan_config(), want["mel_out"][u, :n].numpy()).numpy()
        wave_gate(wav[u, :n * voc.hop], wref)
        assert not wav[u, n * voc.hop:].any()


def test_b1_waveform_covers_the_padded_frames(acoustic, oracle_sd, voc, oracle_voc_sd, tmp_path):
    """the reference's inference is B = 1 and vocodes ALL T_mel frames of mel_out, including the <= 3 frames added by the
    padding to frames_multiple (they repeat the last word and are valid frames; tasks/tts/dict_tts.py:255): mel_lens
    counts them, run_inference writes T_mel * hop samples and the tail equals spec2wav(mel_out) of the oracle"""
    assert voc.precision == abi.VOC_F16 and voc._census   # the benched mode with the census on: a clamp raises
    from scipy.io import wavfile
    from dict_tts_amd import infer
    from oracle import audio_ref
    from oracle import dict_tts_ref as ref
    from oracle import hifigan_ref as href
    st = synth.biaobei_struct()
    hit = False
    for k in range(6):
        batch = synth.make_batch([st["sentences"][k]], gc.SEED)
        b = {key: T(v) for key, v in batch.items()}
        want = ref.forward_infer(oracle_sd, b["word_tokens"], (b["keys"], b["values"], b["key_map"], b["pinyin"], b["pinyin_map"]),
                                 b["pron_modified"], z_p=lambda B_, T4: T(synth.noise(70 + k, B_, T4)))
        T_mel = want["mel_out"].shape[1]
        raw = int((want["dur"].exp() - 1).round().clamp(min=0).sum())
        got = _run(acoustic, batch, z=T(synth.noise(70 + k, 1, T_mel // 4)))
        assert int(got["mel_lens"][0]) == T_mel == int((want["mel2word"] > 0).sum())
        if raw % 4 == 0:
            continue
        hit = True                                        # this utterance really has pad frames
        wref = href.spec2wav(oracle_voc_sd, synth.hifigan_config(), want["mel_out"][0].numpy()).numpy()
        bt = dict(b)
        bt["item_name"], bt["text"] = ["u"], ["t"]
        bt["z_p"] = T(synth.noise(70 + k, 1, T_mel // 4))
        rows = infer.run_inference(acoustic, voc, [bt], str(tmp_path / f"b1_{k}"), [f"p{i}" for i in range(185)], pipeline=False)
        pcm = wavfile.read(os.path.join(tmp_path / f"b1_{k}", "wavs", rows[0]["wav_fn_pred"] + ".wav"))[1]
        assert pcm.shape == (T_mel * voc.hop,)
        wpcm = audio_ref.save_wav_pcm(wref)
        tail = slice((raw - 2) * voc.hop, T_mel * voc.hop)
        d = pcm[tail].astype(np.float64) - wpcm[tail].astype(np.float64)
        assert rms(d) <= 1e-4 * 32767 + 0.5 and np.abs(d).max() <= 40, (rms(d), np.abs(d).max())   # the waveform gate in int16 LSBs (+ truncation)
        break
    assert hit, "none of the six sentences needed padding to frames_multiple"


def test_g10_reference_written_dict_embed_through_the_id_path_vs_oracle(acoustic, oracle_sd, golden_dir):
    """VERDICT r2 #7a: the ``dict_embed.{idx,data}`` dataset the REFERENCE's IndexedDatasetBuilder wrote (fixture G10) is read
    (dict_tts_amd/dict_embed.py), uploaded as the resident table, and dtts_text2mel_encode_ids on word ids is compared with the
    ORACLE on the tensors the reference's collater builds from the same four items: per sentence get_dict_embeddings
    (tasks/tts/dataset_utils.py:305-330: collate_2d / collate_1d over the words, pad 0), per batch collater (:285-297: collate_3d,
    then one row in front / behind along T_w filled with 0 for keys / values / pinyin and 1 for key_map / pinyin_map)."""
    from dict_tts_amd import dict_embed
    from dict_tts_amd.model import decode_pinyin_ids
    from oracle import dict_tts_ref as ref
    base = os.path.join(golden_dir, "g10_dict_embed")
    table = dict_embed.table_from_dict_embed(base, os.path.join(golden_dir, "g10_pinyin_encoder.pkl"))
    items = dict_embed.read_indexed_dataset(base)
    import pickle
    with open(os.path.join(golden_dir, "g10_pinyin_encoder.pkl"), "rb") as f:
        penc = pickle.load(f)
    acoustic.upload_dict_table(table)
    sents = [[1, 2, 3, 0, 2], [3, 1], [2, 2, 2, 1, 0, 3, 1]]           # word ids = dataset item indices (item 0: the zero entry)
    B, Tw = len(sents), max(len(s) for s in sents) + 2
    L_k = max(int(np.asarray(items[w]["key"]).shape[0]) for s in sents for w in s)
    P = max(len(items[w]["pinyin"]) for s in sents for w in s)
    keys = np.zeros((B, Tw, L_k, 768), np.float32)
    key_map = np.zeros((B, Tw, L_k), np.float32)
    pinyin = np.zeros((B, Tw, P), np.int64)
    pinyin_map = np.zeros((B, Tw, P), np.int64)
    word_tokens = np.zeros((B, Tw), np.int64)
    entry = np.full((B, Tw), -2, np.int32)
    pron_modified = np.zeros((B, Tw), np.int64)
    for b, sen in enumerate(sents):
        word_tokens[b, :len(sen) + 2] = [synth.BOS_ID] + [10 + w for w in sen] + [synth.EOS_ID]
        for t, w in enumerate(sen):
            it = items[w]
            k = np.asarray(it["key"], np.float32)
            keys[b, t + 1, :k.shape[0]] = k
            key_map[b, t + 1, :k.shape[0]] = it["key_map"]
            pinyin[b, t + 1, :len(it["pinyin"])] = [penc.index(x) for x in it["pinyin"]]     # dataset_utils.py:322
            pinyin_map[b, t + 1, :len(it["pinyin"])] = it["pinyin_map"]
            entry[b, t + 1] = w
    key_map[:, 0] = key_map[:, -1] = 1                                   # collater :288-289 (value=1), keys / pinyin rows stay 0
    pinyin_map[:, 0] = pinyin_map[:, -1] = 1
    entry[:, 0] = entry[:, -1] = -1
    pron_modified[0, 2] = 2                                              # the two-sense heteronym forced to its second sense
    pron_modified[2, 6] = 3                                              # the three-sense one to its third
    z = lambda B_, T4: T(synth.noise(71, B_, T4))
    want = ref.forward_infer(oracle_sd, T(word_tokens), (T(keys), T(keys.copy()), T(key_map), T(pinyin), T(pinyin_map)), T(pron_modified), z_p=z)
    T_mel = want["mel_out"].shape[1]
    got = acoustic.forward_ids(T(word_tokens), T(entry), T(pron_modified), L_k, P, z_p=z(B, T_mel // 4))
    assert torch.equal(got["mel2word"].cpu(), want["mel2word"])
    assert (got["pron_attn"].cpu() - want["pron_attn"]).abs().max() <= 1e-5
    assert (got["dict_attn"].cpu() - want["dict_attn"]).abs().max() <= 1e-5
    assert (got["word_encoder_out"].cpu() - want["word_encoder_out"]).abs().max() <= 1e-4
    assert (got["mel_out"].cpu() - want["mel_out"]).abs().max() <= 1e-3
    for u in range(B):
        assert decode_pinyin_ids(got["pron_attn"][u], pinyin[u]) == ref.decode_pinyin(want["pron_attn"][u], T(pinyin[u]))
    assert abs(float(got["pron_attn"][0, 2].max()) - 1.0) <= 1e-6        # the forced rows are one-hot over that sense's tokens
    acoustic.upload_dict_table(synth.dict_table(gc.SEED))               # leave the Biaobei table resident for the other tests


def test_config5_full_dictionary_resident_table_vs_oracle(acoustic, oracle_sd):
    """BASELINE.json configs[4] with the FULL dictionary: all 7,030 zh-dict.json entries (211,072 gloss rows, 618 MiB)
    resident in HBM, uploaded once; one GPU's share of the batch (B = 32 mixed-length utterances of 6..60 characters
    drawn from the whole dictionary, heteronyms over-sampled x5, word ids up to 7,999) runs from ids only and is
    compared with the ORACLE on the tensors the reference's collater would build from the same entries (SURVEY 8f-1:
    the id path against the reference arithmetic, not against the HIP tensor path)."""
    from dict_tts_amd.model import decode_pinyin_ids
    from oracle import dict_tts_ref as ref
    full = synth.zh_dict_struct()
    ent = full["entries"]
    assert full["n_entries"] == len(ent) == 7030 and max(ent) == 7032
    table = synth.dict_table(gc.SEED, ent)
    assert table["keys"].shape == (211072, 768)
    acoustic.upload_dict_table(table)
    rng = np.random.default_rng(55)
    ids = np.array(sorted(ent))
    wts = np.array([5.0 if len(ent[i]) > 1 else 1.0 for i in ids])
    wts /= wts.sum()
    sents = [rng.choice(ids, size=int(rng.integers(6, 61)), p=wts).tolist() for _ in range(32)]
    sents[0][0], sents[1][-1] = int(ids[0]), int(ids[-1])            # first and last table rows
    ib = synth.make_id_batch(sents, table, pron_every=3)
    tb = synth.make_batch(sents, gc.SEED, ent, pron_every=3)
    for k in ("word_tokens",):
        ib[k][ib[k] == synth.BOS_ID] = 7999
        tb[k][tb[k] == synth.BOS_ID] = 7999
    assert np.array_equal(ib["word_tokens"], tb["word_tokens"]) and np.array_equal(ib["pron_modified"], tb["pron_modified"])
    assert (ib["L_k"], ib["P"]) == (tb["keys"].shape[2], tb["pinyin"].shape[2])
    b = {k: T(v) for k, v in tb.items()}
    want = ref.forward_infer(oracle_sd, b["word_tokens"], (b["keys"], b["values"], b["key_map"], b["pinyin"], b["pinyin_map"]),
                             b["pron_modified"], z_p=lambda B_, T4: T(synth.noise(33, B_, T4)))
    T_mel = want["mel_out"].shape[1]
    got = acoustic.forward_ids(T(ib["word_tokens"]), T(ib["entry_ids"]), T(ib["pron_modified"]), ib["L_k"], ib["P"],
                               z_p=T(synth.noise(33, 32, T_mel // 4)))
    assert torch.equal(got["mel2word"].cpu(), want["mel2word"])
    assert (got["pron_attn"].cpu() - want["pron_attn"]).abs().max() <= 1e-5
    assert (got["dict_attn"].cpu() - want["dict_attn"]).abs().max() <= 1e-5
    assert (got["mel_out"].cpu() - want["mel_out"]).abs().max() <= 1e-3
    for u in range(32):
        assert decode_pinyin_ids(got["pron_attn"][u], tb["pinyin"][u]) == ref.decode_pinyin(want["pron_attn"][u], b["pinyin"][u])
    # a second upload replaces the table (the previous one is released) and the Biaobei table still works afterwards
    acoustic.upload_dict_table(synth.dict_table(gc.SEED))
    # sense indices beyond DTTS_MAX_SENSES are rejected, not silently zero-weighted
    bad = synth.dict_table(gc.SEED)
    bad["pinyin_map"] = bad["pinyin_map"].copy()
    bad["pinyin_map"][0] = 16
    with pytest.raises(abi.DttsError, match="at most 15 senses"):
        acoustic.upload_dict_table(bad)


@pytest.mark.gpu
def test_s2pa_long_words_vs_oracle_and_repeatable(acoustic, oracle_sd):
    """words with many live gloss rows (64 / 65 / 128 / 129 / 158 of 160, NON-zero glosses so that a missed row would
    show), a long fully-masked word (uniform weights over all rows, every value row read), a long word past its
    utterance's end (nothing read) -- against the oracle, and bit-identical over repeated launches"""
    from oracle import dict_tts_ref as ref
    st = synth.biaobei_struct()
    sents = [st["sentences"][i] for i in (2, 5, 9, 14)]
    batch = synth.make_batch(sents, gc.SEED + 3, pron_every=2)
    B, T_w, L0 = batch["key_map"].shape
    L = 160                                               # widen the gloss axis (extra rows masked, as batch padding is)
    for k in ("keys", "values", "key_map"):
        pad = [(0, 0)] * batch[k].ndim
        pad[2] = (0, L - L0)
        batch[k] = np.ascontiguousarray(np.pad(batch[k], pad))
    batch["key_map"][:, 0, :] = 1                         # the collater's BOS row: key_map all ones over the padded width
    rng = np.random.default_rng(11)
    D = batch["keys"].shape[-1]

    def fill(b, t, n_live, senses=2):
        km = np.zeros(L, np.float32)
        km[1:1 + n_live] = rng.integers(1, senses + 1, n_live)
        batch["key_map"][b, t] = km
        batch["keys"][b, t] = rng.normal(0, 0.5, (L, D)).astype(np.float32)
        batch["values"][b, t] = rng.normal(0, 0.5, (L, D)).astype(np.float32)

    n_words = (batch["word_tokens"] > 0).sum(1)
    assert n_words.min() >= 7
    for b, (t, n_live) in enumerate([(1, 64), (2, 65), (3, 128), (4, 129)]):
        fill(b, t, n_live)
    fill(0, 5, L - 2, senses=3)
    batch["key_map"][1, 5, :] = 0                        # long fully-masked word inside the utterance, non-zero values
    batch["values"][1, 5] = rng.normal(0, 0.5, (L, D)).astype(np.float32)
    short = int(np.argmin(n_words))
    if n_words[short] < T_w:                              # a long live word past the end of a shorter utterance
        fill(short, T_w - 1, 100)
    b_t = {k: T(v) for k, v in batch.items()}
    want = ref.forward_infer(oracle_sd, b_t["word_tokens"], (b_t["keys"], b_t["values"], b_t["key_map"], b_t["pinyin"], b_t["pinyin_map"]),
                             b_t["pron_modified"], z_p=lambda B_, T4: T(synth.noise(9, B_, T4)))
    T_mel = want["mel_out"].shape[1]
    z = T(synth.noise(9, B, T_mel // 4))
    first = None
    for rep in range(12):
        got = _run(acoustic, batch, z=z)
        cur = {k: got[k].cpu() for k in ("dict_attn", "pron_attn", "word_encoder_out", "mel_out", "mel2word")}
        if first is None:
            first = cur
            assert torch.equal(cur["mel2word"], want["mel2word"])
            assert (cur["dict_attn"] - want["dict_attn"]).abs().max() <= 1e-5
            assert (cur["pron_attn"] - want["pron_attn"]).abs().max() <= 1e-5
            assert (cur["word_encoder_out"] - want["word_encoder_out"]).abs().max() <= 1e-4
            assert (cur["mel_out"] - want["mel_out"]).abs().max() <= 1e-3
            assert torch.allclose(cur["dict_attn"][1, 0, :, 5], torch.full((L,), 1.0 / L), atol=1e-6)
        else:
            for k in cur:
                assert torch.equal(cur[k], first[k]), (rep, k)


# ------------------------------------------------------------------------------------------------ FFT blocks
@pytest.mark.gpu
@pytest.mark.parametrize("which", ["dec", "enc"])
def test_g8_fft_blocks_vs_reference_golden(golden_dir, which):
    """dict_tts_amd.fft.FFTBlocks (dtts_fft_blocks_forward) vs the reference's FFTBlocks outputs: value-derived padding,
    positional embedding incl. the first-channel-zero quirk, bias-free attention, k**-0.5 GELU FFN with the LayerNorm
    bias of padded frames leaking through the SAME-padded conv, a 1-frame utterance; and with explicit padding_mask"""
    from dict_tts_amd import fft
    g = np.load(os.path.join(golden_dir, "g8_fft_blocks.npz"))
    cfg = gc.G8_CASES[which]
    m = fft.FFTBlocks(192, cfg["layers"], ffn_kernel_size=cfg["kernel_size"], num_heads=2, use_pos_embed=cfg["use_pos_embed"],
                      use_last_norm=cfg["use_last_norm"], hparams={})
    m.load_state_dict({k: T(v) for k, v in synth.fft_blocks_state_dict(gc.SEED, 192, **cfg).items()})
    x, lens = gc.g8_inputs(which)
    y = m(T(x)).cpu().numpy()
    want = g[which + ".out"]
    assert y.shape == want.shape
    assert np.abs(y - want).max() <= 1e-4, np.abs(y - want).max()
    for b, n in enumerate(lens):
        assert not (y[b, n:] != 0).any()
    pm = T(np.arange(x.shape[1])[None, :] >= lens[:, None])
    y2 = m(T(x), padding_mask=pm).cpu().numpy()
    assert np.array_equal(y, y2)


@pytest.mark.gpu
def test_fft_blocks_long_batch_vs_oracle_and_missing_weight():
    """decoder-sized problem (B=6, T=700 mel frames, ragged) against the oracle; a missing tensor is named"""
    from oracle import fft_blocks_ref as fref
    from dict_tts_amd import fft
    cfg = gc.G8_CASES["dec"]
    sd = synth.fft_blocks_state_dict(gc.SEED + 1, 192, **cfg)
    m = fft.FFTBlocks(192, cfg["layers"], ffn_kernel_size=9, num_heads=2, hparams={})
    m.load_state_dict({k: T(v) for k, v in sd.items()})
    x = synth.randn(gc.SEED, "fft.long.x", (6, 700, 192), 1.0)
    lens = [700, 512, 333, 64, 65, 7]
    for b, n in enumerate(lens):
        x[b, n:] = 0
    want = fref.fft_blocks({k: T(v) for k, v in sd.items()}, T(x), num_heads=2, kernel_size=9)
    got = m(T(x)).cpu()
    assert (got - want).abs().max() <= 2e-4, float((got - want).abs().max())
    bad = dict(sd)
    del bad["layers.2.op.ffn.ffn_2.bias"]
    m2 = fft.FFTBlocks(192, cfg["layers"], ffn_kernel_size=9, num_heads=2, hparams={})
    with pytest.raises(RuntimeError, match="layers.2.op.ffn.ffn_2.bias"):
        m2.load_state_dict({k: T(v) for k, v in bad.items()})


@pytest.mark.gpu
@pytest.mark.parametrize("n_utt", [2, 24])
def test_exact_fp32_decoder_configuration(oracle_sd, n_utt):
    """dtts_config.decoder_fp32 = 1: prior flow, conditioning, g_pre_net and the decoder WaveNet on the fp32-grade engines instead of the
    split-bf16 ones — the GATED form of the convolution kernels (tanh * sigmoid epilogue, two co-tiles per wave): 2 utterances run the
    few-row kernel (conv1d_short_kernel, gated, three-piece bf16 products), 24 utterances the generic one.  Same gates as the default
    configuration, and a tighter mel error than it (no 2^-16 products anywhere)."""
    from dict_tts_amd import abi, hparams, model
    from oracle import dict_tts_ref as ref
    cfg = hparams.fill_abi_config(abi.default_config(), {}, None, n_phone=6)
    cfg.decoder_fp32 = 1
    m = model.PortaSpeech_dict(hparams={}, ctx=abi.Context(cfg))
    m.load_state_dict({k: T(v) for k, v in synth.dict_tts_state_dict(gc.SEED, n_phone=6).items()}, strict=True)
    st = synth.biaobei_struct()
    batch = synth.make_batch(st["sentences"][:n_utt], 77)
    b = {k: T(v) for k, v in batch.items()}
    want = ref.forward_infer(oracle_sd, b["word_tokens"], (b["keys"], b["values"], b["key_map"], b["pinyin"], b["pinyin_map"]),
                             b["pron_modified"], z_p=lambda B, T4: T(synth.noise(78, B, T4)))
    T_mel = want["mel_out"].shape[1]
    got = _run(m, batch, z=T(synth.noise(78, n_utt, T_mel // 4)))
    assert torch.equal(got["mel2word"].cpu(), want["mel2word"])
    err = float((got["mel_out"].cpu() - want["mel_out"]).abs().max())
    assert err <= 1e-4, err   # the gate is 1e-3; the fp32-grade decoder measures ~1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 3, 40, 320])
def test_fft_blocks_short_sequences_every_contraction_split(B):
    """conv1d_short_kernel (T <= 64: the word encoder, B = 1): the launcher splits each output tile's contraction over 4, 2 or 1 waves
    depending on how many workgroups the layer makes (B = 1 / 3 -> 4-way everywhere, B = 40 -> 2-way for the wide layers, B = 320 -> no
    split for them), partial sums meeting in LDS in a fixed order.  Ragged lengths around the 32-row tile edge (1, 31, 32, 33, 64), a tile
    entirely beyond its utterance, against the oracle; and bit-repeatable."""
    from oracle import fft_blocks_ref as fref
    from dict_tts_amd import fft
    cfg = gc.G8_CASES["enc"]
    sd = synth.fft_blocks_state_dict(gc.SEED + 2, 192, **cfg)
    m = fft.FFTBlocks(192, cfg["layers"], ffn_kernel_size=cfg["kernel_size"], num_heads=2, use_pos_embed=cfg["use_pos_embed"],
                      use_last_norm=cfg["use_last_norm"], hparams={})
    m.load_state_dict({k: T(v) for k, v in sd.items()})
    x = synth.randn(gc.SEED, f"fft.short.x{B}", (B, 64, 192), 1.0)
    lens = [(64, 33, 32, 31, 1, 47, 17)[b % 7] for b in range(B)]
    for b, n in enumerate(lens):
        x[b, n:] = 0
    want = fref.fft_blocks({k: T(v) for k, v in sd.items()}, T(x), num_heads=2, kernel_size=cfg["kernel_size"],
                           use_pos_embed=cfg["use_pos_embed"], use_last_norm=cfg["use_last_norm"])
    got = m(T(x)).cpu()
    assert (got - want).abs().max() <= 1e-4, float((got - want).abs().max())
    for b, n in enumerate(lens):
        assert not (got[b, n:] != 0).any()
    assert torch.equal(m(T(x)).cpu(), got)


@pytest.mark.gpu
def test_abi_misuse_fails_loudly(voc_sd):
    """error convention of the C ABI: negative code + message, surfaced as abi.DttsError; no silent fallback"""
    from dict_tts_amd import hparams, model
    m = model.PortaSpeech_dict(hparams={})
    with pytest.raises(RuntimeError, match="load_state_dict"):
        m((torch.zeros(1, 3, dtype=torch.int64), None), None, None, None, None, (None,) * 5, infer=True)
    m.load_state_dict({k: T(v) for k, v in synth.dict_tts_state_dict(gc.SEED, n_phone=6).items()})
    s = torch.cuda.current_stream().cuda_stream
    buf = torch.zeros(64, device="cuda")
    with pytest.raises(abi.DttsError, match="before a successful dtts_text2mel_encode"):
        m.ctx.text2mel_decode(buf.data_ptr(), buf.data_ptr(), s)
    with pytest.raises(abi.DttsError, match="before encode"):
        m.ctx.fetch(abi.OUT_DUR, buf.data_ptr(), s)
    with pytest.raises(abi.DttsError, match="bad argument"):      # L_k beyond the kernel's 1024-row limit
        m.ctx.text2mel_encode(buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), None,
                              None, 1, 2, 2000, 4, s)
    with pytest.raises(abi.DttsError, match="dtts_dict_table_upload"):
        m.ctx.text2mel_encode_ids(buf.data_ptr(), buf.data_ptr(), None, None, 1, 2, 8, 4, s)
    with pytest.raises(abi.DttsError, match="vocoder weights not finalized"):   # acoustic-only handle has no vocoder
        m.ctx.hifigan_forward(buf.data_ptr(), None, 1, 4, buf.data_ptr(), s)
    from dict_tts_amd import vocoder
    v = vocoder.HifiGAN(state_dict=voc_sd, config=synth.hifigan_config())
    with pytest.raises(abi.DttsError, match="utterances per call"):             # the fused kernels' tile table is sized per utterance
        v.ctx.hifigan_forward(buf.data_ptr(), None, 4096, 4, buf.data_ptr(), s)
    with pytest.raises(abi.DttsError, match="FFT block weights not finalized"):
        m.ctx.fft_blocks_forward(buf.data_ptr(), None, None, 0, 1, 4, buf.data_ptr(), s)
    with pytest.raises(NotImplementedError):
        m((torch.zeros(1, 3, dtype=torch.int64), None), None, None, None, None, (None,) * 5, infer=False)
    with pytest.raises(NotImplementedError, match="use_post_glow"):
        hparams.fill_abi_config(abi.default_config(), {"use_post_glow": True})


@pytest.mark.gpu
def test_single_call_forward_equals_two_phase(acoustic):
    """dtts_text2mel_forward (one call, outputs in capacity layout) == encode + decode + fetch, bit for bit on the
    rows it writes; rows >= T_mel untouched; too small a capacity is an error; z_p = NULL draws N(0,1) on the device"""
    batch = synth.biaobei_batch(3, 3, gc.SEED)
    b = {k: T(v).cuda() for k, v in batch.items()}
    B, T_w = b["word_tokens"].shape
    L_k, P = b["keys"].shape[2], b["pinyin"].shape[2]
    s = torch.cuda.current_stream().cuda_stream
    ptr = lambda t: t.data_ptr()
    m2w_np = synth.teacher_mel2word(batch["word_tokens"], 6, 3)
    m2w = T(m2w_np).cuda()
    T_mel = acoustic.ctx.text2mel_encode(ptr(b["word_tokens"]), ptr(b["keys"]), ptr(b["values"]), ptr(b["key_map"]), ptr(b["pinyin"]),
                                         ptr(b["pinyin_map"]), ptr(b["pron_modified"]), (ptr(m2w), m2w.shape[1]), B, T_w, L_k, P, s)
    z = T(synth.noise(3, B, T_mel // 4)).cuda()
    want = torch.empty(B, T_mel, 80, device="cuda")
    acoustic.ctx.text2mel_decode(ptr(z), ptr(want), s)
    want_dur = torch.empty(B, T_w, device="cuda")
    acoustic.ctx.fetch(abi.OUT_DUR, ptr(want_dur), s)
    cap, zc = T_mel + 37, T_mel // 4 + 5
    zp = torch.full((B, 16, zc), 99.0, device="cuda")
    zp[:, :, : T_mel // 4] = z
    got = torch.full((B, cap, 80), -7.0, device="cuda")
    pron, dur = torch.empty(B, T_w, P, device="cuda"), torch.empty(B, T_w, device="cuda")
    t = acoustic.ctx.text2mel_forward(ptr(b["word_tokens"]), ptr(b["keys"]), ptr(b["values"]), ptr(b["key_map"]), ptr(b["pinyin"]),
                                      ptr(b["pinyin_map"]), ptr(b["pron_modified"]), (ptr(m2w), m2w.shape[1]), ptr(zp), zc, B, T_w, L_k,
                                      P, ptr(got), cap, ptr(pron), ptr(dur), s)
    torch.cuda.synchronize()
    assert t == T_mel
    assert torch.equal(got[:, :T_mel], want) and bool((got[:, T_mel:] == -7.0).all())
    assert torch.equal(dur, want_dur)
    with pytest.raises(abi.DttsError, match="exceed the capacity"):
        acoustic.ctx.text2mel_forward(ptr(b["word_tokens"]), ptr(b["keys"]), ptr(b["values"]), ptr(b["key_map"]), ptr(b["pinyin"]),
                                      ptr(b["pinyin_map"]), ptr(b["pron_modified"]), (ptr(m2w), m2w.shape[1]), ptr(zp), zc, B, T_w, L_k,
                                      P, ptr(got), T_mel - 4, None, None, s)
    # no prior sample given: drawn on the device; two calls draw different noise, both finite and mel-like
    a1 = torch.empty(B, cap, 80, device="cuda")
    a2 = torch.empty(B, cap, 80, device="cuda")
    for dst in (a1, a2):
        acoustic.ctx.text2mel_forward(ptr(b["word_tokens"]), ptr(b["keys"]), ptr(b["values"]), ptr(b["key_map"]), ptr(b["pinyin"]),
                                      ptr(b["pinyin_map"]), ptr(b["pron_modified"]), (ptr(m2w), m2w.shape[1]), None, 0, B, T_w, L_k, P,
                                      ptr(dst), cap, None, None, s)
    torch.cuda.synchronize()
    assert torch.isfinite(a1[:, :T_mel]).all() and not torch.equal(a1[:, :T_mel], a2[:, :T_mel])
    assert abs(float(a1[:, :T_mel].std()) - float(want.std())) < 0.25 * float(want.std())


@pytest.mark.gpu
def test_integer_durations_exact_over_seeds(acoustic, oracle_sd):
    """the duration path must round exactly as the CPU does (round(exp(dur) - 1) on fp32 values computed by different
    hardware): eight different batches / gloss embeddings / sentence sets, every mel2word entry equal"""
    from oracle import dict_tts_ref as ref
    st = synth.biaobei_struct()
    for k in range(8):
        sents = [st["sentences"][(17 * k + 5 * i) % len(st["sentences"])] for i in range(4)]
        batch = synth.make_batch(sents, 1000 + k, pron_every=2 + k % 3)
        b = {key: T(v) for key, v in batch.items()}
        want = ref.forward_infer(oracle_sd, b["word_tokens"], (b["keys"], b["values"], b["key_map"], b["pinyin"], b["pinyin_map"]),
                                 b["pron_modified"], z_p=lambda B, T4, k=k: T(synth.noise(50 + k, B, T4)))
        T_mel = want["mel_out"].shape[1]
        got = _run(acoustic, batch, z=T(synth.noise(50 + k, 4, T_mel // 4)))
        assert torch.equal(got["mel2word"].cpu(), want["mel2word"]), k
        # log-durations of magnitude ~3 through four fp32 FFT blocks: summation-order noise (the short-sequence kernels split the
        # contraction over waves) measures 3e-6 .. 1.4e-5 over these batches; the same 1e-4 bound as the golden-fixture tests
        assert (got["dur"].cpu() - want["dur"]).abs().max() <= 1e-4
        assert (got["mel_out"].cpu() - want["mel_out"]).abs().max() <= 1e-3


def test_two_product_upsampler_option_stays_inside_the_gate(voc_sd, oracle_voc_sd, monkeypatch):
    """dtts_config.tune_flags bit 13 (off by default): ups.1 on fp16 operands with two MFMA products instead of three bf16 ones (vconv.hip H2) —
    still inside the waveform gate (7e-5), and different from the default mode's result (the option really ran)."""
    from dict_tts_amd import vocoder
    from oracle import hifigan_ref as href
    mel = synth.random_mel(21, 96, "h2")
    want = href.spec2wav(oracle_voc_sd, synth.hifigan_config(), mel).numpy()
    base = vocoder.HifiGAN(state_dict=voc_sd, config=synth.hifigan_config(), precision="f16").spec2wav(mel)
    opt = vocoder.HifiGAN(state_dict=voc_sd, config={**synth.hifigan_config(), "dtts_tune_flags": 8192}, precision="f16").spec2wav(mel)
    wave_gate(base, want)
    wave_gate(opt, want)
    assert not np.array_equal(base, opt) and rms(base - want) < rms(opt - want)


def test_fp16_inter_iteration_stream_vs_fp32_stream(voc_sd, oracle_voc_sd):
    """Round 6: the residual stream between the three iterations of the per-iteration ResBlock kernels (C >= 128, k >= 7) is fp16 by
    default; dtts_config.tune_flags bit 15 restores round 5's fp32 stream.  fp16(x) is the value the next iteration's MFMA operand was rounded to
    anyway, so the two modes differ only through the residual add: both inside the waveform gate in the census fixture (0 clamped), the fp32 stream
    the more exact one, the fp16 stream below the adoption bound the verdict set (8e-5), the two results different (the option really ran) — and an
    overflow is seen by the always-on detector in BOTH modes (the stored stream does not saturate: +-inf reaches conv_post)."""
    from dict_tts_amd import vocoder
    from oracle import hifigan_ref as href
    cfg = synth.hifigan_config()
    mels = [synth.random_mel(31 + i, 72 + 24 * i, f"s16_{i}") for i in range(3)]
    want = [href.spec2wav(oracle_voc_sd, cfg, m).numpy() for m in mels]
    v16 = vocoder.HifiGAN(state_dict=voc_sd, config=cfg, precision="f16", range_guard=True)
    v32 = vocoder.HifiGAN(state_dict=voc_sd, config={**cfg, "dtts_tune_flags": 1 << 15}, precision="f16", range_guard=True)
    w16, w32 = v16.spec2wav_batch(mels), v32.spec2wav_batch(mels)
    e16 = e32 = 0.0
    for a16, a32, ref in zip(w16, w32, want):
        wave_gate(a16, ref)
        wave_gate(a32, ref)
        assert not np.array_equal(a16, a32)
        e16, e32 = max(e16, rms(a16 - ref)), max(e32, rms(a32 - ref))
    print(f"\n[fp16 stream] waveform RMS error: fp32 stream {e32:.3e}, fp16 stream {e16:.3e} (gate 1e-4)")
    assert e32 < e16 <= 8e-5, (e32, e16)
    # one utterance alone == the same utterance inside the ragged batch, in the fp16-stream mode too (rows of other utterances never mix in)
    assert np.array_equal(v16.spec2wav(mels[1]), w16[1])
    hot = mels[0] * 3e6
    for tune in (0, 1 << 15):
        v = vocoder.HifiGAN(state_dict=voc_sd, config={**cfg, "dtts_tune_flags": tune}, precision="f16")
        with pytest.raises(abi.DttsError, match="overflowed"):
            v.spec2wav(hot)


def test_fp16_validity_is_decided_statically_and_checked_on_every_call(voc_sd, oracle_voc_sd):
    """fp16 ResBlock operands overflow where the reference computes in fp32 (hifigan.py:51-58).  Whether DTTS_VOC_F16 is valid
    is a DECISION: a static bound from the folded weights (dtts_vocoder_fp16_bound) + the conv_post epilogue's always-on detector
    (dtts_vocoder_nonfinite).  No call returns garbage silently: AUTO redoes an overflowed call in DTTS_VOC_BF16X3 on the FIRST call,
    an explicit 'f16' raises; a small-gain generator is PROVEN safe; a generator whose propagated RMS already nears the fp16 limit never
    starts in fp16."""
    import warnings
    from dict_tts_amd import vocoder
    from oracle import hifigan_ref as href
    cfg = synth.hifigan_config()
    mel = synth.random_mel(11, 40, "guard")
    # (1) the healthy synthetic checkpoint: the worst case is astronomically loose (not provable), the RMS estimate is harmless ->
    # fp16 under the detector; calls stay in fp16 and the detector stays silent
    v0 = vocoder.HifiGAN(state_dict=voc_sd, config=cfg)
    wc, est = v0.fp16_bound
    assert v0.precision == abi.VOC_F16 and v0.fp16_status == "checked" and wc > 65504 and est * 16 < 65504, (wc, est)
    w0 = v0.spec2wav(mel)
    assert v0.precision == abi.VOC_F16 and np.isfinite(w0).all() and v0.ctx.vocoder_nonfinite() == 0
    wave_gate(w0, href.spec2wav(oracle_voc_sd, cfg, mel).numpy())
    # the bound is affine in the mel range and grows with it
    assert v0.ctx.vocoder_fp16_bound(12.0)[0] > wc > v0.ctx.vocoder_fp16_bound(0.0)[0] > 0
    # (2) overflow depends on the INPUT too: a mel far outside the stated range (x 3e6) overflows the healthy checkpoint.  FIRST call:
    # detected, redone in bf16x3, the object keeps that mode — equal to an explicit bf16x3 vocoder, finite
    hot = mel * 3e6
    with warnings.catch_warnings(record=True) as ws:
        warnings.simplefilter("always")
        w_hot = v0.spec2wav(hot)
    assert v0.precision == abi.VOC_BF16X3 and v0.fp16_status == "rejected" and any("overflowed" in str(w.message) for w in ws)
    assert np.isfinite(w_hot).all()
    assert np.array_equal(w_hot, vocoder.HifiGAN(state_dict=voc_sd, config=cfg, precision="bf16x3").spec2wav(hot))
    # explicit fp16, no census: the same call RAISES (round 4: "the unguarded calls return garbage without a word"), and the raw batch
    # path shows what the detector did: poisoned samples (NaN, never a plausible +-1) and a non-zero count behind the synchronisation
    v1 = vocoder.HifiGAN(state_dict=voc_sd, config=cfg, precision="f16")
    assert v1.fp16_status == "checked"
    with pytest.raises(abi.DttsError, match="overflowed"):
        v1.spec2wav(hot)
    raw = v1.forward_batch(T(hot[None]).cuda())          # check=False: the pipelined callers' path
    torch.cuda.synchronize()
    assert v1.overflowed() and v1.ctx.vocoder_nonfinite() > 0 and not np.isfinite(raw.cpu().numpy()).all()
    assert np.isfinite(v1.spec2wav(mel)).all() and not v1.overflowed()       # a healthy call afterwards: fine, nothing new counted
    # (3) weights that overflow on ordinary input.  convs1 of the first stage's k = 3 / k = 7 ResBlocks x300 (the third iteration's
    # activations reach ~1e7), ups.1 x1e-6 brings the stream back to O(1) so that the waveform is not a saturated tanh
    big = {k: (v * 300.0 if (k.startswith("resblocks.0.") or k.startswith("resblocks.1.")) and "convs1" in k and k.endswith("weight_g")
               else (v * 1e-6 if k == "ups.1.weight_g" else v)) for k, v in voc_sd.items()}
    want = href.spec2wav(href.fold_weight_norm(big), cfg, mel).numpy()
    with warnings.catch_warnings(record=True) as ws:
        warnings.simplefilter("always")
        v3 = vocoder.HifiGAN(state_dict=big, config=cfg)          # AUTO: rejected STATICALLY (the RMS estimate nears the limit) ...
    assert v3.precision == abi.VOC_BF16X3 and v3.fp16_status == "rejected" and any("not valid for this checkpoint" in str(w.message) for w in ws)
    w_auto = v3.spec2wav(mel)
    assert np.array_equal(w_auto, vocoder.HifiGAN(state_dict=big, config=cfg, precision="bf16x3").spec2wav(mel))
    assert rms(w_auto - want) <= 1e-4, rms(w_auto - want)
    v2 = vocoder.HifiGAN(state_dict=big, config=cfg, precision="f16")   # ... demanded explicitly: the first call raises
    with pytest.raises(abi.DttsError, match="overflowed"):
        v2.spec2wav(mel)
    # the census counter through the C ABI still works (the parity tests' mode), and raises with range_guard=True
    v2.ctx.vocoder_range_guard(True)
    v2.forward_batch(T(mel[None]).cuda())
    n = v2.ctx.vocoder_clamped(torch.cuda.current_stream().cuda_stream)
    assert n > 0 and v2.ctx.vocoder_clamped(torch.cuda.current_stream().cuda_stream) == 0   # (reset by the first read)
    with pytest.raises(abi.DttsError, match="exceeded the fp16 range"):
        vocoder.HifiGAN(state_dict=big, config=cfg, precision="f16", range_guard=True).spec2wav(mel)
    # (4) a small-gain generator (every ResBlock convolution x1e-3: the residual branches barely add; the upsamplers x0.1): the worst
    # case itself stays below 65504 for |mel| <= 6 (~1.0e3; with the upsamplers unscaled it is 1.0e7) -> PROVEN, statically, with no call made
    small = {k: (v * 1e-3 if k.startswith("resblocks.") and k.endswith("weight_g") else (v * 0.1 if k.startswith("ups.") and k.endswith("weight_g") else v))
             for k, v in voc_sd.items()}
    v4 = vocoder.HifiGAN(state_dict=small, config=cfg)
    assert v4.precision == abi.VOC_F16 and v4.fp16_status == "proven" and v4.fp16_bound[0] < 65504, v4.fp16_bound
    wave_gate(v4.spec2wav(mel), href.spec2wav(href.fold_weight_norm(small), cfg, mel).numpy())
    with pytest.raises(abi.DttsError, match="DTTS_VOC_F16 only"):
        vocoder.HifiGAN(state_dict=voc_sd, config=cfg, precision="bf16").ctx.vocoder_range_guard(True)
    assert vocoder.HifiGAN(state_dict=voc_sd, config=cfg, precision="bf16").fp16_status is None
    # (5) ADVICE r5: a SHARED context decides the mode — AUTO takes the context's precision (a bf16x3 context is not "fp16, proven"), an
    # explicit precision that disagrees with the context is an error
    from dict_tts_amd.hparams import HIFIGAN_DEFAULTS, fill_abi_config
    mk_ctx = lambda: abi.Context(fill_abi_config(abi.default_config(), None, {**HIFIGAN_DEFAULTS, **cfg}, vocoder_precision=abi.VOC_BF16X3))
    shared = vocoder.HifiGAN(state_dict=voc_sd, config=cfg, ctx=mk_ctx())
    assert shared.precision == abi.VOC_BF16X3 and shared.fp16_status is None and shared.fp16_bound is None
    assert np.array_equal(shared.spec2wav(mel), vocoder.HifiGAN(state_dict=voc_sd, config=cfg, precision="bf16x3").spec2wav(mel))
    with pytest.raises(abi.DttsError, match="shared context"):
        vocoder.HifiGAN(state_dict=voc_sd, config=cfg, ctx=mk_ctx(), precision="f16")


def _fp16_worst_case_peak(fsd, cfg, M):
    """numpy restatement of the static worst-case rule (context.hip: vocoder_fp16_analysis) evaluated DIRECTLY at |mel| <= M: the largest
    bound any fp16 ResBlock operand (leaky_relu(x) at the start of an iteration, leaky_relu(xt)) can reach, per-channel
    u_out[co] = |b[co]| + sum_ci u_in[ci] sum_k |w[co][ci][k]| (transposed convolutions: the largest output phase)."""
    g = lambda k: fsd[k].double().numpy()
    def conv(name, u):
        return np.abs(g(name + ".bias")) + np.abs(g(name + ".weight")).sum(2) @ u
    nk = len(cfg["resblock_kernel_sizes"])
    u = conv("conv_pre", np.full(cfg.get("audio_num_mel_bins", 80), float(M)))
    peak = 0.0
    for i, (r, k_n) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
        w, b, pad = np.abs(g(f"ups.{i}.weight")), np.abs(g(f"ups.{i}.bias")), (k_n - r) // 2      # [ci][co][k]
        x0 = np.max([b + np.einsum("i,io->o", u, w[:, :, (ph + pad) % r::r].sum(2)) for ph in range(r)], axis=0)
        us = 0.0
        for j in range(nk):
            x = x0.copy()
            for m in range(3):
                peak = max(peak, x.max())
                xt = conv(f"resblocks.{i * nk + j}.convs1.{m}", x)
                peak = max(peak, xt.max())
                x = x + conv(f"resblocks.{i * nk + j}.convs2.{m}", xt)
            us = us + x / nk
        u = us
    return peak


def test_fp16_static_bound_is_a_bound_when_bias_and_gain_peak_in_different_channels(voc_sd):
    """the peak over channels of per-channel bounds a_q + b_q M is CONVEX in M; the secant through the peaks at M = 0 and
    M = 1 (round 5) underestimates it beyond M = 1 when a bias-dominated channel sets both peaks and another channel's gain term overtakes
    it at the stated mel range.  dtts_vocoder_fp16_bound now returns max a_q + M max b_q: never below the directly evaluated peak."""
    from dict_tts_amd import vocoder
    from oracle import hifigan_ref as href
    cfg = synth.hifigan_config()
    small = {k: (v * 1e-3 if k.startswith("resblocks.") and k.endswith("weight_g") else (v * 0.1 if k.startswith("ups.") and k.endswith("weight_g") else v))
             for k, v in voc_sd.items()}
    base6 = _fp16_worst_case_peak(href.fold_weight_norm(small), cfg, 6.0)
    # one bias-dominated channel: the xt operand of the first ResBlock's first convolution, set to 0.6 x the gain-driven peak at M = 6
    small = dict(small)
    bias = small["resblocks.0.convs1.0.bias"].clone()
    bias[0] = 0.6 * base6
    small["resblocks.0.convs1.0.bias"] = bias
    fsd = href.fold_weight_norm(small)
    exact = {M: _fp16_worst_case_peak(fsd, cfg, M) for M in (0.0, 0.5, 1.0, 3.0, 6.0, 12.0)}
    secant6 = exact[0.0] + 6.0 * (exact[1.0] - exact[0.0])
    assert exact[6.0] > 1.2 * secant6, (exact, secant6)          # the case bites: round 5's formula would have reported less than the real peak
    v = vocoder.HifiGAN(state_dict=small, config=cfg, precision="f16")
    for M, want in exact.items():
        got = v.ctx.vocoder_fp16_bound(M)[0]
        assert got >= want * (1 - 1e-6), (M, got, want)
        assert got <= exact[0.0] + want + 1e-6, (M, got, want)   # ... and not absurdly loose: max a + M max b <= peak(0) + peak(M)
    assert v.fp16_status == "proven" and v.fp16_bound[0] >= exact[6.0] * (1 - 1e-6)


def test_fp16_static_margin_is_a_stated_threshold(voc_sd):
    """the static rejection rule is `propagated RMS estimate x EST_SIGMAS > 65504` (dict_tts_amd/vocoder.py: EST_SIGMAS = the
    crest factor — peak / RMS — the rule allows the operands).  A synthetic checkpoint scaled to sit 2 % BELOW the threshold starts in fp16
    ('checked', under the detector); 2 % ABOVE it never starts in fp16 ('rejected' -> DTTS_VOC_BF16X3), with the warning that says why."""
    import warnings
    from dict_tts_amd import vocoder
    cfg = synth.hifigan_config()
    H = vocoder.HifiGAN
    scaled = lambda g: {k: (v * g if k.startswith("resblocks.") and "convs1" in k and k.endswith("weight_g") else v) for k, v in voc_sd.items()}
    est_of = lambda g: H(state_dict=scaled(g), config=cfg, precision="f16").fp16_bound[1]
    lo, hi = 1.0, 64.0
    assert est_of(lo) * H.EST_SIGMAS < H.FP16_MAX < est_of(hi) * H.EST_SIGMAS
    for _ in range(14):                                   # geometric bisection of the gain at which estimate x EST_SIGMAS crosses 65504
        mid = (lo * hi) ** 0.5
        lo, hi = (mid, hi) if est_of(mid) * H.EST_SIGMAS < H.FP16_MAX else (lo, mid)
    g = (lo * hi) ** 0.5
    below, above = scaled(g * 0.98), scaled(g * 1.02)
    vb = H(state_dict=below, config=cfg)
    assert vb.precision == abi.VOC_F16 and vb.fp16_status == "checked" and vb.fp16_bound[1] * H.EST_SIGMAS < H.FP16_MAX, vb.fp16_bound
    with warnings.catch_warnings(record=True) as ws:
        warnings.simplefilter("always")
        va = H(state_dict=above, config=cfg)
    assert va.precision == abi.VOC_BF16X3 and va.fp16_status == "rejected" and any("not valid for this checkpoint" in str(w.message) for w in ws)
    # the rule is conservative on this family: just below the threshold the fp16 forward of an ordinary mel neither clamps nor overflows
    # (an RMS 16 x under the limit leaves the peaks far inside the range; tools/validate_checkpoint.py prints the measured crest)
    mel = synth.random_mel(5, 40, "margin")
    guard = H(state_dict=below, config=cfg, precision="f16", range_guard=True)
    assert np.isfinite(guard.spec2wav(mel)).all() and guard.ctx.vocoder_nonfinite() == 0


def test_auto_precision_falls_back_for_uncovered_generator_shapes(voc_sd):
    """a generator whose ResBlock widths the fused fp16 kernels do not cover (upsample_initial_channel = 384 -> 192 / 96 /
    48 / 24 channels) must still construct when the precision was not chosen explicitly: DTTS_VOC_BF16X3 on the generic kernel;
    an explicit precision='f16' keeps failing loudly."""
    import warnings
    from dict_tts_amd import vocoder
    from oracle import hifigan_ref as href
    cfg = dict(synth.hifigan_config(), upsample_initial_channel=384)
    sd = {k: T(v) for k, v in synth.hifigan_state_dict(gc.SEED, cfg=cfg).items()}
    with warnings.catch_warnings(record=True) as ws:
        warnings.simplefilter("always")
        v = vocoder.HifiGAN(state_dict=sd, config=cfg)
    assert v.precision == abi.VOC_BF16X3 and any("do not cover" in str(w.message) for w in ws)
    mel = synth.random_mel(3, 24, "odd")
    want = href.spec2wav(href.fold_weight_norm(sd), cfg, mel).numpy()
    wave_gate(v.spec2wav(mel), want)
    with pytest.raises(abi.DttsError, match="ResBlock widths"):
        vocoder.HifiGAN(state_dict=sd, config=cfg, precision="f16")


def test_device_prior_sample_seed(acoustic):
    """the DEVICE-side prior sample (dtts_text2mel_decode with z_p = NULL — what bench.py runs; the Python shim draws z_p
    from torch's RNG as the reference does) is seeded per context: two contexts draw different noise, the same seed reproduces a
    run (dtts_set_noise_seed), consecutive calls differ"""
    from dict_tts_amd import model
    st = synth.biaobei_struct()
    b = {k: T(v).cuda() for k, v in synth.make_batch(st["sentences"][:2], gc.SEED).items()}
    other = model.PortaSpeech_dict(hparams={})
    other.load_state_dict({k: T(v) for k, v in synth.dict_tts_state_dict(gc.SEED, n_phone=6).items()}, strict=True)
    s = torch.cuda.current_stream().cuda_stream
    ptr = lambda t: t.data_ptr()

    def draw(m):
        B, T_w = b["word_tokens"].shape
        T_mel = m.ctx.text2mel_encode(ptr(b["word_tokens"]), ptr(b["keys"]), ptr(b["values"]), ptr(b["key_map"]), ptr(b["pinyin"]),
                                      ptr(b["pinyin_map"]), ptr(b["pron_modified"]), None, B, T_w, b["keys"].shape[2], b["pinyin"].shape[2], s)
        mel = torch.empty(B, T_mel, 80, device="cuda")
        m.ctx.text2mel_decode(None, mel.data_ptr(), s)
        torch.cuda.synchronize()
        return mel

    a1, o1 = draw(acoustic), draw(other)
    assert a1.shape == o1.shape and torch.isfinite(a1).all() and not torch.equal(a1, o1)   # different contexts: different z_p
    acoustic.ctx.set_noise_seed(77)
    other.ctx.set_noise_seed(77)
    a2, a3, o2 = draw(acoustic), draw(acoustic), draw(other)
    assert torch.equal(a2, o2) and not torch.equal(a2, a3)                                  # same seed: same first draw; calls differ



# ------------------------------------------------------------------------------------------------ N > 1 control flow on one GPU
def test_bench_two_ranks_on_one_device_testset_sharding():
    """bench.py --gpus 2 --workload testset (BASELINE configs[2]: all 200 test sentences, utterance i -> rank i mod N in
    batches <= 60, tasks/tts/tts_base.py:148-151) launched with torch.distributed.run as the driver does, both ranks mapped to
    cuda:0 (DTTS_BENCH_ONE_DEVICE=1, gloo): rendezvous, shard_indices, the (B, T_mel) exchange + padded mel all-gather, barriers,
    max-over-ranks timing and the frame all-reduce all execute.  The two ranks together must produce the frames of the 1-rank
    pass over the same 200 sentences to within 0.5 %: an utterance's durations depend slightly on its batch (the collater puts
    key_map = pinyin_map = 1 on the LAST row of the padded batch, dataset_utils.py:288-300, which only the longest utterance's
    EOS occupies, and the longest utterance keeps the <= 3 frames_multiple pad frames), so the totals are close, not equal."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "1", "--warmup", "1", "--workload", "testset", "--no-cpu-baseline", "--no-side"]

    def run(cmd, env):
        r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])

    one = run([sys.executable, "bench.py", "--gpus", "1"] + common, dict(os.environ))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, DTTS_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    two = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), "bench.py", "--gpus", "2"] + common, env)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert one["config"]["batches_per_step"] == 4 and two["config"]["batches_per_step"] == 2      # 200 / 60 and 200 / 120 chunks
    f1 = one["value"] * one["ms_per_step"] * 1e-3
    f2 = two["value"] * two["ms_per_step"] * 1e-3
    assert abs(f1 - f2) <= 0.005 * f1 and f1 > 200 * 100, (f1, f2)
    g = two["mel_allgather"]
    assert g["enabled"] and g["calls"] == 3 * 2 and g["disabled_reason"] is None          # (range-guard pass + warm-up + timed) x 2 chunks
    meta = g["last_meta"]                                   # last chunk: sentences 120..199 -> 40 per rank
    assert len(meta) == 2 and meta[0][0] == meta[1][0] == 40 and min(meta[0][1], meta[1][1]) > 100
    assert "+allgather(mel)" in two["config"]["parallelism"] and one["mel_allgather"]["enabled"] is False


def test_bench_eight_ranks_on_one_device_testset_sharding():
    """BASELINE configs[2] AS STATED — the 200-sentence test set over EIGHT ranks (utterance i -> rank i mod 8: 25 per rank, one chunk;
    tasks/tts/tts_base.py:148-151) with the mel all-gather — launched exactly as the driver launches an 8-GPU run, all ranks on cuda:0 under
    the one-device hook (gloo): the 8-rank rendezvous, split, shape exchange (started right behind encode, read after the vocoder has been
    enqueued), padded gather, barriers, max-over-ranks timing and frame all-reduce run for the first time at the stated width (VERDICT r5
    weak #1).  Frames within 1 % of the 1-rank pass over the same sentences; the gather saw 8 rows of 25 utterances."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "1", "--warmup", "1", "--workload", "testset", "--no-cpu-baseline", "--no-side"]

    def run(cmd, env):
        r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0, r.stderr[-3000:]
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])

    one = run([sys.executable, "bench.py", "--gpus", "1"] + common, dict(os.environ))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, DTTS_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    eight = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                 "--master-port", str(port), "bench.py", "--gpus", "8"] + common, env)
    assert eight["n_gpus"] == 8 and eight["scaling"] == "strong" and eight["config"]["batches_per_step"] == 1      # 200 / (8 x 60): one chunk
    f1 = one["value"] * one["ms_per_step"] * 1e-3
    f8 = eight["value"] * eight["ms_per_step"] * 1e-3
    assert abs(f1 - f8) <= 0.01 * f1 and f1 > 200 * 100, (f1, f8)
    g = eight["mel_allgather"]
    assert g["enabled"] and g["disabled_reason"] is None and g["calls"] == 3                 # range-guard pass + warm-up + timed, one chunk each
    meta = g["last_meta"]
    assert len(meta) == 8 and all(m[0] == 25 for m in meta) and min(m[1] for m in meta) > 100, meta
    assert len(eight["ranks_seen"]) == 8 and sorted(r["rank"] for r in eight["ranks_seen"]) == list(range(8))
    assert "+allgather(mel)" in eight["config"]["parallelism"]


def test_mel_allgather_is_off_the_critical_path_two_ranks_one_device():
    """the step with the mel all-gather and with --no-gather.  The shape exchange no longer blocks the host in front of the
    vocoder's launches (dict_tts_amd/shard.py:exchange_shapes; bench.py starts it right behind encode and reads it after the batch has been
    enqueued).  Under the one-device hook both ranks share cuda:0 and the gather itself runs through gloo on the HOST (a D2H of the mel, a
    CPU all-gather) — far more work than RCCL on device buffers — so the bound here is 5 % (measured x0.98 - x0.99 on three runs: the gathered step is not slower at all; printed); what the test
    pins is that the gathered step is not the SERIAL sum it was when the host waited for the decoder before launching the vocoder."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-side"]
    env = dict(os.environ, DTTS_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")

    def run(extra):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", str(port), "bench.py", "--gpus", "2"] + common + extra, cwd=root, env=env, capture_output=True,
                           text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])

    off, on = run(["--no-gather"]), run([])
    assert on["mel_allgather"]["enabled"] and on["mel_allgather"]["calls"] > 0 and not off["mel_allgather"]["enabled"]
    ratio = on["ms_per_step"] / off["ms_per_step"]
    print(f"\n[gather off the critical path] 2 ranks on one device: {off['ms_per_step']:.2f} ms/step without the gather, {on['ms_per_step']:.2f} with it "
          f"(x{ratio:.3f})")
    assert ratio < 1.05, (off["ms_per_step"], on["ms_per_step"])


def test_config5_b128_single_gpu_superset_vs_oracle(acoustic, oracle_sd):
    """BASELINE.json configs[4] AT ITS STATED SIZE on one GPU: B = 128 mixed-length utterances (synth.config5_sentences: the
    same definition bench.py --workload config5 shards over 4 ranks), all 7,030 dictionary entries resident, ids only — against the ORACLE
    on the tensors the reference's collater builds for the SAME 128-utterance batch (mel2word exact, pron_attn / dict_attn <= 1e-5, mel
    <= 1e-3, pinyin strings identical).  The four 32-utterance shards (utterance i -> rank i mod 4, tasks/tts/tts_base.py:148-151) then run
    one by one on the same context: the batch-invariant outputs of every utterance (log-durations, attention maps, pinyin) are BIT-identical
    in its shard and in the superset when the shard has the superset's padded T_w, and within 1e-5 otherwise."""
    from dict_tts_amd.model import decode_pinyin_ids
    from dict_tts_amd.shard import shard_indices
    from oracle import dict_tts_ref as ref
    full = synth.zh_dict_struct()
    ent = full["entries"]
    table = synth.dict_table(gc.SEED, ent)
    acoustic.upload_dict_table(table)
    sents = synth.config5_sentences(128, 55, ent)
    assert len(sents) == 128 and min(map(len, sents)) >= 6 and max(map(len, sents)) <= 60
    ib = synth.make_id_batch(sents, table, pron_every=3)
    tb = synth.make_batch(sents, gc.SEED, ent, pron_every=3)
    assert np.array_equal(ib["word_tokens"], tb["word_tokens"]) and (ib["L_k"], ib["P"]) == (tb["keys"].shape[2], tb["pinyin"].shape[2])
    b = {k: T(v) for k, v in tb.items()}
    want = ref.forward_infer(oracle_sd, b["word_tokens"], (b["keys"], b["values"], b["key_map"], b["pinyin"], b["pinyin_map"]),
                             b["pron_modified"], z_p=lambda B_, T4: T(synth.noise(34, B_, T4)))
    T_mel = want["mel_out"].shape[1]
    got = acoustic.forward_ids(T(ib["word_tokens"]), T(ib["entry_ids"]), T(ib["pron_modified"]), ib["L_k"], ib["P"],
                               z_p=T(synth.noise(34, 128, T_mel // 4)))
    assert (got["dur"].cpu() - want["dur"]).abs().max() <= 1e-4
    # integer durations: exact wherever the reference's own round() is numerically decided (as in the B = 60 test: among ~4.4 k words a
    # few sit within 5e-5 of an x.5 boundary, where two fp32 implementations cannot agree; they are counted and excluded)
    v = want["dur"].exp() - 1
    tie = ((v - v.floor() - 0.5).abs() < 5e-5) & (b["word_tokens"] > 0)
    gi, wi = (got["dur"].cpu().exp() - 1).round().clamp(min=0), v.round().clamp(min=0)
    print(f"\n[config5 B=128] words {int((b['word_tokens'] > 0).sum())}  n_ties {int(tie.sum())}  n_flips {int((gi != wi).sum())}  "
          f"flips outside ties {int(((gi != wi) & ~tie).sum())}")
    assert int(tie.sum()) <= 6 and torch.equal(gi[~tie], wi[~tie])
    if not torch.equal(got["mel2word"].cpu(), want["mel2word"]):   # a tie went the other way: compare the rest on the SAME durations
        want = ref.forward_infer(oracle_sd, b["word_tokens"], (b["keys"], b["values"], b["key_map"], b["pinyin"], b["pinyin_map"]),
                                 b["pron_modified"], mel2word=got["mel2word"].cpu(), z_p=lambda B_, T4: T(synth.noise(34, B_, T4)))
        got = acoustic.forward_ids(T(ib["word_tokens"]), T(ib["entry_ids"]), T(ib["pron_modified"]), ib["L_k"], ib["P"],
                                   z_p=T(synth.noise(34, 128, want["mel_out"].shape[1] // 4)))
    assert torch.equal(got["mel2word"].cpu(), want["mel2word"]) and got["mel_out"].shape == want["mel_out"].shape
    assert (got["pron_attn"].cpu() - want["pron_attn"]).abs().max() <= 1e-5
    assert (got["dict_attn"].cpu() - want["dict_attn"]).abs().max() <= 1e-5
    assert (got["mel_out"].cpu() - want["mel_out"]).abs().max() <= 1e-3
    for u in range(128):
        assert decode_pinyin_ids(got["pron_attn"][u], tb["pinyin"][u]) == ref.decode_pinyin(want["pron_attn"][u], b["pinyin"][u])
    del b, tb, want
    dur_all, pa_all = got["dur"].cpu(), got["pron_attn"].cpu()
    seen = []
    for r in range(4):
        (mine,) = shard_indices(128, r, 4, 32)
        assert mine == list(range(r, 128, 4))
        seen += mine
        sb = synth.make_id_batch([sents[i] for i in mine], table, pron_every=3)
        Tw_s = sb["word_tokens"].shape[1]
        # the SAME forced senses as in the superset (make_id_batch counts "every 3rd heteronym" across its batch)
        sb["pron_modified"] = np.ascontiguousarray(ib["pron_modified"][mine][:, :Tw_s])
        gs = acoustic.forward_ids(T(sb["word_tokens"]), T(sb["entry_ids"]), T(sb["pron_modified"]), sb["L_k"], sb["P"])
        # utterances that own the LAST padded column in either batch see the collater's all-ones key_map / pinyin_map row there
        # (dataset_utils.py:288-300) and are batch-dependent by the reference's own construction: compared are all the others
        n_s = (sb["word_tokens"] > 0).sum(1)
        keep = torch.from_numpy((n_s < Tw_s) & (n_s < ib["word_tokens"].shape[1]))
        assert int(keep.sum()) >= 28, r
        valid = torch.from_numpy(sb["word_tokens"] > 0)[keep]                      # (the padded columns' rows carry the last-row rule too)
        d_s, d_a = gs["dur"].cpu()[keep], dur_all[mine][:, :Tw_s][keep]
        p_s, p_a = gs["pron_attn"].cpu()[keep][valid], pa_all[mine][:, :Tw_s, :sb["P"]][keep][valid]
        if Tw_s == ib["word_tokens"].shape[1] and sb["P"] == ib["P"] and sb["L_k"] == ib["L_k"]:
            assert torch.equal(d_s, d_a) and torch.equal(p_s, p_a), r
        else:
            assert (d_s - d_a).abs().max() <= 1e-5 and (p_s - p_a).abs().max() <= 1e-5, (r, float((d_s - d_a).abs().max()), float((p_s - p_a).abs().max()))
    assert sorted(seen) == list(range(128))
    acoustic.upload_dict_table(synth.dict_table(gc.SEED))   # (the module-scoped model goes back to the Biaobei table)


def test_bench_config5_four_ranks_on_one_device_split():
    """bench.py --workload config5 (BASELINE configs[4]: B = 128 over 4 GPUs = 32 utterances per rank, full dictionary resident) launched as
    the driver launches an N = 4 run, all four ranks mapped to cuda:0 (DTTS_BENCH_ONE_DEVICE=1, gloo): rendezvous, the i mod 4 split,
    the mel all-gather of four shards, max-over-ranks timing and the frame all-reduce execute; the four shards together produce the frames
    of the single-rank B = 128 superset to within 1.5 % (durations depend slightly on the batch: the collater's last padded row, the
    synthetic forced-sense rule)."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "1", "--warmup", "1", "--workload", "config5", "--no-cpu-baseline", "--no-side"]

    def run(cmd, env):
        r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])

    one = run([sys.executable, "bench.py", "--gpus", "1"] + common, dict(os.environ))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, DTTS_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    four = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
                "--master-port", str(port), "bench.py", "--gpus", "4"] + common, env)
    assert one["n_gpus"] == 1 and four["n_gpus"] == 4 and four["scaling"] == "strong"
    assert one["config"]["batch_shapes_B_Tw_Lk"][0][0] == 128 and four["config"]["batch_shapes_B_Tw_Lk"][0][0] == 32
    f1 = one["value"] * one["ms_per_step"] * 1e-3
    f4 = four["value"] * four["ms_per_step"] * 1e-3
    # (0.64 % measured: besides the last padded row, make_id_batch forces "every 7th heteronym" counted ACROSS its batch, so a shard's
    # forced senses are not the superset's; test_config5_b128_... compares shard and superset on identical inputs, bit for bit)
    assert abs(f1 - f4) <= 0.015 * f1 and f1 > 128 * 6 * 10, (f1, f4)
    g = four["mel_allgather"]
    assert g["enabled"] and g["disabled_reason"] is None and len(g["last_meta"]) == 4 and all(mm[0] == 32 for mm in g["last_meta"])
    assert "dp4" in four["config"]["parallelism"] and "configs[4]" in four["config"]["workload"]


# ------------------------------------------------------------------------------------------------ memory safety
def _redzoned(t, pad_elems=4096, fill=float("nan")):
    """a copy of tensor t on the GPU whose storage has `pad_elems` sentinel elements before and after it: (view, check) where check()
    asserts that no kernel wrote into the borders"""
    flat = torch.full((t.numel() + 2 * pad_elems,), fill, dtype=t.dtype, device="cuda")
    view = flat[pad_elems:pad_elems + t.numel()].view(t.shape)
    view.copy_(t)
    sentinel = flat[:pad_elems].clone()

    def check():
        torch.cuda.synchronize()
        lo, hi = flat[:pad_elems], flat[pad_elems + t.numel():]
        same = lambda a: torch.equal(a.view(torch.int32 if a.element_size() == 4 else torch.int64),
                                     sentinel.view(torch.int32 if a.element_size() == 4 else torch.int64))
        assert same(lo) and same(hi), "a kernel wrote outside a caller-owned buffer"
    return view, check


@pytest.fixture(scope="module")
def dbg_pair(voc_sd):
    """the acoustic model and the benched vocoder mode in MEMORY-SAFETY mode (dtts_config.debug_redzone): every workspace buffer and weight
    pack between 4 KiB red zones, workspaces 0xFF(NaN)-filled before each forward"""
    from dict_tts_amd import model, vocoder
    m = model.PortaSpeech_dict(hparams={"dtts_debug_redzone": 1})
    m.load_state_dict({k: T(v) for k, v in synth.dict_tts_state_dict(gc.SEED, n_phone=6).items()}, strict=True)
    v = vocoder.HifiGAN(state_dict=voc_sd, config={**synth.hifigan_config(), "dtts_debug_redzone": 1}, precision="f16")
    return m, v


def _clean(ctx, what):
    n = ctx.debug_check(torch.cuda.current_stream().cuda_stream)
    assert n == 0, f"{what}: {ctx.last_error()}"


def test_memory_safety_harness_notices_a_damaged_red_zone(dbg_pair, voc_plain):
    """the check itself: one byte written just past a workspace buffer is reported (and named), and a release context refuses the call"""
    _, v = dbg_pair
    s = torch.cuda.current_stream().cuda_stream
    v.spec2wav(synth.random_mel(1, 8, "poke"))
    _clean(v.ctx, "before the poke")
    v.ctx.debug_poke(s)
    assert v.ctx.debug_check(s) == 1 and "AFTER vocoder workspace buffer #0" in v.ctx.last_error(), v.ctx.last_error()
    v.spec2wav(synth.random_mel(1, 8, "poke"))   # the next forward refills the workspace
    _clean(v.ctx, "after the refill")
    with pytest.raises(abi.DttsError, match="debug_redzone"):
        voc_plain.ctx.debug_check(s)


def test_memory_safety_vocoder_shapes(dbg_pair, voc_plain):
    """HifiGAN (DTTS_VOC_F16, the benched kernels) under red zones + NaN-poisoned workspaces on the shapes that stress the tile walkers: the
    70-utterance ragged batch (empty / 1-frame rows, several tiles per persistent workgroup), B = 1 short (half-size tiles), the 5,012-frame
    long form (1.28 M rows at the last stage), a B = 60 x 740 batch (the bench shape).  Per shape: no red zone damaged (no out-of-range
    WRITE), the caller's mel / wav borders untouched, the waveform finite, zero past each utterance and BIT-identical to the release
    context's (an out-of-range or stale READ that is consumed would read NaN here and something else there)."""
    _, v = dbg_pair
    rng = np.random.RandomState(7)
    many = [0, 1, 61, 2, 33] + [int(x) for x in rng.randint(0, 62, size=65)]
    cases = [("70 ragged utterances", many, 64), ("B=1 short", [37], 37), ("B=1, 3 frames", [3], 3), ("long form", [5012], 5012),
             ("bench shape", [int(x) for x in rng.randint(250, 741, size=59)] + [740], 740)]
    for name, lens, Tm in cases:
        B = len(lens)
        mel = np.zeros((B, Tm, 80), np.float32)
        for i, n in enumerate(lens):
            if n:
                mel[i, :n] = synth.random_mel(900 + i, n, f"ms{name}{i}") if n < 2000 else np.tile(synth.random_mel(900, 179, "ms.long"), (28, 1))[:n]
        mel_d, chk_mel = _redzoned(T(mel))
        wav_d, chk_wav = _redzoned(torch.zeros(B, Tm * 256))
        lens_d = torch.tensor(lens, dtype=torch.int32, device="cuda")
        s = torch.cuda.current_stream().cuda_stream
        v.ctx.hifigan_forward(mel_d.data_ptr(), lens_d.data_ptr(), B, Tm, wav_d.data_ptr(), s)
        _clean(v.ctx, name)
        chk_mel()
        chk_wav()
        got = wav_d.cpu().numpy()
        assert np.isfinite(got).all(), name
        for i, n in enumerate(lens):
            assert not got[i, n * 256:].any(), (name, i)
        want = voc_plain.forward_batch(T(mel).cuda(), lens_d).cpu().numpy()
        assert np.array_equal(got, want), name


def test_memory_safety_text2mel_shapes(dbg_pair, acoustic):
    """the acoustic model under the same mode: B = 60 at full size (tensor API and the resident-table id path), the 1000-character long form,
    B = 1 — red zones intact after encode and after decode, every output finite and BIT-identical to the release context's"""
    m, _ = dbg_pair
    st = synth.biaobei_struct()
    s = torch.cuda.current_stream().cuda_stream
    keys = ("mel_out", "dur", "word_encoder_out", "pron_attn", "dict_attn", "mel2word")

    def same(a, b, name):
        for k in keys:
            x, y = a[k].cpu(), b[k].cpu()
            assert x.shape == y.shape and torch.isfinite(x.float()).all(), (name, k)
            assert torch.equal(x, y), (name, k, float((x.float() - y.float()).abs().max()))

    ids1000 = [w for sent in st["sentences"] for w in sent][:1000]
    for name, sents, tf in (("B=60", st["sentences"][:60], None), ("B=1", [st["sentences"][5]], None), ("long form", [ids1000], (5, 5))):
        batch = synth.make_batch(sents, gc.SEED)
        m2w = None if tf is None else T(synth.teacher_mel2word(batch["word_tokens"], *tf))
        B = batch["word_tokens"].shape[0]
        # the same noise for both contexts: z_p is passed explicitly (its length from a first, untimed pass when the durations are predicted)
        T_mel = _run(acoustic, batch)["mel_out"].shape[1] if m2w is None else m2w.shape[1] + (-m2w.shape[1]) % 4
        z = T(synth.noise(77, B, T_mel // 4, "ms.z"))
        want = _run(acoustic, batch, z=z, mel2word=m2w)
        got = _run(m, batch, z=z, mel2word=m2w)
        _clean(m.ctx, name)
        same(got, want, name)
    # the resident dictionary (pre-projected rows, weight-pack red zones around the table) + ids
    table = synth.dict_table(gc.SEED)
    m.upload_dict_table(table)
    acoustic.upload_dict_table(table)
    ib = synth.make_id_batch(st["sentences"][:60], table)
    probe = acoustic.forward_ids(T(ib["word_tokens"]), T(ib["entry_ids"]), T(ib["pron_modified"]), ib["L_k"], ib["P"])
    z = T(synth.noise(78, 60, probe["mel_out"].shape[1] // 4, "ms.zi"))
    want = acoustic.forward_ids(T(ib["word_tokens"]), T(ib["entry_ids"]), T(ib["pron_modified"]), ib["L_k"], ib["P"], z_p=z)
    got = m.forward_ids(T(ib["word_tokens"]), T(ib["entry_ids"]), T(ib["pron_modified"]), ib["L_k"], ib["P"], z_p=z)
    _clean(m.ctx, "id path B=60")
    same(got, want, "id path B=60")


def test_memory_safety_other_kernel_families(voc_sd):
    """the generic convolution kernels (DTTS_VOC_BF16X3), the all-bf16 vocoder and the FFT-block stack under the same mode"""
    from dict_tts_amd import fft, vocoder
    s = torch.cuda.current_stream().cuda_stream
    lens = [5, 33, 17, 64, 1, 0]
    mels = [synth.random_mel(100 + i, n, f"rag{i}") for i, n in enumerate(lens) if n]
    for prec in ("bf16x3", "bf16"):
        v = vocoder.HifiGAN(state_dict=voc_sd, config={**synth.hifigan_config(), "dtts_debug_redzone": 1}, precision=prec)
        ref = vocoder.HifiGAN(state_dict=voc_sd, config=synth.hifigan_config(), precision=prec)
        got, want = v.spec2wav_batch(mels), ref.spec2wav_batch(mels)
        _clean(v.ctx, prec)
        for a, b in zip(got, want):
            assert np.isfinite(a).all() and np.array_equal(a, b), prec
    cfg = gc.G8_CASES["dec"]
    sd = {k: T(x) for k, x in synth.fft_blocks_state_dict(gc.SEED + 1, 192, **cfg).items()}
    x = synth.randn(gc.SEED, "fft.ms.x", (5, 300, 192), 1.0)
    pm = torch.zeros(5, 300, dtype=torch.bool)
    for b, n in enumerate([300, 257, 64, 33, 1]):
        x[b, n:] = 0
        pm[b, n:] = True
    outs = []
    for hp in ({"dtts_debug_redzone": 1}, {}):
        f = fft.FFTBlocks(192, cfg["layers"], ffn_kernel_size=9, num_heads=2, hparams=hp)
        f.load_state_dict(sd)
        outs.append(f(T(x), padding_mask=pm).cpu())
        if hp:
            _clean(f.ctx, "fft blocks")
    assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("flag,what", [(512, "all ResBlocks of a C <= 64 stage in one launch, private stage-sum strips under the fused conv_post"),
                                       (4096, "the first two ResBlocks of the C = 32 stage in one launch"),
                                       (16384, "512-row tiles for every k of the C = 64 whole-ResBlock kernel (the default takes 640 rows at k >= 7)")])
def test_optin_resblock_forms_are_bit_identical_to_the_default(voc_sd, voc_plain, flag, what):
    """round 4's measured-and-not-adopted ResBlock launch forms (dtts_config.tune_flags bits 9, 12, 14; LABNOTES round 4; the two-group kernel of
    bit 7 lives in the ablation build only since round 5) compute the SAME bits
    as the default launches — 70 ragged utterances (several tiles per persistent workgroup, odd tile counts, empty rows) and a B = 24 batch
    large enough for the stage-fused form to engage — and stay clean under the memory-safety mode"""
    from dict_tts_amd import vocoder
    v = vocoder.HifiGAN(state_dict=voc_sd, config={**synth.hifigan_config(), "dtts_tune_flags": flag, "dtts_debug_redzone": 1}, precision="f16")
    rng = np.random.RandomState(11)
    for lens, Tm in (([0, 1, 61, 2, 33] + [int(x) for x in rng.randint(0, 62, size=65)], 64), ([int(x) for x in rng.randint(200, 701, size=23)] + [700], 700)):
        B = len(lens)
        mel = np.zeros((B, Tm, 80), np.float32)
        for i, n in enumerate(lens):
            if n:
                mel[i, :n] = synth.random_mel(500 + i, n, f"opt{i}")
        lens_d = torch.tensor(lens, dtype=torch.int32, device="cuda")
        got = v.forward_batch(T(mel).cuda(), lens_d).cpu().numpy()
        _clean(v.ctx, what)
        want = voc_plain.forward_batch(T(mel).cuda(), lens_d).cpu().numpy()
        assert np.isfinite(got).all() and np.array_equal(got, want), what


def test_rccl_world1_gather_mels_and_ranks_seen_on_the_gpu():
    """the CUDA-tensor path of the data-parallel helpers through RCCL itself (backend 'nccl' on ROCm), which the world-2
    gloo tests on the CPU cannot reach: a world-size-1 nccl group on the one GPU, `gather_mels(None, None)` (a rank without a batch),
    `gather_mels(mel, lens)` with device tensors, `ranks_seen`, and the frame all-reduce bench.py does.  Runs in a child process (the process
    group is process-global)."""
    import subprocess
    import sys
    code = r'''
import os, socket, sys, json
import torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
from dict_tts_amd.shard import gather_mels, ranks_seen, group_device
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
assert group_device(dist) == dev
seen = ranks_seen(dist, device_index=0, device_uuid=str(torch.cuda.get_device_properties(dev).uuid))
assert len(seen) == 1 and seen[0]["rank"] == 0 and seen[0]["device_index"] == 0 and seen[0]["device_uuid"]
m0, l0, meta0 = gather_mels(None, None, dist)
assert m0 is None and l0 is None and meta0.tolist() == [[0, 0]]
g = torch.Generator(device="cpu").manual_seed(3)
mel = torch.randn(5, 37, 80, generator=g).to(dev)
lens = torch.tensor([37, 20, 1, 0, 36], dtype=torch.int32, device=dev)
side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):                       # bench.py gathers on a side stream
    ma, la, meta = gather_mels(mel, lens, dist)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
assert ma.is_cuda and ma.shape == (1, 5, 37, 80) and torch.equal(ma[0], mel) and torch.equal(la[0], lens) and meta.tolist() == [[5, 37]]
f = torch.tensor([12345], dtype=torch.int64, device=dev)
dist.all_reduce(f)
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert int(f.item()) == 12345 and float(t.item()) == 1.5
dist.barrier()
dist.destroy_process_group()
print(json.dumps({"ok": True, "seen": seen}))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and '"ok": true' in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


def test_results_do_not_depend_on_batch_composition(acoustic):
    """the arithmetic of every layer is fixed when its weights are packed — the contraction of a short-sequence
    convolution is always summed in the same P parts in the same order however many waves share them (conv1d.hip), the attention kernel is
    chosen from the padded T_w alone — so, for batches on the same side of the 128-word attention threshold, an utterance gets BIT-identical log-durations, encoder output, dictionary attention and (with the same z_p)
    mel whether it runs alone (B = 1: contraction over 4 waves) or as row u of a B = 60 batch (over 2 or 1).  Utterances are padded to the
    batch's T_w / L_k for the comparison's inputs only (the reference pads the same way, dataset_utils.py:264-330)."""
    st = synth.biaobei_struct()
    sents = st["sentences"][:60]
    full = synth.make_batch(sents, gc.SEED)
    r60 = _run(acoustic, full)
    lens60 = (full["word_tokens"] > 0).sum(1)
    for u in (0, 7, 31, 59):
        one = {k: v[u:u + 1] for k, v in full.items()}           # the same padded row, alone
        r1 = _run(acoustic, one)
        n = int(lens60[u])
        for k in ("dur", "word_encoder_out", "pron_attn"):
            a, b = r1[k][0, :n].cpu(), r60[k][u, :n].cpu()
            assert torch.equal(a, b), (u, k, float((a - b).abs().max()))
        assert torch.equal(r1["dict_attn"][0, 0].cpu()[:, :n], r60["dict_attn"][u, 0].cpu()[:, :n]), u
        # same integer durations: the batch row's frames are a prefix of the lone utterance's (alone, the <= 3 frames_multiple pad frames
        # repeat the last word; inside a batch the row is zero-padded)
        m60 = r60["mel2word"][u].cpu()
        k = int((m60 > 0).sum())
        assert k > 0 and torch.equal(r1["mel2word"][0].cpu()[:k], m60[:k]) and r1["mel2word"].shape[1] - k <= 3, u
    # LIMIT of the claim: the attention kernel is chosen from the batch's PADDED T_w — above 128 words the key-split kernel merges
    # its partial softmax sums in another order.  An utterance run at its own T_w <= 128 and the same utterance inside a batch padded to
    # T_w > 128 (one 140-character sentence beside it) therefore agree to fp32 rounding, not bit for bit: integer durations equal,
    # log-durations / encoder output within 1e-5 / 1e-4.  (Biaobei: T_w <= 29, every batch on the same side of the threshold.)
    long_sent = (st["sentences"][3] * 20)[:140]
    assert len(long_sent) == 140
    mixed = synth.make_batch([sents[0], long_sent], gc.SEED)
    assert mixed["word_tokens"].shape[1] > 128
    rm = _run(acoustic, mixed)
    alone = _run(acoustic, synth.make_batch([sents[0]], gc.SEED))
    n0 = int((mixed["word_tokens"][0] > 0).sum())
    assert (alone["dur"][0, :n0].cpu() - rm["dur"][0, :n0].cpu()).abs().max() <= 1e-5
    assert (alone["word_encoder_out"][0, :n0].cpu() - rm["word_encoder_out"][0, :n0].cpu()).abs().max() <= 1e-4
    k0 = int((rm["mel2word"][0] > 0).sum())
    assert k0 > 0 and torch.equal(alone["mel2word"][0].cpu()[:k0], rm["mel2word"][0].cpu()[:k0])
