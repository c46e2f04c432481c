"""CPU-only checks of the drop-in boundary: the C-ABI library builds/loads and exports every symbol declared in
include/dicttts_hip.h (no compute calls without a GPU), and the host-side mirrors behave like the reference's
plugin points."""
import os
import re

import numpy as np
import pytest
import torch

from dict_tts_amd import abi, hparams as hp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(abi.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return abi.load_library()


def test_header_symbols_exported(lib):
    text = open(os.path.join(ROOT, "include", "dicttts_hip.h")).read()
    declared = set(re.findall(r"\b(dtts_[a-z0-9_]+)\s*\(", text))
    assert declared, "no declarations found"
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} declared in include/dicttts_hip.h but not exported"
    assert declared == set(abi.EXPORTS)


def test_library_exports_exactly_the_header(lib):
    """the release library is sealed (-fvisibility=hidden + a linker version script): `nm -D` shows the dtts_* entry points
    of include/dicttts_hip.h and NOTHING else — no kernel host stubs, no C++ template instantiations, no internal helpers."""
    import subprocess
    text = open(os.path.join(ROOT, "include", "dicttts_hip.h")).read()
    declared = set(re.findall(r"\b(dtts_[a-z0-9_]+)\s*\(", text))
    out = subprocess.run(["nm", "-D", "--defined-only", abi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert exported == declared, (sorted(exported - declared)[:10], sorted(declared - exported)[:10])
    assert len(re.findall(r"^DTTS_API ", text, flags=re.M)) == len(declared)


def test_release_library_refuses_untested_tune_bits(lib):
    """a tune_flags bit without a parity / bit-identity test exists only in -DDTTS_ABLATE builds; the release library refuses
    it loudly (argument validation: no device needed) instead of ignoring it.  The tested bits (8, 9, 12, 13, 14) pass this check."""
    cfg = abi.default_config()
    for bit in (0, 1, 2, 3, 4, 5, 6, 7, 10, 11, 16, 17):
        cfg.tune_flags = 1 << bit
        h = abi.C.c_void_p()
        assert lib.dtts_create(abi.C.byref(cfg), abi.C.byref(h)) == -22, bit
        assert b"tune_flags" in lib.dtts_last_error(None)
    text = open(os.path.join(ROOT, "dict_tts_amd", "csrc", "tune_env.h")).read()
    assert "TUNE_RELEASE_MASK = (1 << 8) | (1 << 9) | (1 << 12) | (1 << 13) | (1 << 14)" in text


def test_default_config_matches_reference_hparams(lib):
    cfg = abi.default_config()
    assert (cfg.hidden_size, cfg.num_heads, cfg.enc_ffn_kernel_size, cfg.gloss_dim) == (192, 2, 5, 768)
    assert (cfg.latent_size, cfg.prior_glow_hidden, cfg.fvae_dec_n_layers, cfg.frames_multiple) == (16, 64, 4, 4)
    assert list(cfg.upsample_rates)[:4] == [8, 8, 2, 2] and list(cfg.upsample_kernel_sizes)[:4] == [16, 16, 4, 4]
    assert [list(r) for r in cfg.resblock_dilation_sizes][:3] == [[1, 3, 5]] * 3
    # struct layout: ctypes mirror and the C struct agree on the size
    assert abi.C.sizeof(abi.DttsConfig) == lib.dtts_config_sizeof() == 4 * (25 + 8 + 8 + 1 + 4 + 12 + 1 + 4 + 5)
    assert cfg.vocoder_precision == abi.VOC_F16 and cfg.vocoder_unfused == 0   # the waveform-exact mode is the default
    assert (cfg.fft_layers, cfg.fft_kernel_size, cfg.fft_use_pos_embed, cfg.fft_use_last_norm) == (4, 9, 1, 1)   # base.yaml:68,72


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful without a GPU")
def test_no_cpu_fallback(lib):
    with pytest.raises(abi.DttsError, match="no HIP device|no CPU fallback"):
        abi.Context()
    from dict_tts_amd import vocoder
    with pytest.raises(abi.DttsError, match="no CPU fallback"):
        vocoder.HifiGAN(state_dict={}, config={})


def test_yaml_chain_and_overrides(tmp_path):
    (tmp_path / "base.yaml").write_text("hidden_size: 256\nnum_heads: 2\nnested: {a: 1, b: 2}\nlst: [1, 2]\nflag: false\n")
    (tmp_path / "sub").mkdir()
    (tmp_path / "sub" / "cfg.yaml").write_text("base_config: ../base.yaml\nhidden_size: 192\nnested: {b: 3}\n")
    got = hp.load_config_chain(str(tmp_path / "sub" / "cfg.yaml"))
    assert got["hidden_size"] == 192 and got["num_heads"] == 2 and got["nested"] == {"a": 1, "b": 3}
    hp.apply_overrides(got, "hidden_size=128,flag=True,lst=[3 4 5],nested.a=7")
    assert got["hidden_size"] == 128 and got["flag"] is True and got["lst"] == [3, 4, 5] and got["nested"]["a"] == 7


def test_fill_abi_config_rejects_unsupported(lib):
    cfg = abi.default_config()
    with pytest.raises(NotImplementedError):
        hp.fill_abi_config(cfg, {"use_post_glow": True})
    with pytest.raises(NotImplementedError):
        hp.fill_abi_config(cfg, None, {**hp.HIFIGAN_DEFAULTS, "resblock": "2"})
    hp.fill_abi_config(cfg, {"hidden_size": 192}, hp.HIFIGAN_DEFAULTS, vocoder_precision=abi.VOC_BF16X3)
    assert cfg.vocoder_precision == abi.VOC_BF16X3


def test_checkpoint_layouts(tmp_path):
    """G7: the two checkpoint nestings the loaders must read (utils/trainer.py:436-449, vocoders/hifigan.py:18-24)"""
    from dict_tts_amd import model, vocoder
    w = {"conv_pre.bias": torch.zeros(4)}
    for step in (1000, 20000, 3000):
        torch.save({"state_dict": {"model": {"x.weight": torch.full((2,), float(step))}, "mel_disc": {}},
                    "global_step": step, "epoch": 1, "optimizer_states": []}, tmp_path / f"model_ckpt_steps_{step}.ckpt")
    sd, path = model.load_checkpoint_state(str(tmp_path))
    assert path.endswith("model_ckpt_steps_20000.ckpt") and float(sd["x.weight"][0]) == 20000.0
    vdir = tmp_path / "voc"
    vdir.mkdir()
    (vdir / "config.yaml").write_text("resblock: '1'\nupsample_rates: [8, 8, 2, 2]\n")
    torch.save({"state_dict": {"model_gen": w, "model_disc": {}}}, vdir / "model_ckpt_steps_5.ckpt")
    torch.save({"state_dict": {"model_gen": {"conv_pre.bias": torch.ones(4)}}}, vdir / "model_ckpt_steps_12.ckpt")
    cfg, st = vocoder.find_vocoder_checkpoint(str(vdir))
    assert cfg["upsample_rates"] == [8, 8, 2, 2] and float(st["conv_pre.bias"][0]) == 1.0
    jdir = tmp_path / "vocj"
    jdir.mkdir()
    (jdir / "config.json").write_text('{"resblock": "1"}')
    torch.save({"generator": w}, jdir / "generator_v1")
    cfg, st = vocoder.find_vocoder_checkpoint(str(jdir))
    assert cfg["resblock"] == "1" and "conv_pre.bias" in st


def test_vocoder_registry():
    from dict_tts_amd import vocoder
    assert vocoder.get_vocoder_cls({"vocoder": "hifigan"}) is vocoder.HifiGAN
    assert vocoder.get_vocoder_cls({"vocoder": "dict_tts_amd.vocoder.HifiGAN"}) is vocoder.HifiGAN


def test_synth_batch_layout():
    """make_batch reproduces DictTTSDataset.collater's padding rules (dataset_utils.py:264-302)"""
    from dict_tts_amd import synth
    st = synth.biaobei_struct()
    b = synth.make_batch(st["sentences"][:3], 1234)
    B, Tw = b["word_tokens"].shape
    assert b["keys"].shape == (B, Tw, b["key_map"].shape[2], 768) and b["keys"].dtype == np.float32
    assert (b["key_map"][:, 0] == 1).all() and (b["key_map"][:, -1] == 1).all()
    assert (b["keys"][:, 0] == 0).all() and (b["keys"][:, -1] == 0).all()
    assert (b["pinyin_map"][:, 0] == 1).all() and (b["pinyin"][:, 0] == 0).all()
    lens = (b["word_tokens"] > 0).sum(1)
    short = int(np.argmin(lens))
    if lens[short] < Tw:  # the true EOS row of a shorter sentence is all-zero
        assert (b["key_map"][short, lens[short] - 1] == 0).all()
    assert b["pinyin"].max() < 185 and b["pron_modified"].max() <= 6
