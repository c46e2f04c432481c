"""Configuration for the MI355X path: reads the SAME YAML chains / checkpoint config the reference reads.

Mirrors the subset of ``utils/hparams.py:25-126`` the inference path needs: ``base_config`` inheritance (a
string or a list; entries starting with '.' are relative to the including file, others to the working
directory), the saved ``checkpoints/<exp>/config.yaml`` layered on top, and ``--hparams=k=v,...`` overrides
with the reference's typing rule.  ``hparams`` is a module-level dict, as in the reference, so the vocoder
plugin can be constructed with no arguments (vocoders/base_vocoder.py:15-23).
"""
import os

import yaml

hparams = {}


def _override(old, new):
    for k, v in new.items():
        if isinstance(v, dict) and isinstance(old.get(k), dict):
            _override(old[k], v)
        else:
            old[k] = v


def load_config_chain(config_fn, _seen=None):
    seen = set() if _seen is None else _seen
    if not os.path.exists(config_fn):
        return {}
    with open(config_fn) as f:
        cur = yaml.safe_load(f) or {}
    seen.add(config_fn)
    out = {}
    bases = cur.get("base_config", [])
    if not isinstance(bases, list):
        bases = [bases]
    for c in bases:
        if c.startswith("."):
            c = os.path.normpath(os.path.join(os.path.dirname(config_fn), c))
        if c not in seen:
            _override(out, load_config_chain(c, seen))
    _override(out, cur)
    return out


def apply_overrides(hp, hparams_str):
    """--hparams="a=1,b.c=2,d=[1 1 1]" (utils/hparams.py:85-99)"""
    if not hparams_str:
        return hp
    for item in hparams_str.split(","):
        k, v = item.split("=")
        v = v.strip("'\" ")
        node = hp
        for part in k.split(".")[:-1]:
            node = node[part]
        k = k.split(".")[-1]
        old = node.get(k)
        if v in ("True", "False") or isinstance(old, (bool, list, dict)):
            if isinstance(old, list):
                v = v.replace(" ", ",")
            node[k] = eval(v, {"__builtins__": {}}, {})  # literals only
        elif old is None:
            node[k] = yaml.safe_load(v)
        else:
            node[k] = type(old)(v)
    return hp


def set_hparams(config="", exp_name="", hparams_str="", global_hparams=True, work_root="checkpoints"):
    hp = {}
    if config:
        hp.update(load_config_chain(config))
    work_dir = os.path.join(work_root, exp_name) if exp_name else ""
    if work_dir and os.path.exists(os.path.join(work_dir, "config.yaml")):
        with open(os.path.join(work_dir, "config.yaml")) as f:
            hp.update(yaml.safe_load(f) or {})
    hp["work_dir"] = work_dir
    apply_overrides(hp, hparams_str)
    hp["exp_name"] = exp_name
    hp["infer"] = True
    if global_hparams:
        hparams.clear()
        hparams.update(hp)
    return hp


# resolved defaults of egs/datasets/audio/biaobei/dict_tts.yaml (+ use_word_input/word_size/use_dict from the README
# command line), SURVEY.md §5 "Config / flags"; used when no YAML is at hand (synthetic runs)
BIAOBEI_DEFAULTS = {
    "hidden_size": 192, "num_heads": 2, "enc_ffn_kernel_size": 5, "word_size": 8000, "value_embedding_size": 185,
    "audio_num_mel_bins": 80, "latent_size": 16, "fvae_enc_dec_hidden": 192, "fvae_kernel_size": 5,
    "fvae_dec_n_layers": 4, "fvae_enc_n_layers": 8, "prior_glow_hidden": 64, "glow_kernel_size": 3,
    "prior_glow_n_blocks": 4, "dur_predictor_layers": 3, "dur_predictor_kernel": 5, "frames_multiple": 4,
    "language": "zh", "use_post_glow": False, "use_prior_glow": True, "dur_scale": "log", "dur_level": "word",
    "audio_sample_rate": 22050, "hop_size": 256, "use_spk_embed": False, "use_spk_id": False, "num_spk": 1,
    "vocoder": "dict_tts_amd.vocoder.HifiGAN", "vocoder_ckpt": "", "use_word_input": True, "use_dict": True,
}

HIFIGAN_DEFAULTS = {
    "resblock": "1", "upsample_rates": [8, 8, 2, 2], "upsample_kernel_sizes": [16, 16, 4, 4],
    "upsample_initial_channel": 512, "resblock_kernel_sizes": [3, 7, 11],
    "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
}


def fill_abi_config(cfg, hp=None, voc=None, n_phone=None, vocoder_precision=None):
    """copy reference hparams into a DttsConfig (abi.DttsConfig); unsupported settings fail loudly"""
    hp = {**BIAOBEI_DEFAULTS, **(hp or {})}
    if hp.get("use_post_glow"):
        raise NotImplementedError("use_post_glow=True is outside the Dict-TTS path (egs/egs_bases/tts/dict_tts.yaml:4)")
    if hp.get("use_spk_embed") or hp.get("use_spk_id"):
        raise NotImplementedError("speaker embeddings are not part of the Biaobei Dict-TTS path (num_spk=1)")
    if not hp.get("use_prior_glow", True) or hp.get("dur_scale", "log") != "log":
        raise NotImplementedError("only use_prior_glow=True / dur_scale=log are implemented")
    for k in ("hidden_size", "num_heads", "enc_ffn_kernel_size", "word_size", "value_embedding_size",
              "audio_num_mel_bins", "latent_size", "fvae_enc_dec_hidden", "fvae_kernel_size", "fvae_dec_n_layers",
              "fvae_enc_n_layers", "prior_glow_hidden", "glow_kernel_size", "prior_glow_n_blocks",
              "dur_predictor_layers", "dur_predictor_kernel", "frames_multiple"):
        setattr(cfg, k, int(hp[k]))
    cfg.language_zh = 1 if hp.get("language", "zh") == "zh" else 0
    if n_phone is not None:
        cfg.n_phone = int(n_phone)
    if voc is not None:
        if str(voc.get("resblock", "1")) != "1":
            raise NotImplementedError("only ResBlock1 generators are implemented (hifigan.yaml: resblock '1')")
        ur, uk = list(voc["upsample_rates"]), list(voc["upsample_kernel_sizes"])
        rk, rd = list(voc["resblock_kernel_sizes"]), [list(d) for d in voc["resblock_dilation_sizes"]]
        cfg.upsample_initial_channel = int(voc["upsample_initial_channel"])
        cfg.n_upsamples = len(ur)
        for i in range(len(ur)):
            cfg.upsample_rates[i] = int(ur[i])
            cfg.upsample_kernel_sizes[i] = int(uk[i])
        cfg.n_resblock_kernels = len(rk)
        for i in range(len(rk)):
            cfg.resblock_kernel_sizes[i] = int(rk[i])
            if len(rd[i]) != 3:
                raise NotImplementedError("ResBlock1 expects 3 dilations per kernel size")
            for j in range(3):
                cfg.resblock_dilation_sizes[i][j] = int(rd[i][j])
    if vocoder_precision is not None:
        cfg.vocoder_precision = int(vocoder_precision)
    # A/B switches of tuning experiments (include/dicttts_hip.h: dtts_config.tune_flags; 0 = the measured defaults).  An explicit
    # key of the hparams / vocoder config, never the process environment.
    cfg.tune_flags = int((hp or {}).get("dtts_tune_flags", 0)) | int((voc or {}).get("dtts_tune_flags", 0))
    cfg.debug_redzone = 1 if ((hp or {}).get("dtts_debug_redzone") or (voc or {}).get("dtts_debug_redzone")) else 0
    return cfg
