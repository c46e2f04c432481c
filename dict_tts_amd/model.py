"""Drop-in for the reference acoustic model ``modules.dict_tts.model.PortaSpeech_dict`` at inference
(modules/dict_tts/model.py:14-121): same ``forward`` signature and return keys, ``load_state_dict`` with the
reference's key names, every tensor op executed by libdicttts_hip.so.

What the reference ignores at inference is accepted and ignored here too: ``ph_tokens`` (txt_tokens[1]),
``key_value_map``, ``ph2word``, ``word_len``, ``mel2ph``, ``tgt_mels``, ``spk_embed`` (num_spk = 1).
One extra keyword, ``z_p`` ([B, latent, T_mel/4]): the prior sample, which the reference draws from the CPU
global RNG (modules/dict_tts/fvae_semantics.py:110-111); when omitted it is drawn the same way here.
"""
import numpy as np
import torch

from . import abi
from . import hparams as hparams_mod
from .hparams import fill_abi_config

# state-dict groups that exist in a Dict-TTS checkpoint but are never used by PortaSpeech_dict at inference
# (SURVEY.md §8a "Parameter inventory"): accepted by load_state_dict, not uploaded
UNUSED_PREFIXES = ("fvae.encoder.", "attn.", "enc_pos_proj.", "dec_query_proj.", "dec_res_proj.",
                   "dict_encoder.S2PA_module.emb.", "spk_embed_proj.", "post_flow.", "sin_pos.")


def load_checkpoint_state(work_dir, child="model"):
    """newest ``model_ckpt_steps_*.ckpt`` under work_dir -> state_dict[child]
    (utils/ckpt_utils.py:8-25, utils/trainer.py:348-376, 436-449)"""
    import glob
    import re
    paths = sorted(glob.glob(f"{work_dir}/model_ckpt_steps_*.ckpt"),
                   key=lambda x: -int(re.findall(r".*steps\_(\d+)\.ckpt", x)[0]))
    if not paths:
        raise FileNotFoundError(f"no model_ckpt_steps_*.ckpt under {work_dir}")
    ckpt = torch.load(paths[0], map_location="cpu", weights_only=False)
    return ckpt["state_dict"][child], paths[0]


class PortaSpeech_dict(torch.nn.Module):
    def __init__(self, dictionary=None, out_dims=None, hparams=None, ctx=None):
        """dictionary: anything with __len__ (the phoneme TokenTextEncoder of the reference); only its length
        matters (rows of the unused phoneme embedding)."""
        super().__init__()
        if not torch.cuda.is_available():
            raise abi.DttsError("dict_tts_amd.model.PortaSpeech_dict needs a ROCm GPU: the HIP path has no CPU fallback")
        if hparams is None:   # the global hparams, looked up at call time (the INTEGRATION.md hook may rebind them late)
            if not hparams_mod.hparams:
                raise RuntimeError("PortaSpeech_dict(hparams=None) needs the global hparams: call dict_tts_amd.hparams.set_hparams(...) "
                                   "or alias the reference's (INTEGRATION.md), or pass hparams={} for the Biaobei defaults")
            hparams = hparams_mod.hparams
        hp = dict(hparams)
        self.hparams = hp
        n_phone = len(dictionary) if dictionary is not None else None
        if ctx is None:
            ctx = abi.Context(fill_abi_config(abi.default_config(), hp, None, n_phone=n_phone))
        self.ctx = ctx
        self.cfg = ctx.cfg
        self._state = {}
        self._ready = False
        self.device = torch.device("cuda", torch.cuda.current_device())

    # -- checkpoint compatibility --------------------------------------------------------------------------
    def load_state_dict(self, state_dict, strict=True):
        unexpected = []
        for k in state_dict:
            if k.endswith((".weight", ".bias", ".weight_g", ".weight_v", ".gamma", ".beta", ".in_proj_weight")):
                continue
            unexpected.append(k)
        if strict and unexpected:
            raise RuntimeError(f"Unexpected key(s) in state_dict: {unexpected[:5]}")
        used = {k: v for k, v in state_dict.items() if not k.startswith(UNUSED_PREFIXES)}
        self._state = dict(state_dict)
        self.ctx.load_state_dict("model", used)
        try:
            self.ctx.finalize(abi.PART_ACOUSTIC)   # names the first missing tensor (strict)
        except abi.DttsError as e:
            raise RuntimeError(f"Error(s) in loading state_dict for PortaSpeech_dict: {e}") from e
        self._ready = True
        return torch.nn.modules.module._IncompatibleKeys([], [])

    def state_dict(self, *args, **kwargs):
        return dict(self._state)

    # -- inference -----------------------------------------------------------------------------------------
    def forward(self, txt_tokens, pron_modified, key_value_map, ph2word, word_len, dict_msg, mel2word=None, mel2ph=None,
                spk_embed=None, infer=False, tgt_mels=None, forward_post_glow=True, two_stage=True, z_p=None):
        if not infer:
            raise NotImplementedError("the MI355X path implements inference only (infer=True)")
        if not self._ready:
            raise RuntimeError("load_state_dict() must be called first")
        dev = self.device
        i64 = lambda t: None if t is None else t.to(device=dev, dtype=torch.int64).contiguous()
        f32 = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
        word_tokens = i64(txt_tokens[0])
        keys, values, key_map = f32(dict_msg[0]), f32(dict_msg[1]), f32(dict_msg[2])
        pinyin, pinyin_map = i64(dict_msg[3]), i64(dict_msg[4])
        pron_modified = i64(pron_modified)
        mel2word = i64(mel2word)
        B, T_w = word_tokens.shape
        L_k, P = keys.shape[2], pinyin.shape[2]
        assert keys.shape == (B, T_w, L_k, self.cfg.gloss_dim) and values.shape == keys.shape
        assert key_map.shape == (B, T_w, L_k) and pinyin.shape == pinyin_map.shape == (B, T_w, P)
        stream = torch.cuda.current_stream().cuda_stream
        ptr = lambda t: None if t is None else t.data_ptr()
        return self._finish(self.ctx.text2mel_encode(ptr(word_tokens), ptr(keys), ptr(values), ptr(key_map), ptr(pinyin),
                                         ptr(pinyin_map), ptr(pron_modified),
                                         (ptr(mel2word), mel2word.shape[1]) if mel2word is not None else None, B, T_w,
                                         L_k, P, stream), z_p, B, T_w, L_k, P)

    def upload_dict_table(self, table):
        """make the dictionary resident in HBM (dict_tts_amd/synth.py:dict_table layout = the reference's dict_embed items)"""
        self.ctx.dict_table_upload(table["tok_off"], table["keys"], table.get("values"), table["key_map"], table["pin_off"],
                                   table["pinyin"], table["pinyin_map"])

    def forward_ids(self, word_tokens, entry_ids, pron_modified, L_k, P, mel2word=None, z_p=None):
        """forward(infer=True) with the dictionary tensors replaced by ids into the resident table"""
        dev = self.device
        word_tokens = word_tokens.to(device=dev, dtype=torch.int64).contiguous()
        entry_ids = entry_ids.to(device=dev, dtype=torch.int32).contiguous()
        pron_modified = None if pron_modified is None else pron_modified.to(device=dev, dtype=torch.int64).contiguous()
        mel2word = None if mel2word is None else mel2word.to(device=dev, dtype=torch.int64).contiguous()
        B, T_w = word_tokens.shape
        stream = torch.cuda.current_stream().cuda_stream
        ptr = lambda t: None if t is None else t.data_ptr()
        T_mel = self.ctx.text2mel_encode_ids(ptr(word_tokens), ptr(entry_ids), ptr(pron_modified),
                                             (ptr(mel2word), mel2word.shape[1]) if mel2word is not None else None, B, T_w,
                                             int(L_k), int(P), stream)
        return self._finish(T_mel, z_p, B, T_w, int(L_k), int(P))

    def _finish(self, T_mel, z_p, B, T_w, L_k, P):
        dev = self.device
        f32 = lambda t: t.to(device=dev, dtype=torch.float32).contiguous()
        stream = torch.cuda.current_stream().cuda_stream
        Z = self.cfg.latent_size
        if z_p is None:
            z_p = torch.distributions.Normal(0, 1).sample([B, Z, T_mel // 4])  # fvae_semantics.py:110
        z_p = f32(z_p)
        assert tuple(z_p.shape) == (B, Z, T_mel // 4), (tuple(z_p.shape), (B, Z, T_mel // 4))
        n_mel, H = self.cfg.audio_num_mel_bins, self.cfg.hidden_size
        mel = torch.empty(B, T_mel, n_mel, dtype=torch.float32, device=dev)
        self.ctx.text2mel_decode(z_p.data_ptr(), mel.data_ptr(), stream)
        ret = {}
        out = lambda shape, dt=torch.float32: torch.empty(*shape, dtype=dt, device=dev)
        ret["pron_attn"] = out((B, T_w, P))
        ret["dur"] = out((B, T_w))
        ret["dict_attn"] = out((B, 1, L_k, T_w))
        ret["word_encoder_out"] = out((B, T_w, H))
        ret["x_mask"] = out((B, T_mel, 1))
        ret["mel2word"] = out((B, T_mel), torch.int64)
        ret["mel_lens"] = out((B,), torch.int32)
        for key, what in (("pron_attn", abi.OUT_PRON_ATTN), ("dur", abi.OUT_DUR), ("dict_attn", abi.OUT_DICT_ATTN),
                          ("word_encoder_out", abi.OUT_WORD_ENCODER_OUT), ("x_mask", abi.OUT_X_MASK),
                          ("mel2word", abi.OUT_MEL2WORD), ("mel_lens", abi.OUT_MEL_LENS)):
            self.ctx.fetch(what, ret[key].data_ptr(), stream)
        ret["mel_out"] = ret["mel_out_fvae"] = mel
        ret["rel"] = ret["dp_attn"] = None
        ret["z_p_in"] = z_p
        return ret


def decode_pinyin_ids(pron_attn, pinyin):
    """after_infer's pinyin decode for ONE utterance (tasks/tts/dict_tts.py:294-304): pron_attn [T_w,P],
    pinyin [T_w,P] -> list of pinyin-token ids, two per inner word"""
    pron_attn = torch.as_tensor(np.asarray(pron_attn.detach().cpu() if hasattr(pron_attn, "detach") else pron_attn))
    pinyin = torch.as_tensor(np.asarray(pinyin.detach().cpu() if hasattr(pinyin, "detach") else pinyin))
    _, max_idx = pron_attn.max(dim=-1)
    ids = []
    for i in range(1, pinyin.shape[0] - 1):
        ids += pinyin[i][max_idx[i]:max_idx[i] + 2].tolist()
    return ids
