"""ctypes binding of libdicttts_hip.so (include/dicttts_hip.h).  No torch types cross the boundary: device
pointers are passed as integers (``tensor.data_ptr()``), the stream as ``torch.cuda.current_stream().cuda_stream``.

There is NO CPU fallback here: if the shared library is missing, or no GPU is visible, construction fails loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdicttts_hip.so")

DTTS_F32, DTTS_I64 = 0, 1
VOC_BF16, VOC_BF16X3, VOC_F16 = 0, 1, 2
VOC_PRECISIONS = {"f16": VOC_F16, "bf16": VOC_BF16, "bf16x3": VOC_BF16X3}
PART_ACOUSTIC, PART_VOCODER, PART_FFT = 1, 2, 4
OUT_PRON_ATTN, OUT_DUR, OUT_MEL2WORD, OUT_DICT_ATTN, OUT_WORD_ENCODER_OUT, OUT_X_MASK, OUT_CONTEXT, OUT_MEL_LENS = range(1, 9)
TIMER_VOC_CONV, TIMER_S2PA = 1, 2
TIMER_STAGE_ENCODER, TIMER_STAGE_DICT_ENCODER, TIMER_STAGE_FVAE, TIMER_STAGE_HIFIGAN = 3, 4, 5, 6   # the reference's profile_infer names

EXPORTS = ["dtts_default_config", "dtts_config_sizeof", "dtts_create", "dtts_destroy", "dtts_last_error", "dtts_load_weight",
           "dtts_finalize_weights", "dtts_dict_table_upload", "dtts_text2mel_encode", "dtts_text2mel_encode_ids", "dtts_text2mel_decode", "dtts_text2mel_fetch",
           "dtts_load_weights", "dtts_text2mel_plan", "dtts_text2mel_forward", "dtts_text2mel_forward_ids",
           "dtts_length_regulate", "dtts_hifigan_forward", "dtts_hifigan_hop", "dtts_wav_to_int16", "dtts_fft_blocks_forward",
           "dtts_timer_enable", "dtts_timer_read", "dtts_timer_reset", "dtts_set_noise_seed", "dtts_vocoder_range_guard", "dtts_vocoder_clamped", "dtts_vocoder_nonfinite", "dtts_vocoder_fp16_bound",
           "dtts_debug_check", "dtts_debug_poke"]


class DttsConfig(C.Structure):
    """struct dtts_config (include/dicttts_hip.h); field <- reference hparams key"""
    _fields_ = [(n, C.c_int32) for n in (
        "hidden_size", "num_heads", "enc_ffn_kernel_size", "enc_layers", "gloss_dim", "word_size",
        "value_embedding_size", "n_phone", "audio_num_mel_bins", "latent_size", "fvae_enc_dec_hidden",
        "fvae_kernel_size", "fvae_dec_n_layers", "fvae_enc_n_layers", "prior_glow_hidden", "glow_kernel_size",
        "prior_glow_n_blocks", "prior_glow_n_layers", "dur_predictor_layers", "dur_predictor_kernel", "dur_chans",
        "frames_multiple", "language_zh", "upsample_initial_channel", "n_upsamples")] + [
        ("upsample_rates", C.c_int32 * 8), ("upsample_kernel_sizes", C.c_int32 * 8), ("n_resblock_kernels", C.c_int32),
        ("resblock_kernel_sizes", C.c_int32 * 4), ("resblock_dilation_sizes", (C.c_int32 * 3) * 4),
        ("vocoder_precision", C.c_int32), ("fft_layers", C.c_int32), ("fft_kernel_size", C.c_int32),
        ("fft_use_pos_embed", C.c_int32), ("fft_use_last_norm", C.c_int32), ("vocoder_unfused", C.c_int32), ("decoder_fp32", C.c_int32),
        ("vocoder_range_guard", C.c_int32), ("debug_redzone", C.c_int32), ("tune_flags", C.c_int32)]


class DttsError(RuntimeError):
    pass


_lib = None


def load_library(path=None):
    """dlopen the in-tree library; raise (never fall back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or LIB_PATH
    try:   # PyTorch-ROCm bundles its own HIP runtime: load it FIRST so that this library binds to the same one (a process that loads
        import torch  # noqa: F401  # the system libamdhip64 first and torch's afterwards ends up with two runtimes and no visible device)
    except ImportError:
        pass
    if not os.path.exists(path):
        raise DttsError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        f"or `make -C dict_tts_amd/csrc` (there is no CPU fallback for the HIP path)")
    lib = C.CDLL(path)
    vp, i32, i64p = C.c_void_p, C.c_int, C.POINTER(C.c_int64)
    lib.dtts_default_config.argtypes = [C.POINTER(DttsConfig)]
    lib.dtts_default_config.restype = None
    lib.dtts_config_sizeof.argtypes = []
    lib.dtts_config_sizeof.restype = C.c_int
    lib.dtts_create.argtypes = [C.POINTER(DttsConfig), C.POINTER(vp)]
    lib.dtts_destroy.argtypes = [vp]
    lib.dtts_destroy.restype = None
    lib.dtts_last_error.argtypes = [vp]
    lib.dtts_last_error.restype = C.c_char_p
    lib.dtts_load_weight.argtypes = [vp, C.c_char_p, vp, i64p, i32, i32]
    lib.dtts_finalize_weights.argtypes = [vp, i32]
    lib.dtts_text2mel_encode.argtypes = [vp] + [vp] * 8 + [i32] * 5 + [C.POINTER(C.c_int32), vp]
    lib.dtts_text2mel_decode.argtypes = [vp, vp, vp, vp]
    lib.dtts_text2mel_encode_ids.argtypes = [vp, vp, vp, vp, vp] + [i32] * 5 + [C.POINTER(C.c_int32), vp]
    lib.dtts_dict_table_upload.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp]
    lib.dtts_text2mel_fetch.argtypes = [vp, i32, vp, vp]
    lib.dtts_hifigan_forward.argtypes = [vp, vp, vp, i32, i32, vp, vp]
    lib.dtts_length_regulate.argtypes = [vp, vp, vp, i32, i32, vp, i32, C.POINTER(C.c_int32), vp]
    lib.dtts_hifigan_hop.argtypes = [vp]
    lib.dtts_wav_to_int16.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp]
    lib.dtts_text2mel_forward.argtypes = [vp] + [vp] * 8 + [i32, vp, i32, i32, i32, i32, i32, vp, i32, C.POINTER(C.c_int64), vp, vp, vp]
    lib.dtts_text2mel_forward_ids.argtypes = [vp] + [vp] * 4 + [i32, vp, i32, i32, i32, i32, i32, vp, i32, C.POINTER(C.c_int64), vp, vp, vp]
    lib.dtts_fft_blocks_forward.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, vp]
    lib.dtts_timer_enable.argtypes = [vp, i32]
    lib.dtts_timer_read.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.dtts_timer_reset.argtypes = [vp]
    lib.dtts_set_noise_seed.argtypes = [vp, C.c_uint64]
    lib.dtts_vocoder_range_guard.argtypes = [vp, i32]
    lib.dtts_vocoder_clamped.argtypes = [vp, C.POINTER(C.c_int64), i32, vp]
    lib.dtts_vocoder_nonfinite.argtypes = [vp, C.POINTER(C.c_int64)]
    lib.dtts_vocoder_fp16_bound.argtypes = [vp, C.c_float, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.dtts_debug_check.argtypes = [vp, C.POINTER(C.c_int64), vp]
    lib.dtts_debug_poke.argtypes = [vp, vp]
    _lib = lib
    return lib


def default_config():
    cfg = DttsConfig()
    lib = load_library()
    if lib.dtts_config_sizeof() != C.sizeof(DttsConfig):
        raise DttsError(f"dtts_config layout mismatch: library {lib.dtts_config_sizeof()} B, abi.DttsConfig {C.sizeof(DttsConfig)} B")
    lib.dtts_default_config(C.byref(cfg))
    return cfg


class Context:
    """One dtts_handle (single owner, one per GPU / process)."""

    def __init__(self, cfg=None):
        self.lib = load_library()
        self.cfg = cfg or default_config()
        self.h = C.c_void_p()
        rc = self.lib.dtts_create(C.byref(self.cfg), C.byref(self.h))
        if rc != 0:
            raise DttsError(f"dtts_create failed ({rc}): {self.lib.dtts_last_error(None).decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.lib.dtts_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_error(self):
        return self.lib.dtts_last_error(self.h).decode()

    def _chk(self, rc, what):
        if rc != 0:
            raise DttsError(f"{what} failed ({rc}): {self.lib.dtts_last_error(self.h).decode()}")

    def load_state_dict(self, prefix, state):
        """state: mapping name -> numpy array or torch tensor (fp32); names are the reference's keys"""
        for k, v in state.items():
            a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
            if a.dtype != np.float32:
                if not np.issubdtype(a.dtype, np.floating):
                    continue  # e.g. num_batches_tracked-style integer buffers: not part of this path
                a = a.astype(np.float32)
            a = np.ascontiguousarray(a)
            shape = (C.c_int64 * max(a.ndim, 1))(*a.shape)
            self._chk(self.lib.dtts_load_weight(self.h, f"{prefix}.{k}".encode(), a.ctypes.data_as(C.c_void_p), shape,
                                                a.ndim, DTTS_F32), f"dtts_load_weight({k})")

    def finalize(self, parts):
        self._chk(self.lib.dtts_finalize_weights(self.h, parts), "dtts_finalize_weights")

    def hop(self):
        return self.lib.dtts_hifigan_hop(self.h)

    def text2mel_encode(self, word_tokens, keys, values, key_map, pinyin, pinyin_map, pron_modified, mel2word, B, T_w,
                        L_k, P, stream):
        """all tensor arguments are device pointers (ints, 0/None = NULL); returns T_mel"""
        t_mel = C.c_int32(0)
        m2w_ptr, T_m2w = (mel2word if mel2word else (None, 0))
        self._chk(self.lib.dtts_text2mel_encode(self.h, word_tokens, keys, values, key_map, pinyin, pinyin_map,
                                                pron_modified or None, m2w_ptr, T_m2w, B, T_w, L_k, P, C.byref(t_mel),
                                                stream), "dtts_text2mel_encode")
        return t_mel.value

    def dict_table_upload(self, tok_off, keys, values, key_map, pin_off, pinyin, pinyin_map):
        """numpy arrays (host): tok_off/pin_off int32 [n+1], keys/values f32 [sum_L,768] (values may be None = keys),
        key_map f32 [sum_L], pinyin / pinyin_map int64 [sum_P]"""
        c = lambda a, dt: np.ascontiguousarray(a, dtype=dt)
        tok_off, pin_off = c(tok_off, np.int32), c(pin_off, np.int32)
        keys, key_map = c(keys, np.float32), c(key_map, np.float32)
        values = None if values is None else c(values, np.float32)
        pinyin, pinyin_map = c(pinyin, np.int64), c(pinyin_map, np.int64)
        p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        self._chk(self.lib.dtts_dict_table_upload(self.h, len(tok_off) - 1, p(tok_off), p(keys), p(values), p(key_map),
                                                  p(pin_off), p(pinyin), p(pinyin_map)), "dtts_dict_table_upload")

    def text2mel_encode_ids(self, word_tokens, entry_ids, pron_modified, mel2word, B, T_w, L_k, P, stream):
        t_mel = C.c_int32(0)
        m2w_ptr, T_m2w = (mel2word if mel2word else (None, 0))
        self._chk(self.lib.dtts_text2mel_encode_ids(self.h, word_tokens, entry_ids, pron_modified or None, m2w_ptr, T_m2w, B,
                                                    T_w, L_k, P, C.byref(t_mel), stream), "dtts_text2mel_encode_ids")
        return t_mel.value

    def text2mel_decode(self, z_p, mel_out, stream):
        self._chk(self.lib.dtts_text2mel_decode(self.h, z_p, mel_out, stream), "dtts_text2mel_decode")

    def fetch(self, what, dst, stream):
        self._chk(self.lib.dtts_text2mel_fetch(self.h, what, dst, stream), "dtts_text2mel_fetch")

    def length_regulate(self, dur, ilens, B, T_w, mel2word, cap, stream):
        t_max = C.c_int32(0)
        self._chk(self.lib.dtts_length_regulate(self.h, dur, ilens, B, T_w, mel2word, cap, C.byref(t_max), stream),
                  "dtts_length_regulate")
        return t_max.value

    def hifigan_forward(self, mel, lens, B, T, wav, stream):
        self._chk(self.lib.dtts_hifigan_forward(self.h, mel, lens or None, B, T, wav, stream), "dtts_hifigan_forward")

    def wav_to_int16(self, wav, lens, B, T, norm, out, stream):
        self._chk(self.lib.dtts_wav_to_int16(self.h, wav, lens or None, B, T, int(bool(norm)), out, stream), "dtts_wav_to_int16")

    def text2mel_forward(self, word_tokens, keys, values, key_map, pinyin, pinyin_map, pron_modified, mel2word, z_p, z_cap, B, T_w, L_k,
                         P, mel_out, mel_cap, pron_attn, dur, stream):
        """single call: mel2word = (ptr, T_m2w) or None; z_p = device ptr [B,latent,z_cap] or None; returns T_mel"""
        m2w, t_m2w = mel2word if mel2word else (None, 0)
        t_mel = C.c_int64(0)
        self._chk(self.lib.dtts_text2mel_forward(self.h, word_tokens, keys, values, key_map, pinyin, pinyin_map, pron_modified or None,
                                                 m2w, t_m2w, z_p or None, int(z_cap), B, T_w, L_k, P, mel_out, int(mel_cap),
                                                 C.byref(t_mel), pron_attn or None, dur or None, stream), "dtts_text2mel_forward")
        return t_mel.value

    def fft_blocks_forward(self, x, lens, pos_table, n_pos, B, T, y, stream):
        self._chk(self.lib.dtts_fft_blocks_forward(self.h, x, lens or None, pos_table or None, int(n_pos), B, T, y, stream),
                  "dtts_fft_blocks_forward")

    def timer_enable(self, which):
        self._chk(self.lib.dtts_timer_enable(self.h, which), "dtts_timer_enable")

    def timer_read(self, which):
        ms, n = C.c_double(0), C.c_int64(0)
        self._chk(self.lib.dtts_timer_read(self.h, which, C.byref(ms), C.byref(n)), "dtts_timer_read")
        return ms.value, n.value

    def timer_reset(self):
        self._chk(self.lib.dtts_timer_reset(self.h), "dtts_timer_reset")

    def set_noise_seed(self, seed):
        """seed of the device-side prior sample (z_p == NULL); contexts otherwise start from a time / pid / device dependent seed"""
        self._chk(self.lib.dtts_set_noise_seed(self.h, int(seed) & (2 ** 64 - 1)), "dtts_set_noise_seed")

    def vocoder_range_guard(self, enable):
        self._chk(self.lib.dtts_vocoder_range_guard(self.h, int(bool(enable))), "dtts_vocoder_range_guard")

    def debug_check(self, stream):
        """memory-safety mode (config.debug_redzone): red-zone bytes damaged so far (synchronises the stream); 0 = every kernel stayed
        inside its buffers.  The first damaged zone is named by last_error()."""
        n = C.c_int64(0)
        self._chk(self.lib.dtts_debug_check(self.h, C.byref(n), stream), "dtts_debug_check")
        return n.value

    def debug_poke(self, stream):
        """self-test of the memory-safety mode: damage one red-zone byte"""
        self._chk(self.lib.dtts_debug_poke(self.h, stream), "dtts_debug_poke")

    def vocoder_clamped(self, stream, reset=True):
        """activations the fp16 ResBlock operands could not represent since the last reset (synchronises the stream)"""
        n = C.c_int64(0)
        self._chk(self.lib.dtts_vocoder_clamped(self.h, C.byref(n), int(bool(reset)), stream), "dtts_vocoder_clamped")
        return n.value

    def vocoder_nonfinite(self):
        """cumulative count of non-finite pre-tanh samples the always-on conv_post detector saw (no synchronisation: valid for every
        forward whose stream the caller has synchronised with)"""
        n = C.c_int64(0)
        self._chk(self.lib.dtts_vocoder_nonfinite(self.h, C.byref(n)), "dtts_vocoder_nonfinite")
        return n.value

    def vocoder_fp16_bound(self, mel_abs_max=6.0):
        """(worst_case, rms_estimate) of the values the ResBlocks round to fp16 for |mel| <= mel_abs_max (dtts_vocoder_fp16_bound)"""
        wc, est = C.c_double(0), C.c_double(0)
        self._chk(self.lib.dtts_vocoder_fp16_bound(self.h, C.c_float(mel_abs_max), C.byref(wc), C.byref(est)), "dtts_vocoder_fp16_bound")
        return wc.value, est.value
