/*
 * libdicttts_hip.so — C ABI of the MI355X (gfx950) Dict-TTS inference hot path.
 *
 * The reference (Zain-Jiang/Dict-TTS) is pure Python/PyTorch and has no FFI; the two plugin points this
 * library sits behind are:
 *   - the acoustic model  PortaSpeech_dict.forward(..., infer=True)     modules/dict_tts/model.py:36-62
 *   - the vocoder         HifiGAN.spec2wav(mel[T,80]) -> wav[T*256]      vocoders/hifigan.py:54-62
 * The Python shims in dict_tts_amd/ (model.py, vocoder.py) keep those signatures and call the entry points
 * below through ctypes.  No torch types cross this boundary: plain pointers, sizes and a hipStream_t.
 *
 * Conventions
 *   - every function returns 0 on success or a negative DTTS_E_* code; the message is dtts_last_error().
 *   - "dev" pointers are DEVICE pointers owned by the caller (e.g. torch tensors' data_ptr()), contiguous,
 *     row-major in the reference's layouts.  "host" pointers are ordinary host memory.
 *   - a handle is single-owner and not thread-safe; all work is enqueued on the caller's stream.  The only
 *     host synchronisation on the path is the T_mel scalar read in dtts_text2mel_encode().
 */
#ifndef DICTTTS_HIP_H
#define DICTTTS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: exactly the entry points declared here (DTTS_API) are exported. */
#if defined(__GNUC__)
#define DTTS_API __attribute__((visibility("default")))
#else
#define DTTS_API
#endif

typedef struct dtts_ctx* dtts_handle;
typedef void* dtts_stream; /* hipStream_t */

#define DTTS_OK 0
#define DTTS_E_INVAL (-22)   /* bad argument / shape                       */
#define DTTS_E_NOMEM (-12)   /* device allocation failed                   */
#define DTTS_E_NOENT (-2)    /* a required weight tensor was never loaded  */
#define DTTS_E_STATE (-1)    /* call order violated (e.g. decode before encode) */
#define DTTS_E_HIP (-5)      /* a HIP runtime call failed                  */

#define DTTS_F32 0
#define DTTS_I64 1

/* Largest sense index (key_map / pinyin_map value) a word may carry: the S2PA kernel keeps 16 sense slots, index 0 = "no
 * sense".  zh-dict.json holds at most 6 pronunciations per character.  Larger indices are rejected (DTTS_E_INVAL) by dtts_dict_table_upload and, on the
 * tensor API, at the T_mel synchronisation of dtts_text2mel_encode; with teacher-forced mel2word (no synchronisation) they
 * are not checked and get weight 0. */
#define DTTS_MAX_SENSES 15

/* vocoder arithmetic (fp32 accumulation and an fp32 residual stream in every mode).
 *   DTTS_VOC_F16 (default): the fused kernels with fp16 MFMA operands in the ResBlocks and bf16 hi/lo split operands (three
 *     products) in the six serial convolutions (conv_pre, upsamplers, conv_post: 3 % of the FLOPs but 87 % of the 16-bit
 *     rounding noise, tools/precision_sim.py) — waveform RMS error vs the fp32 reference ~5e-5 (gate 1e-4);
 *     activations saturate at the fp16 maximum 65504.
 *   DTTS_VOC_BF16: the same kernels on bf16 operands throughout — RMS error ~1e-3 (fails the waveform gate), a few % faster.
 *   DTTS_VOC_BF16X3: every convolution on split operands on the generic kernel — fp32-class (~2e-6), several times slower. */
#define DTTS_VOC_BF16 0
#define DTTS_VOC_BF16X3 1
#define DTTS_VOC_F16 2

/* Model hyper-parameters.  Field <- reference hparams key (resolved values, SURVEY.md §5 "Config / flags"). */
typedef struct dtts_config {
    int32_t hidden_size;          /* hidden_size 192                         egs/egs_bases/tts/ps_flow.yaml:10 */
    int32_t num_heads;            /* num_heads 2                             egs/egs_bases/tts/base.yaml:70    */
    int32_t enc_ffn_kernel_size;  /* enc_ffn_kernel_size 5                   ps_flow.yaml:11                   */
    int32_t enc_layers;           /* 4, hard-coded at modules/dict_tts/layers/dict_encoder.py:104-128         */
    int32_t gloss_dim;            /* 768: key/value size of S2PAAttention    dict_encoder.py:18                */
    int32_t word_size;            /* word_size 8000                                                          */
    int32_t value_embedding_size; /* value_embedding_size 185                                                */
    int32_t n_phone;              /* rows of the unused phoneme embedding (len(phone_set)+3)                  */
    int32_t audio_num_mel_bins;   /* 80                                      base.yaml:50                      */
    int32_t latent_size;          /* 16                                      ps_flow.yaml:25                   */
    int32_t fvae_enc_dec_hidden;  /* 192                                                                      */
    int32_t fvae_kernel_size;     /* 5                                                                        */
    int32_t fvae_dec_n_layers;    /* 4                                                                        */
    int32_t fvae_enc_n_layers;    /* 8 (posterior encoder: loaded, unused at inference)                       */
    int32_t prior_glow_hidden;    /* 64                                      ps_flow.yaml:32                   */
    int32_t glow_kernel_size;     /* 3                                                                        */
    int32_t prior_glow_n_blocks;  /* 4                                                                        */
    int32_t prior_glow_n_layers;  /* 4, hard-coded at modules/dict_tts/fvae_semantics.py:78-79                */
    int32_t dur_predictor_layers; /* 3                                       ps_flow.yaml:17-22                */
    int32_t dur_predictor_kernel; /* 5                                                                        */
    int32_t dur_chans;            /* 128, hard-coded at modules/portaspeech/model.py:164-169                  */
    int32_t frames_multiple;      /* 4                                       ps_flow.yaml:61                   */
    int32_t language_zh;          /* language == 'zh' -> add_pron_rule       layers/utils.py:109-115           */
    /* HifiGAN generator                                                    egs/egs_bases/tts/vocoder/hifigan.yaml:3-10 */
    int32_t upsample_initial_channel; /* 512 */
    int32_t n_upsamples;              /* 4   */
    int32_t upsample_rates[8];        /* 8,8,2,2   */
    int32_t upsample_kernel_sizes[8]; /* 16,16,4,4 */
    int32_t n_resblock_kernels;       /* 3   */
    int32_t resblock_kernel_sizes[4]; /* 3,7,11 */
    int32_t resblock_dilation_sizes[4][3]; /* (1,3,5) x3 */
    int32_t vocoder_precision;        /* DTTS_VOC_F16 (default) | DTTS_VOC_BF16 | DTTS_VOC_BF16X3 */
    /* FFT block stack (FFTBlocks, modules/fastspeech/tts_modules.py:458-493); width = hidden_size, heads = num_heads */
    int32_t fft_layers;               /* dec_layers 4                          egs/egs_bases/tts/base.yaml:68 */
    int32_t fft_kernel_size;          /* dec_ffn_kernel_size 9                 base.yaml:72                   */
    int32_t fft_use_pos_embed;        /* FFTBlocks(use_pos_embed=True)                                        */
    int32_t fft_use_last_norm;        /* FFTBlocks(use_last_norm=True)                                        */
    int32_t vocoder_unfused;          /* testing aid, DTTS_VOC_BF16 only: 1 = one kernel per convolution instead of the fused ResBlock kernels */
    int32_t decoder_fp32;             /* 1 = FVAE decoder WaveNet on exact fp32 MFMA (round 1); 0 (default) = bf16 hi/lo split operands */
    int32_t vocoder_range_guard;      /* DTTS_VOC_F16: 1 = start with the fp16 range guard on (dtts_vocoder_range_guard) */
    int32_t debug_redzone;            /* testing aid: 1 = memory-safety mode — every workspace buffer and weight pack sits between 4 KiB red
                                         zones, workspaces are filled with 0xFF (NaN) before each forward; dtts_debug_check verifies the zones */
    int32_t tune_flags;               /* A/B switches of tuning experiments (tools/ab_*.sh); 0 = the measured defaults.  The library never
                                         reads the process environment: arithmetic and layout follow this struct alone.
                                         The RELEASE library honours exactly the bits that have a parity / bit-identity test behind them and
                                         refuses every other one (dtts_create: DTTS_E_INVAL):
                                           8  (arithmetic) prior flow launch by launch on the exact-fp32 kernels
                                           9  (schedule)   all ResBlocks of a C <= 64 stage in one launch (less HBM traffic, not faster)
                                           12 (schedule)   the first two ResBlocks of the C = 32 stage in one launch (neutral)
                                           13 (arithmetic) two-product fp16 ups.1 (eats waveform margin)
                                           14 (schedule)   512-row tiles for every k at C = 64
                                           15 (arithmetic) fp32 stream between the three iterations of the C >= 128, k >= 7 ResBlocks (round 5's
                                                           form; the default stores it as fp16: half the bytes, waveform error 5.3e-5 -> 6.7e-5)
                                         Builds made with -DDTTS_ABLATE (`make ablate`, tools/ab_tune.sh) additionally carry the untested
                                         experiments — 0 conv_post as its own kernel, 1 upsamplers without the zero-tap skip, 2 static tile
                                         assignment, 3 no whole-ResBlock fusion at C >= 128, 4 per-launch timer events, 5 128-row tiles for the
                                         narrow upsamplers, 6 raw (unprojected) dictionary table, 7 two-group phase-shifted ResBlock kernel
                                         (rblock2.hip), 10 fp32 g_pre_net, 11 fp32 flow conditioning, 16 strided g_pre_net, 17 fp32 MFMA instead of the
                                         three-piece bf16 products — OR the DTTS_TUNE environment variable in and honour a few schedule-only variables. */
} dtts_config;

/* Fill *cfg with the Biaobei Dict-TTS + HifiGAN defaults listed above. */
DTTS_API void dtts_default_config(dtts_config* cfg);
/* sizeof(dtts_config) as this library was compiled: a binding checks its own mirror of the struct against it. */
DTTS_API int dtts_config_sizeof(void);

/* Create / destroy a context on the current HIP device. */
DTTS_API int dtts_create(const dtts_config* cfg, dtts_handle* out);
DTTS_API void dtts_destroy(dtts_handle h);
DTTS_API const char* dtts_last_error(dtts_handle h); /* h may be NULL: message of the last failed dtts_create */

/*
 * Weight loading — replaces torch's load_state_dict for the two children the path uses:
 *   state_dict['model']      (utils/trainer.py:348-376, keys as in SURVEY.md §8a) -> names "model.<key>"
 *   state_dict['model_gen']  (vocoders/hifigan.py:16-32)                           -> names "vocoder.<key>"
 * One call per tensor, fp32 host data, the library copies it (caller keeps ownership).  Weight-norm pairs
 * arrive as <name>.weight_g / <name>.weight_v and are folded in dtts_finalize_weights (w = g*v/||v||, what
 * remove_weight_norm does at tasks/tts/ps_flow.py:262-268 and modules/hifigan/hifigan.py:144-151).
 * Unknown names (fvae.encoder.*, attn.*, enc_pos_proj.* ... — loaded but unused at inference) are accepted and ignored.
 */
DTTS_API int dtts_load_weight(dtts_handle h, const char* name, const void* host_ptr, const int64_t* shape, int ndim, int dtype);

#define DTTS_PART_ACOUSTIC 1
#define DTTS_PART_VOCODER 2
#define DTTS_PART_FFT 4 /* an FFTBlocks state dict loaded under "fft.<key>"  */
/* Fold, repack into MFMA fragment order and upload.  Fails with DTTS_E_NOENT naming the first missing tensor. */
DTTS_API int dtts_finalize_weights(dtts_handle h, int parts);

/*
 * Acoustic model, phase 1 — replaces run_text_encoder (modules/dict_tts/model.py:84-110): S2PA dictionary
 * encoder, duration predictor, length regulator.  Inputs are the tensors of DictTTSDataset.collater
 * (tasks/tts/dataset_utils.py:264-302):
 *   word_tokens [B,T_w] i64 · keys, values [B,T_w,L_k,gloss_dim] f32 · key_map [B,T_w,L_k] f32 ·
 *   pinyin, pinyin_map [B,T_w,P] i64 · pron_modified [B,T_w] i64 (may be NULL = zeros) ·
 *   mel2word [B,T_m2w] i64 or NULL (teacher-forced durations, model.py:77).
 * On return *T_mel_host is the frame count padded to frames_multiple (model.py:98-100).  This call
 * synchronises the stream once to read that scalar when mel2word is NULL.
 */
DTTS_API int dtts_text2mel_encode(dtts_handle h, const int64_t* word_tokens_dev, const float* keys_dev, const float* values_dev,
                         const float* key_map_dev, const int64_t* pinyin_dev, const int64_t* pinyin_map_dev,
                         const int64_t* pron_modified_dev, const int64_t* mel2word_dev, int T_m2w, int B, int T_w, int L_k,
                         int P, int32_t* T_mel_host, dtts_stream stream);

/*
 * Resident dictionary: the reference ships keys + values [B,T_w,L_k,768] f32 host->device for every
 * batch (tasks/tts/dataset_utils.py:305-330 looks every character up in the ``dict_embed`` indexed dataset and pads) —
 * ~1.5 GB at B=60.  Uploaded once, the ragged table (one entry per dictionary word: gloss embeddings [L_e,768],
 * key_map [L_e], pinyin / pinyin_map [P_e]; items of data_gen/tts/binarizer_zh.py:301-309) stays in HBM and a batch
 * carries only ids.  All pointers are HOST pointers; values may be NULL (the reference stores value == key).
 * What stays resident is the PROJECTED table: K = k_transform(key), V = v_transform(value), hidden_size floats per row each
 * (modules/dict_tts/layers/dict_encoder.py:36-39 projects every gloss row of every batch; here once, on the device, at upload):
 * 2 * hidden_size * 4 = 1,536 B per gloss row instead of 3,072 (6,144 with separate values), and the id path computes
 * logits = K . q and context = Wo sum_l w_l V_l in the reference's own association order.  Consequences: the acoustic weights
 * must be finalized BEFORE this call (DTTS_E_STATE otherwise), and re-loading them requires a new upload.
 */
DTTS_API int dtts_dict_table_upload(dtts_handle h, int n_entries, const int32_t* tok_off_host, const float* keys_host,
                           const float* values_host, const float* key_map_host, const int32_t* pin_off_host,
                           const int64_t* pinyin_host, const int64_t* pinyin_map_host);
/*
 * dtts_text2mel_encode with the dictionary tensors replaced by entry ids [B,T_w] i32 (device): >= 0 a table entry,
 * -1 the BOS / last row the collater pads onto every sentence (zero vectors, key_map = pinyin_map = 1), -2 batch
 * padding.  L_k / P are the batch maxima the collated tensors would have had (they size dict_attn / pron_attn).
 * Results are identical to dtts_text2mel_encode on the tensors collated from the same table.
 */
DTTS_API int dtts_text2mel_encode_ids(dtts_handle h, const int64_t* word_tokens_dev, const int32_t* entry_ids_dev,
                             const int64_t* pron_modified_dev, const int64_t* mel2word_dev, int T_m2w, int B, int T_w, int L_k,
                             int P, int32_t* T_mel_host, dtts_stream stream);

/*
 * Acoustic model, phase 2 — replaces the gather-expand and run_decoder (model.py:105-121,
 * fvae_semantics.py:109-115): z_p [B,latent,T_mel/4] f32 is the prior sample (the reference draws it from the
 * CPU RNG; here it is an explicit input for parity runs) or NULL: N(0,1) drawn on the device (counter-based, a new stream
 * per call).  mel_out [B,T_mel,80] f32.
 */
DTTS_API int dtts_text2mel_decode(dtts_handle h, const float* z_p_dev, float* mel_out_dev, dtts_stream stream);

/*
 * Single-call forms and names of SURVEY.md 8(b).  dtts_text2mel_forward = dtts_text2mel_encode + dtts_text2mel_decode +
 * the two fetches every inference consumer needs, with the outputs in CAPACITY layout because T_mel is not known before the
 * call: mel_out [B, mel_cap, n_mel] (only rows < T_mel of each utterance are written), *T_mel_out (host) = padded frame
 * count, DTTS_E_INVAL if it exceeds mel_cap; z_p [B, latent, z_cap] with z_cap >= T_mel/4, or NULL = the library draws
 * N(0,1) on the device (counter-based; the reference's CPU-RNG draw, fvae_semantics.py:110-111, cannot be reproduced bit
 * for bit, so parity runs pass z_p); pron_attn [B,T_w,P] / dur [B,T_w] may be NULL.  dtts_text2mel_plan is the
 * two-phase entry (= dtts_text2mel_encode: returns T_mel after the duration kernel); dtts_load_weights = dtts_load_weight.
 */
DTTS_API int dtts_load_weights(dtts_handle h, const char* name, const void* host_ptr, const int64_t* shape, int ndim, int dtype);
DTTS_API int dtts_text2mel_plan(dtts_handle h, const int64_t* word_tokens_dev, const float* keys_dev, const float* values_dev,
                       const float* key_map_dev, const int64_t* pinyin_dev, const int64_t* pinyin_map_dev,
                       const int64_t* pron_modified_dev, const int64_t* mel2word_dev, int T_m2w, int B, int T_w, int L_k, int P,
                       int32_t* T_mel_host, dtts_stream stream);
DTTS_API int dtts_text2mel_forward(dtts_handle h, const int64_t* word_tokens_dev, const float* keys_dev, const float* values_dev,
                          const float* key_map_dev, const int64_t* pinyin_dev, const int64_t* pinyin_map_dev,
                          const int64_t* pron_modified_dev, const int64_t* mel2word_dev, int T_m2w, const float* z_p_dev, int z_cap,
                          int B, int T_w, int L_k, int P, float* mel_out_dev, int mel_cap, int64_t* T_mel_out_host,
                          float* pron_attn_dev, float* dur_dev, dtts_stream stream);
DTTS_API int dtts_text2mel_forward_ids(dtts_handle h, const int64_t* word_tokens_dev, const int32_t* entry_ids_dev,
                              const int64_t* pron_modified_dev, const int64_t* mel2word_dev, int T_m2w, const float* z_p_dev,
                              int z_cap, int B, int T_w, int L_k, int P, float* mel_out_dev, int mel_cap, int64_t* T_mel_out_host,
                              float* pron_attn_dev, float* dur_dev, dtts_stream stream);

/* Copy an intermediate of the last encode/decode into a caller buffer (device to device, on the stream). */
#define DTTS_OUT_PRON_ATTN 1        /* [B,T_w,P] f32      ret['pron_attn']          */
#define DTTS_OUT_DUR 2              /* [B,T_w] f32        ret['dur'] (log domain)   */
#define DTTS_OUT_MEL2WORD 3         /* [B,T_mel] i64                                */
#define DTTS_OUT_DICT_ATTN 4        /* [B,1,L_k,T_w] f32  ret['dict_attn']          */
#define DTTS_OUT_WORD_ENCODER_OUT 5 /* [B,T_w,hidden] f32                           */
#define DTTS_OUT_X_MASK 6           /* [B,T_mel,1] f32    ret['x_mask']             */
#define DTTS_OUT_CONTEXT 7          /* [B,T_w,hidden] f32 S2PA context              */
#define DTTS_OUT_MEL_LENS 8         /* [B] i32 frames with mel2word > 0 AFTER the padding to frames_multiple (the frames the
                                      reference's B = 1 inference vocodes: an utterance that reaches T_mel keeps its pad frames) */
DTTS_API int dtts_text2mel_fetch(dtts_handle h, int what, void* dst_dev, dtts_stream stream);

/*
 * Length regulator alone — replaces the integer part of add_dur + LengthRegulator.forward
 * (modules/dict_tts/model.py:78-81, modules/fastspeech/tts_modules.py:215-251): dur [B,T_w] f32 (log domain),
 * ilens [B] i32 valid words; d = clamp(round_half_even(exp(dur) - 1), 0), an utterance whose durations are all zero
 * gets ones.  Writes mel2word [B,cap] i64 (1-based word index, 0 = padding; columns beyond the longest utterance
 * are zero) and, on the host, the per-batch maximum frame count (unpadded).  Returns DTTS_E_INVAL if it exceeds cap.
 */
DTTS_API int dtts_length_regulate(dtts_handle h, const float* dur_dev, const int32_t* ilens_dev, int B, int T_w, int64_t* mel2word_dev,
                         int cap, int32_t* T_max_host, dtts_stream stream);

/*
 * Vocoder — replaces HifiGanGenerator.forward as used by HifiGAN.spec2wav (vocoders/hifigan.py:54-62), for a
 * batch: mel [B,T,80] f32 (the layout of ret['mel_out'] and of spec2wav's argument), lens [B] i32 valid
 * frames per utterance (NULL = T for all).  wav [B, T*hop] f32; utterance b is exactly what the reference
 * produces for mel[b,:lens[b]] alone, samples past lens[b]*hop are zero.  B <= DTTS_MAX_VOCODER_BATCH per call in the fused modes.
 */
#define DTTS_MAX_VOCODER_BATCH 2048
DTTS_API int dtts_hifigan_forward(dtts_handle h, const float* mel_dev, const int32_t* lens_dev, int B, int T, float* wav_dev,
                         dtts_stream stream);
DTTS_API int dtts_hifigan_hop(dtts_handle h); /* product of upsample_rates (256) */

/*
 * FastSpeech FFT block stack — replaces FFTBlocks.forward (modules/fastspeech/tts_modules.py:495-523) at inference:
 * x [B,T,hidden] f32 -> y [B,T,hidden] f32.  lens [B] i32 = valid frames per utterance, or NULL: derived from the
 * values as the reference does (frames whose |x| sums to zero are padding; padding must be a suffix).
 * pos_table [n_pos][hidden] f32 = SinusoidalPositionalEmbedding.weights (modules/commons/common_layers.py:110-127,
 * row 0 = padding) when the stack was built with use_pos_embed, n_pos > T; otherwise NULL.  Positions follow
 * make_positions(x[...,0]) (utils/tts_utils.py:6-18): a frame whose first channel is exactly 0 gets row 0.
 * Weights: dtts_load_weight("fft.<key of FFTBlocks.state_dict()>"), dtts_finalize_weights(h, DTTS_PART_FFT).
 */
DTTS_API int dtts_fft_blocks_forward(dtts_handle h, const float* x_dev, const int32_t* lens_dev, const float* pos_table_dev, int n_pos,
                            int B, int T, float* y_dev, dtts_stream stream);

/*
 * Output side — replaces the sample conversion of save_wav (utils/audio.py:11-16; called from after_infer,
 * tasks/tts/dict_tts.py:282-283) for a batch, on the device, so that the device->host copy is int16:
 * wav [B, T*hop] f32 as dtts_hifigan_forward wrote it, lens [B] i32 valid frames (NULL = T for all);
 * per utterance over its own lens[b]*hop samples: norm != 0 -> w / max|w| ; w * 32767 (fp32) ; truncating cast.
 * out [B, T*hop] i16, samples past an utterance's end are 0.
 */
DTTS_API int dtts_wav_to_int16(dtts_handle h, const float* wav_dev, const int32_t* lens_dev, int B, int T, int norm, int16_t* out_dev,
                      dtts_stream stream);

/* Seed of the device-side prior sample (z_p == NULL in dtts_text2mel_decode / _forward*).  Every context starts from a seed mixed from
 * the time, the process id, the device and a per-process instance count, so that data-parallel ranks and restarts draw different
 * noise (the reference draws from torch's global RNG, modules/dict_tts/fvae_semantics.py:110-111); setting it makes a run repeatable. */
DTTS_API int dtts_set_noise_seed(dtts_handle h, uint64_t seed);

/* fp16 range guard (DTTS_VOC_F16), the CENSUS form.  The ResBlock convolutions convert their activations to fp16: |v| > 65504 becomes
 * +-inf, where the reference computes in fp32 (modules/hifigan/hifigan.py:51-58).  While the guard is on, the fused ResBlock kernels run an
 * instantiation that COUNTS such activations at every one of the 72 conversion points (a few % slower); dtts_vocoder_clamped returns the
 * count accumulated since the last reset (it synchronises `stream`).  A testing / diagnosis aid: the always-on detector below needs no mode. */
DTTS_API int dtts_vocoder_range_guard(dtts_handle h, int enable);
DTTS_API int dtts_vocoder_clamped(dtts_handle h, int64_t* count, int reset, dtts_stream stream);

/* fp16 validity of DTTS_VOC_F16 as a DECISION, in two parts (the reference computes the ResBlocks in fp32, modules/hifigan/hifigan.py:51-58):
 *
 * (1) ALWAYS-ON detector.  The fp16 conversion of the ResBlock operands does not saturate: an activation beyond +-65504 becomes +-inf,
 *     every sum it enters is inf / NaN from there on, and it reaches conv_post as a non-finite pre-tanh value.  The conv_post epilogue of
 *     EVERY forward (release instantiations included, no mode to switch on) writes NaN for such a sample instead of tanh's plausible +-1
 *     and counts it; dtts_hifigan_forward copies the running count to pinned host memory behind its last kernel.  dtts_vocoder_nonfinite
 *     returns that word WITHOUT synchronising: once the caller has synchronised with the forward's stream (it must, to read the waveform)
 *     the value covers that forward.  The count is cumulative over the context's life; compare it with the value before the call.  A call
 *     that overflowed is therefore ALWAYS reported (dict_tts_amd/vocoder.py redoes it in DTTS_VOC_BF16X3, or raises when fp16 was demanded).
 * (2) STATIC bound, computed by dtts_finalize_weights from the folded weights.  worst_case: an upper bound of every value the ResBlocks round
 *     to fp16, for ANY mel with |mel| <= mel_abs_max (per-channel L1 propagation: |b| + sum|w| * bound_in through all 72 convolutions; the
 *     reference's log10-mel lies in [-6, 1.5], egs/egs_bases/tts/base.yaml:59-60).  worst_case < 65504 PROVES that no call in that range can
 *     overflow.  rms_estimate: the propagated root-mean-square of the largest such channel (independence assumed: an estimate, not a bound);
 *     a checkpoint whose estimate already nears 65504 / 16 should not be run in fp16 at all.  For trained generators the worst case is
 *     astronomically loose (it compounds sum|w| ~ 10-40 per convolution), so part (1) is what protects them.  Either pointer may be NULL.
 *     Other precisions: both 0 (bf16 has fp32's exponent range). */
DTTS_API int dtts_vocoder_nonfinite(dtts_handle h, int64_t* count);
DTTS_API int dtts_vocoder_fp16_bound(dtts_handle h, float mel_abs_max, double* worst_case, double* rms_estimate);

/* Memory-safety mode (dtts_config.debug_redzone = 1; a testing aid with no counterpart in the reference — the kernels behind this ABI
 * address raw HBM).  Every workspace buffer of the last encode / decode / vocoder / FFT-block call and every weight pack / table sits
 * between two 4 KiB red zones filled with 0xFF, and the workspaces are filled with 0xFF (NaN) before each forward.  dtts_debug_check
 * synchronises `stream` and counts the red-zone bytes that are no longer 0xFF (= an out-of-range WRITE; dtts_last_error names the first
 * damaged zone); an out-of-range or stale READ that is consumed shows up as NaN in the outputs.  DTTS_E_STATE without the mode. */
DTTS_API int dtts_debug_check(dtts_handle h, int64_t* damaged_bytes, dtts_stream stream);
/* self-test of the mode: damages one red-zone byte (as an off-by-one store would); the next dtts_debug_check must report it */
DTTS_API int dtts_debug_poke(dtts_handle h, dtts_stream stream);

/*
 * Instrumentation used by bench.py: accumulated device time (hipEvent pairs recorded on the caller's stream
 * around every launch of one kernel family) since the last reset.
 */
#define DTTS_TIMER_VOC_CONV 1   /* the vocoder's MFMA convolution kernel (dominant kernel) */
#define DTTS_TIMER_S2PA 2       /* the S2PA dictionary-attention kernel                    */
/* stage spans under the names of the reference's profile_infer timers (utils.Timer): one event pair per call, launches = calls */
#define DTTS_TIMER_STAGE_ENCODER 3      /* 'encoder'      modules/dict_tts/model.py:50  = dtts_text2mel_encode* (device span, incl. the T_mel sync wait) */
#define DTTS_TIMER_STAGE_DICT_ENCODER 4 /* 'dict_encoder' modules/dict_tts/model.py:86  = embedding, both encoders, S2PA                                   */
#define DTTS_TIMER_STAGE_FVAE 5         /* 'fvae'         modules/dict_tts/model.py:57  = dtts_text2mel_decode (gather-expand + prior flow + decoder)       */
#define DTTS_TIMER_STAGE_HIFIGAN 6      /* 'hifigan'      vocoders/hifigan.py:59        = dtts_hifigan_forward                                              */
#define DTTS_TIMER_COUNT 7
DTTS_API int dtts_timer_enable(dtts_handle h, int which);
DTTS_API int dtts_timer_read(dtts_handle h, int which, double* ms_total, int64_t* launches); /* synchronises */
DTTS_API int dtts_timer_reset(dtts_handle h);

#ifdef __cplusplus
}
#endif
#endif
