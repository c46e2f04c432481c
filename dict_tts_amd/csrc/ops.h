// Non-GEMM kernels of the Dict-TTS path (HBM-bound / launch-bound work): embedding, channel LayerNorm,
// multi-head attention, S2PA dictionary attention, duration -> mel2word, gather-expand.
// All activations are channels-last fp32 [B][T][C].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dtts {

// x[b,t,:] = table[tok[b,t]] * scale; lens[b] = #(tok > 0)          (dict_encoder.py:132-133)
hipError_t embed_launch(const int64_t* tok, const float* table, float scale, float* x, int* lens, int B, int T, int C,
                        int n_rows, hipStream_t s);

// y = LayerNorm_C(x) with biased variance over the channel dim.  Options: mask_in -> rows >= lens[b] of x are
// first zeroed IN PLACE (Encoder.forward's x = x * x_mask, rel_transformer_encoder.py:58); mask_out -> rows
// >= lens[b] of y are zero (dur predictor / last_ln * x_mask).
hipError_t layernorm_launch(float* x, float* y, const float* gamma, const float* beta, float eps, const int* lens,
                            int mask_in, int mask_out, int B, int T, int C, hipStream_t s);

// Multi-head self-attention on a fused qkv buffer [B][T][3C] (q | k | v), window_size=None
// (rel_transformer_encoder.py:132-158): scores/sqrt(dk), masked_fill(mask==0, -1e4), softmax, PV.
hipError_t mha_launch(const float* qkv, float* out, const int* lens, int B, int T, int C, int heads, hipStream_t s);

// S2PA dictionary attention, re-associated so that the gloss embeddings are streamed exactly once
// (dict_encoder.py:32-66; layers/utils.py:40-58,109-115):
//   logits[l] = keys[l,:] . qk          (qk = Wk^T (Wq x * key_size^-0.5), computed by the caller)
//   logits[key_map == 0] = -1e9 ; w = softmax_l ; wv = sum_l w[l] * values[l,:]   (context = Wo Wv wv, by the caller)
//   s_i = sum_l w[l] [key_map == i] ; pron_w[p] = s_{pinyin_map[p]} ; forced rows one-hot (add_pron_rule)
//   pron = sum_p pron_w[p] * pinyin_emb[pinyin[p]]
// dict_attn is kept as [B][T_w][L_k] (a word's weights are one coalesced row); dtts_text2mel_fetch(DTTS_OUT_DICT_ATTN) transposes it into the
// reference's [B,1,L_k,T_w] view on request.
struct S2paArgs {
    const float* qk;       // [B*T_w][D]
    const float* keys;     // [B*T_w][L_k][D]
    const float* values;   // [B*T_w][L_k][D]
    const float* key_map;  // [B*T_w][L_k]
    const int64_t* pinyin; // [B*T_w][P]
    const int64_t* pinyin_map;
    const int64_t* pron_modified; // [B*T_w] or null
    const float* pinyin_emb;      // [n_pinyin][H]
    const int* pm_max;            // device scalar: max(pinyin_map) over the batch
    const int* lens;              // [B] words per utterance, or null: the caller zeroes context for t >= lens[b]
                                  // (dict_encoder.py:140), so the value rows of those words are not streamed
    float* wv;                    // [B*T_w][D]
    float* dict_attn;             // [B][T_w][L_k]  (NOT the reference's transposed view: see above)
    float* pron_attn;             // [B*T_w][P]
    float* pron;                  // [B*T_w][H]
    int B, T_w, L_k, P, D, H, n_pinyin, language_zh;
    // resident-dictionary mode (entry != null): keys/values/key_map/pinyin/pinyin_map above are ignored and every word
    // row is gathered from the device-resident ragged table by its entry id.  entry[row] >= 0: dictionary entry;
    // -1: the BOS / last row the collater pads onto every sentence (zero vectors, key_map and pinyin_map all 1,
    // tasks/tts/dataset_utils.py:287-300); -2: batch padding (everything zero / masked).
    const int* entry;          // [B*T_w]
    const int* t_off;          // [n_entries + 1] gloss-token offsets
    const float* t_keys;       // [sum_L][D]
    const float* t_values;     // [sum_L][D]
    const float* t_key_map;    // [sum_L]
    const int* t_poff;         // [n_entries + 1] pinyin-token offsets
    const int64_t* t_pinyin;   // [sum_P]
    const int64_t* t_pinyin_map;
};
hipError_t s2pa_launch(const S2paArgs& a, hipStream_t s);
hipError_t max_i64_launch(const int64_t* x, long long n, int* out, hipStream_t s);
// table mode: out = max over rows of the entry's max pinyin_map (t_pmmax[e]; 1 for entry -1, 0 for -2)
hipError_t max_entry_pm_launch(const int* entry, const int* t_pmmax, long long n, int* out, hipStream_t s);

// y = a + b (elementwise, n floats, n % 4 == 0)
hipError_t add_launch(const float* a, const float* b, float* y, long long n, hipStream_t s);

// mask rows: y[b,t,:] = t < lens[b] ? x : 0
hipError_t mask_rows_launch(const float* x, float* y, const int* lens, int B, int T, int C, hipStream_t s);

// src_padding = (|x|.sum(-1) == 0) ; ilens[b] = #non-padding rows      (model.py:73, :81)
hipError_t rowcount_nonzero_launch(const float* x, int* ilens, int B, int T, int C, hipStream_t s);

// dur[b,t] = softplus(h[b,t,:] . w + bias) * (t < ilens[b])             (portaspeech/model.py:56,64-65)
hipError_t dur_head_launch(const float* h, const float* w, const float* bias, const int* ilens, float* dur, int B, int T,
                           int C, hipStream_t s);

// d = clamp(round_half_even(exp(dur) - 1), 0) for t < ilens[b]; all-zero utterance -> all ones; starts = exclusive
// prefix sum; total[b] = sum.                                            (model.py:78-81, tts_modules.py:215-251)
hipError_t durations_launch(const float* dur, const int* ilens, int* starts, int* total, int B, int T, hipStream_t s);
// mel2word[b, f] for f < T_mel: word index (1-based) or 0; columns >= T_raw repeat column T_raw-1 (model.py:98-100)
hipError_t mel2word_fill_launch(const int* starts, int* total, const int* ilens, int64_t* m2w, int B, int T_w,
                                int T_raw, int T_mel, hipStream_t s);
// teacher-forced: copy [B,T_in] i64 into [B,T_mel] with the same last-column padding; total[b] = #(m2w > 0)
hipError_t mel2word_copy_launch(const int64_t* src, int64_t* dst, int* total, int B, int T_in, int T_mel, hipStream_t s);
// x[b,f,:] = m2w[b,f] > 0 ? weo[b, m2w-1, :] : 0 ; x_mask[b,f] = m2w > 0   (model.py:101-107, :53)
// zero_row: [C] row written where m2w == 0 instead of zeros (null: zeros); x_mask may be null
hipError_t expand_launch(const float* weo, const int64_t* m2w, float* x, float* x_mask, int B, int T_w, int T_mel, int C,
                         hipStream_t s, const float* zero_row = nullptr);

// save_wav's sample conversion on the device (utils/audio.py:11-16): per utterance b over its n_b = lens[b] * hop valid
// samples: norm -> w / max|w|; w * 32767 in fp32; truncating cast to int16.  Samples past n_b are written as 0.
// amax_bits [B] u32 scratch (only used when norm).
hipError_t wav_to_int16_launch(const float* wav, const int* lens, int hop, int B, long long N, int norm, unsigned* amax_bits,
                               int16_t* out, hipStream_t s);

// FFTBlocks input stage (tts_modules.py:503-509): y[b,t,:] = (x[b,t,:] + alpha * table[pos[b,t]]) * (t < lens[b]) with
// pos = make_positions(x[...,0], 0) = running count of frames whose first channel is non-zero (0 for the others);
// table == null: y = x * (t < lens[b]).  alpha is a device scalar (pos_embed_alpha) or null = 1.
// pos_scratch: [B][T] i32 (used when table != null).
hipError_t fft_input_launch(const float* x, const float* table, int n_pos, const float* alpha, const int* lens, int* pos_scratch,
                            float* y, int B, int T, int C, hipStream_t s);

// [B][C][T] -> [B][T][C] transpose (z_p arrives channels-first as the reference samples it)
// ldT: time pitch of the source (0 = T)
hipError_t transpose_cf_to_cl_launch(const float* x, float* y, int B, int C, int T, hipStream_t s, int ldT = 0);
// y[i] ~ N(0,1), counter-based on (seed, i)
hipError_t normal_fill_launch(float* y, long long n, unsigned long long seed, hipStream_t s);

} // namespace dtts
