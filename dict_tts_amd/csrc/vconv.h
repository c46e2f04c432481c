// HifiGAN bf16 convolution kernel: see vconv.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dtts {

struct VConvParams {
    const unsigned short* x;  // bf16 [B][T][ldx], already activated; ldx = C_in_pad (multiple of 8)
    int ldx;
    const float* xf;          // waveform-exact mode (x == null): fp32 [B][T][ldx] NOT activated; leaky_relu(in_slope) + hi/lo split while staging
    float in_slope;           //   slope of that leaky_relu (1 = identity)
    int C_in;                 //   valid input channels (multiple of 4; channels C_in..C_in_pad are zero)
    const uint4* wlo;         //   bf16 lo pack (w - bf16(w)), same layout as w
    int h2;                   //   1: fp16 two-product form (vconv.hip H2): w is a SINGLE fp16 pack, wlo unused
    const uint4* w;           // packed bf16 weights (context.hip:pack_conv)
    const float* bias;        // [C_out_pad] (zero padded) or null
    const int* lens;          // [B] valid rows (stride-1 convs: input rows == output rows); null -> T
    int B, T, C_in_pad, C_out, C_out_pad, K, dil, pad;
    float* yf;                // fp32 result [B][T][ldyf] or null
    int ldyf;
    unsigned short* ya;       // bf16 leaky_relu(result, slope) [B][T][ldya] or null (slope 1 = identity)
    int ldya;
    float slope;
    const float* res;         // fp32 residual(s), same row indexing
    int ldres;
    const float* res2;
    int ldres2;
    // WaveNet forms (the FVAE decoder's layers on split operands, X3 only)
    int gate_H;               // > 0: packed co-tiles alternate (tanh tile, sigmoid tile); out[c] = tanh(a_t + b_t + cond_t) * sigmoid(a_s + b_s + cond_s),
                              //      c < gate_H; bias / gbias are in LOGICAL order [tanh gate_H | sigmoid gate_H] and read through gbias
    const float* gbias;
    const float* cond;        // gated: conditioning [B][T][ld_cond], channels cond_coff + (c | gate_H + c)
    int ld_cond, cond_coff;
    const long long* cond_m2w; // optional [B][T]: the conditioning row of frame (b, t) is cond row b * cond_Tw + m2w (1-based word index), or row 0
    int cond_Tw;              //           for m2w outside 1..cond_Tw (padding frames: row 0 holds conv(0) = the bias) — the gather-expand of the
                              //           word-level conditioning folded into the epilogue instead of a [B*T, ld_cond] tensor in HBM
    int split;                // > 0: output channels >= split go to the second segment (column - split): WaveNet res / skip
    float* yf2;
    int ldyf2;
    const float* res_b;       // residual of the second segment
    int ldres_b;
    int small_tiles;          // X3: 64-row tiles (4 workgroups / CU) for the memory-bound narrow upsamplers (C_out_pad 128 / 64)
    int in_half;              // the first half of the INPUT channels skips the first tap, the second half the last (the polyphase g_pre_net)
    int poly_half;            // PackedConv::poly_half: waves in the first half of the packed channels skip the last tap, the others the first
    float div;                // 1 or num_kernels (true division)
    int post_tanh;
    unsigned* bad;            // with post_tanh: device counter of non-finite pre-tanh values (the always-on overflow detector), or null
    int dbg;                  // -DDTTS_ABLATE builds only (DTTS_VCONV_DBG): 1 = skip the contraction, 2 = skip the epilogue, 4 = skip staging
};

hipError_t vconv_launch(const VConvParams& p, hipStream_t stream);
hipError_t f32_to_bf16_pad_launch(const float* x, unsigned short* y, long long rows, int C, int C_pad, hipStream_t s);

} // namespace dtts
