// Channels-last 1-D convolution as an implicit GEMM on the gfx950 matrix cores.
//
// One kernel family serves every dense contraction of the Dict-TTS path:
// 1x1 projections, k=3/5/7/11 (dilated) convolutions, the strided g_pre_net, and transposed convolutions
// (rewritten as polyphase ordinary convolutions at pack time).  Activations are [B][T][C] fp32 in HBM, the
// contraction runs on
//   ENG_F32   : v_mfma_f32_32x32x2_f32  (exact fp32 fma chain — acoustic model, protects duration rounding)
//   ENG_BF16  : v_mfma_f32_32x32x16_bf16 (vocoder, bf16 operands / fp32 accumulate)
//   ENG_BF16X3: the same instruction on hi/lo split operands, 3 products (fp32-class vocoder mode)
//   ENG_BF16X6: three bf16 pieces per operand (exact 24-bit split), the 6 products down to 2^-24: the fp32 MFMA's accuracy at 2.7x its
//               rate (gfx950 has no xf32) — conv1d_short_kernel's form of the ENG_F32 layers
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dtts {

enum Engine { ENG_F32 = 0, ENG_BF16 = 1, ENG_BF16X3 = 2, ENG_F16 = 3, ENG_BF16X6 = 4 };   // ENG_F16: packed for the fused vocoder kernels only (vpair / rblock);
                                                                                        // ENG_BF16X6: the short-sequence kernel's form of an ENG_F32 layer (PackedConv::x6)

// One output segment of the epilogue: y[b][t][coff + c] = ((acc + bias) + res + res2) / div
struct ConvSeg {
    float* y;          // destination, row pitch ld (floats)
    const float* res;  // optional residual (same indexing as y, its own pitch)
    const float* res2; // optional second residual
    int ld, ld_res, ld_res2;
    int coff, coff_res, coff_res2;
};

struct ConvParams {
    const float* x;      // [B][T_in][ldx], channels [x_coff, x_coff + C_in)
    int ldx, x_coff;
    long long x_bstride; // elements between batch items
    const void* w_hi;    // packed weights (fragment order), fp32 or bf16
    const void* w_lo;    // bf16x3 / bf16x6: second piece
    const void* w_lo2;   // bf16x6: third piece
    const float* bias;   // [C_out] logical order, may be null
    const int* in_lens;  // [B] valid input rows (rows >= len read as zero); null -> T_in
    const int* out_lens; // [B] valid output rows; null -> T_out
    int B, T_in, T_out;
    int C_in, C_in_pad, C_out, C_out_pad;
    int K, dil, stride, pad;
    int pre_act;         // 0 none; 1 = leaky-relu with pre_slope applied to the input while staging (slope 0 = relu)
    float pre_slope;
    int post_act;        // 0 none, 1 relu, 2 tanh, 3 gelu (erf form)
    int zero_masked;     // write zeros to rows >= out_len (else: leave untouched)
    float out_div;       // 1.0 or e.g. 3.0 (true division, as the reference's xs / num_kernels)
    float out_mul;       // 1.0 or a scale applied after bias (S2PA's q * key_size**-0.5)
    // gated mode (WaveNet layer): packed co-tiles alternate (tanh half, sigmoid half); output has gate_H channels
    int gate_H;          // 0 = off
    const float* cond;   // gated: conditioning [B][T_out][ld_cond] added before the nonlinearity (+cond_coff)
    int ld_cond, cond_coff;
    long long y_bstride_rows; // rows per batch item in every output/residual tensor (T_out)
    int split;           // channels < split go to seg[0], the rest (minus split) to seg[1]
    ConvSeg seg[2];
};

// Host-side description of a packed layer
struct PackedConv {
    void* w_hi = nullptr;
    void* w_lo = nullptr;
    void* x6[3] = {nullptr, nullptr, nullptr};   // ENG_F32 layers: the same weights as three bf16 pieces (k-groups of 16) for conv1d_short_kernel
    float* bias = nullptr;
    int engine = ENG_F32;
    int C_in = 0, C_in_pad = 0, C_out = 0, C_out_pad = 0, K = 1, CK = 64;
    int dil = 1, stride = 1, pad = 0;
    int gate_H = 0;
    int poly_half = 0;        // polyphase form of a k = 2u, pad = u/2 transposed convolution: packed channels of the first half of the
                              // phases have an all-zero LAST tap, those of the second half an all-zero FIRST tap (context.hip:pack_transposed)
    double flops_per_row = 0; // algorithmic 2*MAC per output row (unpadded), for the roofline report
};

// Launch on `stream`.  Returns hipSuccess or the launch error.
hipError_t conv1d_launch(const PackedConv& L, ConvParams p, hipStream_t stream);

// Host packers (fill caller-provided host buffers in fragment order).
// getw(co_logical, ci, tap) -> float.  co_map maps packed co -> logical co (or -1 for padding).
size_t packed_elems(const PackedConv& L);

} // namespace dtts
