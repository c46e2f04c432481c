// Shared device helpers of the fused HifiGAN kernels (rblock.hip, vpair.hip): bf16 conversion, the weight-fragment
// ring preload and the static-offset MFMA contraction loop over an LDS activation tile.
#pragma once
// Cache policy of the fused vocoder kernels' global accesses (buffer-instruction aux bits on gfx950: 1 = sc0, 2 = nt, 16 = sc1).
// Their results are consumed by the NEXT launch, ~1 ms and ~1 GB of traffic later: streaming them (nt + sc1) keeps them from evicting
// what the kernel re-reads — the x tile between staging and the epilogue's residual, the weight fragments.  Same-box A/B
// (LABNOTES (yy)): stores 0 -> 18: -0.9 %; + the read-once operands (the stage sum, the epilogue's last read of x): -1.2 %; streaming
// rblock's x loads as well: +2 % (worse: neighbouring tiles' halos re-read them).
#ifndef VP_ST_AUX
#define VP_ST_AUX 18   // y / activated-copy stores
#endif
#ifndef VP_LD_AUX
#define VP_LD_AUX 18   // the stage sum read once by the accumulating ResBlocks
#endif
#ifndef VP_XI_AUX
#define VP_XI_AUX 2    // vpair's epilogue: the last read of the x tile
#endif
#ifndef RB_X_AUX
#define RB_X_AUX 0     // rblock's residual-stream loads: cached (halo rows are shared with the neighbouring tiles)
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "voc_el.h"

namespace dtts {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ unsigned rf2bf(float f) {  // round-to-nearest-even fp32 -> bf16 bits (hardware convert)
    const __bf16 h = (__bf16)f;
    return (unsigned)__builtin_bit_cast(unsigned short, h);
}

typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// two fp32 -> one dword of two bf16 (round-to-nearest-even), a single v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pack2bf(float a, float b) {
    const bf16x2_t v = __builtin_convertvector(f32x2_t{a, b}, bf16x2_t);
    return __builtin_bit_cast(unsigned, v);
}
// leaky_relu for 0 <= slope <= 1 as max(a, a * slope): v_mul + v_max (the asm keeps hipcc from adding the
// canonicalising v_max a,a in front of fmaxf; NaNs do not occur on this path)
__device__ __forceinline__ float lrelu(float a, float slope) {
    float r;
    const float b = a * slope;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// ---- element type of the 16-bit MFMA operands.  EL_BF16: v_mfma_f32_32x32x16_bf16 (8-bit significand); EL_F16:
// v_mfma_f32_32x32x16_f16 (11-bit significand, same rate) — the ResBlock stages of the waveform-exact vocoder mode.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;

template <int EL>
__device__ __forceinline__ unsigned pack2(float a, float b) {   // two fp32 -> one dword of two 16-bit values, round-to-nearest-even
    if constexpr (EL == EL_F16) {
        const f16x2_t v = __builtin_convertvector(f32x2_t{a, b}, f16x2_t);   // v_cvt_pk_f16_f32
        return __builtin_bit_cast(unsigned, v);
    } else {
        return pack2bf(a, b);
    }
}
// leaky_relu feeding a 16-bit operand.  fp16 saturates at 65504 instead of overflowing to inf: median(a, a*slope, 65504)
// = min(max(a, a*slope), 65504) for every a > -655040 — still two instructions (v_mul + v_med3).
template <int EL>
__device__ __forceinline__ float lrelu_op(float a, float slope) {
    if constexpr (EL == EL_F16) return __builtin_amdgcn_fmed3f(a, a * slope, 65504.f);
    else return lrelu(a, slope);
}
// four consecutive channels -> two dwords of 16-bit leaky_relu(v, slope): the slope products as two packed fp32 multiplies
// (v_pk_mul_f32), the max / saturation per value, two packed converts — 8 VALU instructions for 4 values, same arithmetic as
// pack2<EL>(lrelu_op<EL>(.), .) value by value (10)
template <int EL>
__device__ __forceinline__ uint2 act4(const f32x4& v, float slope) {
    const f32x2_t a = {v[0], v[1]}, b = {v[2], v[3]};
    const f32x2_t ta = a * slope, tb = b * slope;
    if constexpr (EL == EL_F16) {
#ifdef DTTS_ACT_F32   // round 3's form: leaky_relu in fp32 (v_pk_mul_f32 + v_med3 saturating at 65504), then the conversion: 8 VALU per 4 values
        return make_uint2(pack2<EL>(__builtin_amdgcn_fmed3f(a[0], ta[0], 65504.f), __builtin_amdgcn_fmed3f(a[1], ta[1], 65504.f)),
                          pack2<EL>(__builtin_amdgcn_fmed3f(b[0], tb[0], 65504.f), __builtin_amdgcn_fmed3f(b[1], tb[1], 65504.f)));
#else
        // convert first, then leaky_relu on PACKED fp16 pairs: v_cvt_pk_f16_f32, v_pk_mul_f16, v_pk_max_f16 = 6 VALU per 4 values (the
        // activation rewrites of the narrow ResBlock kernels are VALU-bound, LABNOTES round 4 (C)).  The slope is fp16(0.1) and negative
        // values round twice: simulated waveform error 5.19e-5 -> 5.28e-5 (tools/precision_sim.py arithmetic).  NO saturation, on purpose: a
        // pre-activation beyond the fp16 range becomes +-inf (also a NEGATIVE one below -65504, whose exact leaky_relu would still be
        // representable), the inf / NaN it makes of every later sum reaches conv_post, and the always-on detector there (rblock.hip /
        // vconv.hip: non-finite pre-tanh value) reports the call — no overflow returns plausible-looking samples.  ovf4 below counts
        // exactly these values.
        const _Float16 hs = (_Float16)slope;   // (a compile-time constant at every call site: folds)
        const f16x2_t slope2 = {hs, hs};
        const f16x2_t ha = __builtin_convertvector(a, f16x2_t), hb = __builtin_convertvector(b, f16x2_t);
        const f16x2_t ra = __builtin_elementwise_max(ha, ha * slope2), rb = __builtin_elementwise_max(hb, hb * slope2);
        (void)ta;
        (void)tb;
        return make_uint2(__builtin_bit_cast(unsigned, ra), __builtin_bit_cast(unsigned, rb));
#endif
    } else {
        float r0, r1, r2, r3;
        asm("v_max_f32 %0, %1, %2" : "=v"(r0) : "v"(a[0]), "v"(ta[0]));
        asm("v_max_f32 %0, %1, %2" : "=v"(r1) : "v"(a[1]), "v"(ta[1]));
        asm("v_max_f32 %0, %1, %2" : "=v"(r2) : "v"(b[0]), "v"(tb[0]));
        asm("v_max_f32 %0, %1, %2" : "=v"(r3) : "v"(b[1]), "v"(tb[1]));
        return make_uint2(pack2<EL>(r0, r1), pack2<EL>(r2, r3));
    }
}

// fp16 range guard (GUARD instantiations of vpair / rblock, DTTS_VOC_F16 only): how many of four pre-activation values the 16-bit
// conversion of act4 turns into +-inf — the conversion comes FIRST there, so |v| > 65504 overflows on either side (round 3's form,
// -DDTTS_ACT_F32, applied leaky_relu in fp32 first: v > 65504 saturated, v * slope < -65504 overflowed).  The reference computes these
// convolutions in fp32 (modules/hifigan/hifigan.py:51-58): a non-zero count means the fp16 mode is not valid for this checkpoint /
// input, and the caller falls back to DTTS_VOC_BF16X3 (dict_tts_amd/vocoder.py).
__device__ __forceinline__ int ovf4(const f32x4& v, float slope) {
    int n = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#ifdef DTTS_ACT_F32
        n += (v[e] > 65504.f || v[e] * slope < -65504.f) ? 1 : 0;
#else
        n += (__builtin_fabsf(v[e]) > 65504.f) ? 1 : 0;
#endif
    }
    return n;
}

template <int EL>
__device__ __forceinline__ f32x16 mfma16(const uint4& a, const uint4& b, const f32x16& c) {
    if constexpr (EL == EL_F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// Tuning ablations (skip a phase of a kernel) exist only in builds made with -DDTTS_ABLATE; in the release library the
// tests below are compile-time false and the branches fold away.
#ifdef DTTS_ABLATE
#define DTTS_DBG(p, bit) ((p).dbg & (bit))
#else
#define DTTS_DBG(p, bit) 0
#endif

constexpr int RB_GUARD = 40;  // zero rows on both sides of the LDS tile (>= max pad 25 + one padded tap + one prefetched tap, dilation 5)

// acc += W * act, all taps; act is the LDS tile (bf16, pitch PITCH), weights in fragment order [step][co-tile][lane]
// first PF = 3 weight fragments of a convolution (issued early: before the barriers / activation writes that precede it)
template <int NT>
__device__ __forceinline__ void rb_preload(uint4 (&ring)[4][NT], const uint4* w, int kg_stride) {
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int n = 0; n < NT; ++n) ring[s][n] = w[(size_t)s * kg_stride + n * 64];
}

// acc += W * act over all taps.  Steps are processed 4 at a time (= TU taps); inside a group every LDS / global
// offset is a compile-time immediate off two VGPR bases that advance once per group, so the loop body is MFMAs,
// ds_read_b128, global_load_dwordx4 and ~4 address instructions.  Weight fragments run 3 steps ahead (register ring),
// activation fragments 1 step ahead.  The packed weights carry >= 4 zero steps of slack, the LDS tile >= one extra tap
// of guard rows, so the prefetches past the last step need no clamping.
// One group of 4 k-steps of the contraction (see rb_contract).  CINIT: the very first MFMA of every accumulator tile
// takes its C operand from cinit[n] (the bias pattern of this lane's 16 channel slots, identical for every row tile), so
// the accumulators need no initialisation pass at all.
// MH > 1: the wave owns MH * MT row tiles, processed as MH passes of MT tiles per weight fragment (pass h covers rows
// h * MT * 32 ...): a weight fragment is fetched once per step and used for MH * MT MFMAs, while only 2 * MT activation
// fragments are live at a time.
// XA1 (round 5 (Y); what every vpair / rblock instantiation uses): ONE set of activation fragments.  Instead of reading the next (step, pass)'s MT fragments in one
// burst at the top of a step (the double buffer), row tile m's next fragment is read right behind the MFMAs that consumed the current one and lands while the other
// row tiles' MFMAs run: eight waves' bursts no longer queue on the CU's LDS pipe in front of the matrix pipe (-1.5 ... -5.9 % per kernel), MT * 4 registers less.
// The scheduling barriers around the read keep it where it is written (+0.4 % without them).  (The line "(NT == 1 here)" below: the read follows the LAST co-tile's MFMA.)
template <int EL, int MT, int NT, int NKG, int PITCH, bool CINIT, int MH = 1, bool XA1 = false>
__device__ __forceinline__ void rb_group(f32x16 (&acc)[MH * MT][NT], const f32x16 (&cinit)[NT], uint4 (&ring)[4][NT], uint4 (&xa)[XA1 ? 1 : 2][MT],
                                         const char* act, const uint4* wpf, int xb, int dilP, int g) {
    constexpr int GPT = (NKG >= 4) ? NKG / 4 : 1;     // groups per tap
    constexpr int KGS = (NKG / 2) * 64;               // uint4 elements between consecutive steps (= NCT * 64, NCT = NKG / 2)
    constexpr int HSTRIDE = MT * 32 * PITCH;          // LDS bytes between two passes
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int h = 0; h < MH; ++h) {
            if (h == 0) {
#pragma unroll
                for (int n = 0; n < NT; ++n) ring[(u + 3) & 3][n] = wpf[u * KGS + n * 64];
            }
            int xa1_off = 0;
            {   // activation fragments of the next (step, pass)
                int off;
                if (h + 1 < MH) {
                    // same step, next pass
                    if constexpr (NKG >= 4) off = xb + ((g * 4 + u) % NKG) * 32 + (h + 1) * HSTRIDE;
                    else off = xb + (u / NKG) * dilP + (u % NKG) * 32 + (h + 1) * HSTRIDE;
                } else if constexpr (NKG >= 4) {
                    const int kgn = (g * 4 + u + 1);            // k-group index within the tap (may be NKG: next tap)
                    off = (u == 3 && g == GPT - 1) ? xb + dilP : xb + (kgn % NKG) * 32;
                } else {
                    const int un = u + 1;                        // step within the group of TU taps
                    off = xb + (un / NKG) * dilP + (un % NKG) * 32;
                }
                if constexpr (!XA1) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) xa[(u * MH + h + 1) & 1][m] = *(const uint4*)(act + off + m * 32 * PITCH);
                } else xa1_off = off;
            }
#if defined(RB_WINO_PROBE) && RB_WINO_PROBE >= 2   // TIMING PROBE ONLY (wrong results): the input transform of Winograd F(2,3) — one packed fp16 add per dword of every fragment
            if constexpr (NKG >= 8) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    uint4& f = xa[(u * MH + h) & 1][m];
                    asm volatile("v_pk_add_f16 %0, %0, %4\n\tv_pk_add_f16 %1, %1, %4\n\tv_pk_add_f16 %2, %2, %4\n\tv_pk_add_f16 %3, %3, %4"
                                 : "+v"(f.x), "+v"(f.y), "+v"(f.z), "+v"(f.w) : "v"(0));
                }
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    if (CINIT && u == 0) acc[h * MT + m][n] = mfma16<EL>(ring[u][n], xa[XA1 ? 0 : (u * MH + h) & 1][m], cinit[n]);
                    else acc[h * MT + m][n] = mfma16<EL>(ring[u][n], xa[XA1 ? 0 : (u * MH + h) & 1][m], acc[h * MT + m][n]);
                    if constexpr (XA1) {   // row tile m's fragment of the next step, behind the MFMAs that read the current one (NT == 1 here)
                        if (n == NT - 1) {
                            __builtin_amdgcn_sched_barrier(0);
                            xa[0][m] = *(const uint4*)(act + xa1_off + m * 32 * PITCH);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// acc (+)= W * act over all taps.  Steps are processed 4 at a time (= TU taps); inside a group every LDS / global
// offset is a compile-time immediate off two bases that advance once per group, so the loop body is MFMAs,
// ds_read_b128, global_load_dwordx4 and ~4 address instructions.  Weight fragments run 3 steps ahead (register ring),
// activation fragments 1 step ahead.  The packed weights carry >= 4 zero steps of slack, the LDS tile >= one extra tap
// of guard rows, so the prefetches past the last step need no clamping.  CINIT: acc = cinit + W * act (acc not read).
template <int EL, int MT, int NT, int NKG, int PITCH, bool CINIT = false, int MH = 1, bool XA1 = false>
__device__ __forceinline__ void rb_contract(f32x16 (&acc)[MH * MT][NT], uint4 (&ring)[4][NT], const char* act, int xrow0, const uint4* w,
                                            int S, int dilP, int kg_stride_unused, const f32x16 (*cinit)[NT] = nullptr) {
    constexpr int TU = (NKG >= 4) ? 1 : 4 / NKG;      // taps per group of 4 steps
    constexpr int GPT = (NKG >= 4) ? NKG / 4 : 1;     // groups per tap
    constexpr int KGS = (NKG / 2) * 64;
    static_assert(MH == 1 || (MH & 1) == 0, "the activation double buffer alternates per pass: MH must be 1 or even");
    uint4 xa[XA1 ? 1 : 2][MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) xa[0][m] = *(const uint4*)(act + xrow0 + m * 32 * PITCH);
    const uint4* wpf = w + 3 * KGS;                   // prefetch pointer, 3 steps ahead
    int xb = xrow0;                                   // LDS byte offset of (tap of this group, kg 0)
    int g = 0;
    int s0 = 0;
    if constexpr (CINIT) {
        if (S > 0) {
            rb_group<EL, MT, NT, NKG, PITCH, true, MH, XA1>(acc, *cinit, ring, xa, act, wpf, xb, dilP, 0);
            s0 = 4;
            wpf += 4 * KGS;
            if constexpr (NKG >= 4) {
                if (++g == GPT) {
                    g = 0;
                    xb += dilP;
                }
            } else {
                xb += TU * dilP;
            }
        } else {   // ablation path (no contraction): acc = cinit
#pragma unroll
            for (int m = 0; m < MH * MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = (*cinit)[n];
        }
    }
    for (; s0 < S; s0 += 4) {
        const f32x16(&dummy)[NT] = *(const f32x16(*)[NT])acc[0];
        rb_group<EL, MT, NT, NKG, PITCH, false, MH, XA1>(acc, dummy, ring, xa, act, wpf, xb, dilP, g);
        wpf += 4 * KGS;
        if constexpr (NKG >= 4) {
            if (++g == GPT) {
                g = 0;
                xb += dilP;
            }
        } else {
            xb += TU * dilP;
        }
    }
}

// ---- contraction over the REAL k-steps with a weight ring of RD register sets (fragments RD - 1 steps ahead; the activation fragments one
// step ahead).  rblock2.hip (one wave per SIMD in a matrix phase: a step takes 32 * MT * NT cycles, rb_contract's 3-step prefetch would
// expose the L2 round trip) uses RD = 8; rblock.hip's C = 32 configurations use RD = 4 for the second property:  Steps are the REAL k-steps K * NKG (the zero
// padding of the packs to a multiple of four steps is not computed): the loop is unrolled over RD steps with a uniform exit at every tap
// boundary.  The packs carry >= 8 k-steps of slack and the LDS tile a spare tap of guard rows, so the prefetches past the end need no clamps.
template <int NT, int RD>
__device__ __forceinline__ void rb2_preload(uint4 (&ring)[RD][NT], const uint4* w, int kgs) {
#pragma unroll
    for (int s = 0; s < RD - 1; ++s)
#pragma unroll
        for (int n = 0; n < NT; ++n) ring[s][n] = w[(size_t)s * kgs + n * 64];
}

template <int EL, int MT, int NT, int NKG, int PITCH, int RD, bool FIRST, bool CINIT, bool XA1 = false>
__device__ __forceinline__ bool rb2_group(f32x16 (&acc)[MT][NT], const f32x16 (&cinit)[NT], uint4 (&ring)[RD][NT], uint4 (&xa)[XA1 ? 1 : 2][MT],
                                          const char* act, const uint4* wpf, int xb, int dilP, int left) {   // left: steps still to do (> 0)
    constexpr int KGS = (NKG / 2) * 64;
    static_assert(RD % NKG == 0 && RD % 2 == 0, "a ring turn covers whole taps");
#pragma unroll
    for (int u = 0; u < RD; ++u) {
        if (u && u % NKG == 0 && u >= left) return true;                // (uniform) the last tap is done
#pragma unroll
        for (int n = 0; n < NT; ++n) ring[(u + RD - 1) % RD][n] = wpf[u * KGS + n * 64];
        const int off = xb + ((u + 1) / NKG) * dilP + ((u + 1) % NKG) * 32;
        if constexpr (!XA1) {
#pragma unroll
            for (int m = 0; m < MT; ++m) xa[(u + 1) & 1][m] = *(const uint4*)(act + off + m * 32 * PITCH);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                if (FIRST && CINIT && u == 0) acc[m][n] = mfma16<EL>(ring[u][n], xa[XA1 ? 0 : u & 1][m], cinit[n]);
                else acc[m][n] = mfma16<EL>(ring[u][n], xa[XA1 ? 0 : u & 1][m], acc[m][n]);
                if constexpr (XA1) {   // row tile m's fragment of the next step, behind the MFMAs that read the current one
                    if (n == NT - 1) {
                        __builtin_amdgcn_sched_barrier(0);
                        xa[0][m] = *(const uint4*)(act + off + m * 32 * PITCH);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        __builtin_amdgcn_sched_barrier(0);
    }
    return left <= RD;
}

// acc (+)= W * act over the S = K * NKG real steps; ring holds steps 0 .. RD - 2 on entry (rb2_preload).  CINIT: acc = cinit + W * act.
template <int EL, int MT, int NT, int NKG, int PITCH, int RD, bool CINIT, bool XA1 = false>
__device__ __forceinline__ void rb2_contract(f32x16 (&acc)[MT][NT], uint4 (&ring)[RD][NT], const char* act, int xrow0, const uint4* w, int S,
                                             int dilP, const f32x16 (&cinit)[NT]) {
    constexpr int KGS = (NKG / 2) * 64;
    uint4 xa[XA1 ? 1 : 2][MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) xa[0][m] = *(const uint4*)(act + xrow0 + m * 32 * PITCH);
    const uint4* wpf = w + (RD - 1) * KGS;
    int xb = xrow0;
    if (rb2_group<EL, MT, NT, NKG, PITCH, RD, true, CINIT, XA1>(acc, cinit, ring, xa, act, wpf, xb, dilP, S)) return;
    for (int left = S - RD;; left -= RD) {
        wpf += RD * KGS;
        xb += (RD / NKG) * dilP;
        if (rb2_group<EL, MT, NT, NKG, PITCH, RD, false, CINIT, XA1>(acc, cinit, ring, xa, act, wpf, xb, dilP, left)) return;
    }
}

} // namespace dtts
