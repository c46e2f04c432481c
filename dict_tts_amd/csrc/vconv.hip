// HifiGAN convolution kernel (bf16 MFMA, fp32 accumulate), the dominant kernel of the path.
//
// Differences from the generic conv1d_cl_kernel (conv1d.hip):
//   * the input is the bf16, already-activated tensor the PRODUCING kernel wrote next to its fp32 result
//     (leaky_relu and the bf16 rounding happen exactly once, in the producer's epilogue) - staging is a pure
//     16 B/lane copy of half the bytes;
//   * weights are the A operand, activations the B operand: D[co][t], so a lane owns 4 consecutive output
//     channels of one time row -> 16 B residual loads / fp32 stores and 8 B bf16 stores instead of 4 B ones;
//   * waves are arranged over OUTPUT CHANNELS first (each weight fragment is fetched by exactly one wave of the
//     block), every wave owns MT=4 row tiles, and weight fragments are prefetched through a register ring PF
//     k-steps ahead, so L2 latency is covered by MFMA work instead of being exposed per k-step;
//   * dual-output epilogue: fp32 residual stream + bf16 leaky_relu copy for the next convolution.
// Same packed-weight format as conv1d.hip (context.hip:pack_conv), same contraction, same rounding points.
#include "vconv.h"
// the fp32 result rows are consumed by the next launch: non-temporal stores (rb_common.h: cache policy; -0.5 % same box)
#ifndef VC_NT_STORE
#define VC_NT_STORE 1
#endif
#include "rb_common.h"

#ifndef VC_SB
#define VC_SB 1
#endif
#ifndef VC_SB1
#define VC_SB1 1   // the one-co-tile split-operand configurations (decoder WaveNet): one fragment set, three workgroups per CU
#endif
#ifndef VC_H2_RING
#define VC_H2_RING 2
#endif
namespace dtts {


__device__ __forceinline__ unsigned vf2bf(float f) {  // round-to-nearest-even fp32 -> bf16 bits (hardware convert)
    const __bf16 h = (__bf16)f;
    return (unsigned)__builtin_bit_cast(unsigned short, h);
}

// X3: the waveform-exact mode of the six serial convolutions (conv_pre, the upsamplers, conv_post: 3 % of the FLOPs, 87 % of
// the bf16 rounding noise, tools/precision_sim.py).  The input is the fp32 tensor itself (leaky_relu(in_slope) applied while
// staging), split into bf16 hi + lo LDS tiles; weights arrive as hi + lo packs; three MFMAs per step: Wlo*Xhi + Whi*Xlo +
// Whi*Xhi (16-bit significand products, fp32 accumulation).
// H2 (with X3): two-product form on fp16 operands — activations split into fp16 hi + lo LDS tiles exactly like X3's bf16 pair, weights a
// SINGLE fp16 pack: W * Xlo + W * Xhi.  The weight rounding (11-bit significand) is the only error left; tools/precision_sim.py scheme
// "hh/h": 7e-5 waveform RMS when ups.1 alone runs this way (gate 1e-4, 5.3e-5 with three products everywhere).  fp16 hi saturates at
// 65504 and lo carries the rest, so the pair represents |a| up to 1.3e5.
template <int MT, int NT, int WT, int WC, int CK, bool X3, bool H2 = false>
#ifndef VC_WPE_DEC
#define VC_WPE_DEC 4
#endif
__global__ __launch_bounds__(256, X3 ? (MT <= 2 ? (NT == 2 ? 3 : (CK == 64 && WC == 4 ? VC_WPE_DEC : 4)) : ((VC_SB1 && NT == 1) ? 3 : 2)) : (NT == 1 ? 3 : 2)) void vconv_kernel(const VConvParams p) {
    static_assert(!H2 || X3, "H2 is a variant of the fp32-input path");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PITCH = CK * 2 + 16;
    constexpr int TT = 32 * MT * WT;
    constexpr int CO_T = 32 * NT * WC;
    constexpr int NKG = CK / 16;
    constexpr int R = X3 ? (H2 ? VC_H2_RING : 2) : (NKG < 4 ? NKG : 4);  // weight-fragment ring slots (X3: three MFMAs per fragment pair, one step of prefetch is enough)
    constexpr int PF = R - 1;             // prefetch distance in k-steps

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wt = wave % WT, wc = wave / WT;
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * TT;
    const int len = __builtin_amdgcn_readfirstlane(p.lens ? p.lens[b] : p.T);   // (uniform: it sizes a buffer resource)
    if (t0 >= len) return;
    const int ct0 = blockIdx.y * (CO_T / 32) + wc * NT;
    const int NCT = p.C_out_pad >> 5;
    const int NG = p.C_in_pad >> 4;
    const int rows = TT + (p.K - 1) * p.dil;   // + one spare tap of rows is allocated (never staged, only prefetched)
    const int in0 = t0 - p.pad;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const unsigned short* xb = p.x + (long long)b * p.T * p.ldx;
    const float* xfb = p.xf + (long long)b * p.T * p.ldx;
    const int lo_off = (TT + p.K * p.dil) * PITCH;   // X3: byte offset of the lo tile (the hi tile incl. its spare tap precedes it)
    const int xoff = ((wt * MT) * 32 + (lane & 31)) * PITCH + (lane >> 5) * 16;  // B-operand base of this lane

    // weight-fragment ring.  One running pointer walks the packed weights PF steps ahead of the MFMAs: +kg_stride per
    // step, +tap_jump when the prefetched step wraps to the next tap, so the loop body carries two 64-bit adds instead
    // of div/mod/multiply address arithmetic.  The first PF fragments of a chunk are requested BEFORE its activation
    // tile is staged (L2 latency overlaps the staging loads); the packed buffer has PF+1 steps of slack at its end.
    const size_t kg_stride = (size_t)NCT * 64;
    const size_t tap_jump = (size_t)NG * NCT * 64 - (size_t)NKG * kg_stride;
    constexpr bool WLO = X3 && !H2;   // a lo weight pack exists
    uint4 ring[R][NT], ringl[WLO ? R : 1][NT];
    const long long wlo_d = WLO ? (const char*)p.wlo - (const char*)p.w : 0;   // the lo pack mirrors the hi pack: one pointer walks both
    auto wlo = [&](const uint4* q) { return *(const uint4*)((const char*)q + wlo_d); };
    const uint4* wpf = p.w + (size_t)ct0 * 64 + lane;
    // polyphase upsamplers (k = 2u): this wave's channels use two of the three taps (the third is all zero)
    int tap_lo = 0, tap_hi = p.K;
    if (p.poly_half) {
        const bool second = ct0 * 32 >= (p.C_out_pad >> 1);
        tap_lo = second ? 1 : 0;
        tap_hi = second ? p.K : p.K - 1;
    }
    // in_half (the strided g_pre_net as a 3-tap convolution over 4-row phase groups, context.hip: pack_gpre_poly): the first half of the
    // INPUT channels has an all-zero first tap, the second half an all-zero last tap — the tap range follows the chunk
    auto chunk_taps = [&](int ci0) {
        if (!p.in_half) return;
        const bool second = ci0 >= (p.C_in_pad >> 1);
        tap_lo = second ? 0 : 1;
        tap_hi = second ? p.K - 1 : p.K;
    };
    auto preload = [&](int ci0) {
        chunk_taps(ci0);
        wpf = p.w + (((size_t)tap_lo * NG + (ci0 >> 4)) * NCT + ct0) * 64 + lane;
#pragma unroll
        for (int s = 0; s < PF; ++s) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                ring[s % R][n] = wpf[n * 64];
                if constexpr (WLO) ringl[s % R][n] = wlo(wpf + n * 64);
            }
            wpf += kg_stride;
            if ((s + 1) % NKG == 0) wpf += tap_jump;
        }
    };
    preload(0);
    for (int ci0 = 0; ci0 < p.C_in_pad; ci0 += CK) {
        if (ci0) __syncthreads();
        if constexpr (X3) {   // fp32 in: leaky_relu, bf16 hi / lo split, two LDS tiles; 4 channels per 16 B load
#ifndef VC_U_SMALL
#define VC_U_SMALL 9   // (LABNOTES (L): measured with -DVC_U_SMALL=9, and the default was left at 4 until round 5 (S))
#endif
            // A thread keeps its 4-channel piece and walks the rows in steps of 256 / PIECES.  Buffer loads over the utterance's rows
            // [0, len): a row outside it (t < 0 wraps to a huge unsigned offset, t >= len exceeds num_records) returns zeros = the zero
            // padding, no per-access compare, one VGPR of address per load.  U loads in flight per thread and batch: the narrow 64-row
            // upsampler tiles (8-9 pieces per thread) take ONE batch = one exposed HBM round trip per tile instead of three.
            constexpr int U = MT <= 2 ? VC_U_SMALL : 4, PIECES = CK / 4, RS = 256 / PIECES;
            // (PIECES = 48 at CK = 192: five rows per pass, the last 16 threads of the workgroup stage nothing)
            typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
            const auto rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)xfb, 0, len * p.ldx * 4, 0x00020000);
            const int c = tid % PIECES, r0 = tid / PIECES;
            const bool c_ok = ci0 + c * 4 < p.C_in;
            const int voff0 = ((in0 + r0) * p.ldx + ci0 + c * 4) * 4, vstep = RS * p.ldx * 4;
            const int nk = tid < RS * PIECES ? (rows - r0 + RS - 1) / RS : 0;   // rows this thread stages
            for (int kb = 0; kb < nk; kb += U) {
                u32x4 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (c_ok && kb + u < nk) ? voff0 + (kb + u) * vstep : (int)0x80000000, 0, 0);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (kb + u >= nk) continue;
                    const int r = r0 + (kb + u) * RS;
                    const f32x4 vf = __builtin_bit_cast(f32x4, v[u]);
                    float a[4], lo[4];
                    unsigned hb[4];
                    if constexpr (H2) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            a[e] = lrelu(vf[e], p.in_slope);
                            const _Float16 hh = (_Float16)__builtin_amdgcn_fmed3f(a[e], -65504.f, 65504.f);   // saturate: lo carries the rest
                            hb[e] = (unsigned)__builtin_bit_cast(unsigned short, hh);
                            lo[e] = a[e] - (float)hh;
                        }
                        *(uint2*)(smem + r * PITCH + c * 8) = make_uint2(hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16));
                        *(uint2*)(smem + lo_off + r * PITCH + c * 8) = make_uint2(pack2<EL_F16>(lo[0], lo[1]), pack2<EL_F16>(lo[2], lo[3]));
                        continue;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        a[e] = lrelu(vf[e], p.in_slope);
                        hb[e] = rf2bf(a[e]);
                        lo[e] = a[e] - __builtin_bit_cast(float, hb[e] << 16);
                    }
                    *(uint2*)(smem + r * PITCH + c * 8) = make_uint2(hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16));
                    *(uint2*)(smem + lo_off + r * PITCH + c * 8) = make_uint2(pack2bf(lo[0], lo[1]), pack2bf(lo[2], lo[3]));
                }
            }
        } else if (!DTTS_DBG(p, 4)) {   // stage the activation tile: batches of U independent 16 B loads in flight per thread, then the LDS writes
            constexpr int U = 8, PIECES = CK / 8;
            const int total = rows * PIECES;
            for (int base = tid; base < total; base += 256 * U) {
                uint4 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int idx = base + u * 256;
                    const int r = idx / PIECES, c = idx % PIECES;
                    const int t = in0 + r;
                    v[u] = make_uint4(0, 0, 0, 0);
                    if (idx < total && t >= 0 && t < len) v[u] = *(const uint4*)(xb + (long long)t * p.ldx + ci0 + c * 8);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int idx = base + u * 256;
                    const int r = idx / PIECES, c = idx % PIECES;
                    if (idx < total) *(uint4*)(smem + r * PITCH + c * 16) = v[u];
                }
            }
        }
        __syncthreads();
        // activation fragments are double-buffered in registers: step s+1 is read from LDS before the MFMAs of step s.
        // SB (the two-co-tile split-operand configuration: 128 accumulators + hi / lo rings + two fragment sets = 256 VGPRs and 9-12
        // spilled): ONE fragment set; row tile m's next fragments are read right behind the MFMAs that consumed the current ones and land
        // while the other row tiles' MFMAs execute.
        constexpr bool SB = X3 && ((VC_SB && NT == 2) || (VC_SB1 && NT == 1 && MT == 4));
        constexpr int XB = SB ? 1 : 2;
        uint4 xa[XB][MT], xl[X3 ? XB : 1][MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            xa[0][m] = *(const uint4*)(smem + xoff + tap_lo * p.dil * PITCH + m * 32 * PITCH);
            if constexpr (X3) xl[0][m] = *(const uint4*)(smem + lo_off + xoff + tap_lo * p.dil * PITCH + m * 32 * PITCH);
        }
        const int ntap = DTTS_DBG(p, 1) ? tap_lo : tap_hi;
        const int dilP = p.dil * PITCH;
        int arow = xoff + tap_lo * dilP;
        for (int tap = tap_lo; tap < ntap; ++tap) {
#pragma unroll
            for (int kg = 0; kg < NKG; ++kg) {
#pragma unroll
                for (int n = 0; n < NT; ++n) {   // step s + PF
                    ring[(kg + PF) % R][n] = wpf[n * 64];
                    if constexpr (WLO) ringl[(kg + PF) % R][n] = wlo(wpf + n * 64);
                }
                wpf += kg_stride;
                if ((kg + PF + 1) % NKG == 0) wpf += tap_jump;
                const int nxt = (kg + 1 < NKG) ? arow + (kg + 1) * 32 : arow + dilP;   // LDS tile has one spare tap of rows
                if constexpr (!SB) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        xa[(kg + 1) & 1][m] = *(const uint4*)(smem + nxt + m * 32 * PITCH);
                        if constexpr (X3) xl[(kg + 1) & 1][m] = *(const uint4*)(smem + lo_off + nxt + m * 32 * PITCH);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);  // the prefetches above stay above this step's MFMAs
#pragma unroll
                for (int m = 0; m < MT; ++m) {
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        if constexpr (H2) {   // fp16, two products: the small one first
                            acc[m][n] = mfma16<EL_F16>(ring[kg % R][n], xl[kg & (XB - 1)][m], acc[m][n]);
                            acc[m][n] = mfma16<EL_F16>(ring[kg % R][n], xa[kg & (XB - 1)][m], acc[m][n]);
                            continue;
                        }
                        if constexpr (X3) {   // the two small products first
                            acc[m][n] = mfma16<EL_BF16>(ringl[kg % R][n], xa[kg & (XB - 1)][m], acc[m][n]);
                            acc[m][n] = mfma16<EL_BF16>(ring[kg % R][n], xl[kg & (XB - 1)][m], acc[m][n]);
                        }
                        acc[m][n] = mfma16<EL_BF16>(ring[kg % R][n], xa[kg & (XB - 1)][m], acc[m][n]);
                    }
                    if constexpr (SB) {   // this row tile's fragments of the next step, behind the MFMAs that read the current ones
                        __builtin_amdgcn_sched_barrier(0);
                        xa[0][m] = *(const uint4*)(smem + nxt + m * 32 * PITCH);
                        xl[0][m] = *(const uint4*)(smem + lo_off + nxt + m * 32 * PITCH);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            arow += dilP;
        }
        if (ci0 + CK < p.C_in_pad) preload(ci0 + CK);
    }

    if (DTTS_DBG(p, 2)) {
        if (acc[0][0][0] == 123.456f) p.yf[0] = 1.f;
        return;
    }
    // ---- epilogue.  The accumulators are D[co][t] (lane = one time row, 4 consecutive channels per register
    // quad).  Stored straight from that layout every wave instruction would touch 32 different rows with 32 B each;
    // instead each 32-row slab goes through LDS (the activation tile is dead by now) and leaves as whole rows:
    // consecutive lanes -> consecutive 16 B of one row, for the residual loads and both stores.
    if constexpr (X3) {
        if (p.gate_H || p.split) {
            // ---- WaveNet epilogue (FVAE decoder layers): whole rows through LDS as below; a thread owns 4 consecutive logical
            // channels of one row.  Gated: the tanh quad and its sigmoid partner sit 32 columns apart in the staged row.
            constexpr int EP = CO_T * 4 + 16;
            constexpr int F4 = CO_T / 4, TOTAL = WT * 32 * F4, PER = TOTAL / 256;
            const int co_blk = blockIdx.y * CO_T;
            const int n_out = p.gate_H ? p.gate_H : p.C_out;
            __syncthreads();
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if (m) __syncthreads();
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int col = (wc * NT + n) * 32 + 8 * q + 4 * (lane >> 5);
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[m][n][4 * q + e];
                        *(f32x4*)(smem + (wt * 32 + (lane & 31)) * EP + col * 4) = v;
                    }
                __syncthreads();
#pragma unroll
                for (int u = 0; u < PER; ++u) {
                    const int idx = tid + u * 256;
                    const int rl = idx / F4, c4 = idx % F4;
                    const int t = t0 + ((rl >> 5) * MT + m) * 32 + (rl & 31);
                    const int pcol = co_blk + c4 * 4;       // packed column of this quad
                    int co = pcol;
                    bool active = true;
                    if (p.gate_H) {
                        active = !((pcol >> 5) & 1);        // tanh tiles produce the output; sigmoid tiles are read as partners
                        co = (pcol >> 6) * 32 + (pcol & 31);
                    }
                    if (!(active && t < len && co < n_out)) continue;
                    const long long row = (long long)b * p.T + t;
                    f32x4 o = *(const f32x4*)(smem + rl * EP + c4 * 16);
                    if (p.gate_H) {
                        f32x4 g = *(const f32x4*)(smem + rl * EP + (c4 + 8) * 16);
                        o += *(const f32x4*)(p.gbias + co);
                        g += *(const f32x4*)(p.gbias + p.gate_H + co);
                        if (p.cond) {
                            long long crow = row;
                            if (p.cond_m2w) {
                                const long long w = p.cond_m2w[row];
                                crow = (w > 0 && w <= p.cond_Tw) ? (long long)b * p.cond_Tw + w : 0;
                            }
                            const float* c = p.cond + crow * p.ld_cond + p.cond_coff;
                            o += *(const f32x4*)(c + co);
                            g += *(const f32x4*)(c + p.gate_H + co);
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            // tanh(a) * sigmoid(g) = (1 - 2 / (e^{2a} + 1)) / (1 + e^{-g}) on the hardware exp2 / rcp (a few ulp; the form the fused
                            // prior flow uses, flowstack.hip): 8.5 M gates per layer at B = 60, ~45 -> ~12 instructions each.  Saturates
                            // cleanly (e^{2a} = inf -> 1, 0 -> -1)
                            const float ea = __expf(2.f * o[e]), eg = __expf(-g[e]);
                            o[e] = (1.f - 2.f * __frcp_rn(ea + 1.f)) * __frcp_rn(1.f + eg);
                        }
                    } else if (p.gbias) {
                        o += *(const f32x4*)(p.gbias + co);
                    }
                    if (p.split && co >= p.split) {
                        const int cs = co - p.split;
                        if (p.res_b) o += *(const f32x4*)(p.res_b + row * p.ldres_b + cs);
                        *(f32x4*)(p.yf2 + row * p.ldyf2 + cs) = o;
                    } else {
                        if (p.res) o += *(const f32x4*)(p.res + row * p.ldres + co);
                        *(f32x4*)(p.yf + row * p.ldyf + co) = o;
                    }
                }
            }
            return;
        }
    }
    if ((p.C_out & 3) == 0) {
        constexpr int EP = CO_T * 4 + 16;  // bytes per staged row (+16: conflict-free ds_write_b128)
        constexpr int F4 = CO_T / 4, TOTAL = WT * 32 * F4, PER = TOTAL / 256;
        const int co_blk = blockIdx.y * CO_T;
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (m) __syncthreads();
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int col = (wc * NT + n) * 32 + 8 * q + 4 * (lane >> 5);
                    const f32x4 bias = p.bias ? *(const f32x4*)(p.bias + co_blk + col) : f32x4{0.f, 0.f, 0.f, 0.f};
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[m][n][4 * q + e] + bias[e];
                    *(f32x4*)(smem + (wt * 32 + (lane & 31)) * EP + col * 4) = v;
                }
            __syncthreads();
            constexpr int UB = PER < 4 ? PER : 4;  // loads in flight per thread per batch (bounds VGPR use)
#pragma unroll
            for (int u0 = 0; u0 < PER; u0 += UB) {
                f32x4 v[UB], r1[UB];
                long long off[UB];
                bool ok[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int idx = tid + (u0 + u) * 256;
                    const int rl = idx / F4, c4 = idx % F4;
                    const int t = t0 + ((rl >> 5) * MT + m) * 32 + (rl & 31);
                    const int co = co_blk + c4 * 4;
                    ok[u] = t < len && co < p.C_out;
                    off[u] = ((long long)b * p.T + t);
                    v[u] = *(const f32x4*)(smem + rl * EP + c4 * 16);
                    r1[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (ok[u] && p.res) r1[u] = *(const f32x4*)(p.res + off[u] * p.ldres + co);
                }
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    if (!ok[u]) continue;
                    const int co = co_blk + ((tid + (u0 + u) * 256) % F4) * 4;
                    f32x4 o = v[u] + r1[u];
                    if (p.res2) o += *(const f32x4*)(p.res2 + off[u] * p.ldres2 + co);
                    if (p.div != 1.f) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = o[e] / p.div;
                    }
                    if (p.post_tanh) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = tanhf(o[e]);
                    }
                    if (p.yf) {
#if VC_NT_STORE
                        __builtin_nontemporal_store(o, (f32x4*)(p.yf + off[u] * p.ldyf + co));
#else
                        *(f32x4*)(p.yf + off[u] * p.ldyf + co) = o;
#endif
                    }
                    if (p.ya)
                        *(uint2*)(p.ya + off[u] * p.ldya + co) = make_uint2(pack2bf(lrelu(o[0], p.slope), lrelu(o[1], p.slope)),
                                                                            pack2bf(lrelu(o[2], p.slope), lrelu(o[3], p.slope)));
                }
            }
        }
        return;
    }
    // C_out not a multiple of 4 (conv_post, C_out = 1): scalar stores; with ld = 1 the 32 rows of a tile are contiguous
#pragma unroll
    for (int n = 0; n < NT; ++n) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int co = (ct0 + n) * 32 + 8 * q + 4 * (lane >> 5);
            if (co >= p.C_out) continue;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int t = t0 + (wt * MT + m) * 32 + (lane & 31);
                if (t >= len) continue;
                const long long row = (long long)b * p.T + t;
                for (int e = 0; e < 4 && co + e < p.C_out; ++e) {
                    float u = acc[m][n][4 * q + e] + (p.bias ? p.bias[co + e] : 0.f);
                    if (p.res) u += p.res[row * p.ldres + co + e];
                    if (p.res2) u += p.res2[row * p.ldres2 + co + e];
                    if (p.div != 1.f) u = u / p.div;
                    if (p.post_tanh) {   // (+ the always-on overflow detector, as in rblock.hip's fused conv_post)
                        const bool nonfin = !(__builtin_fabsf(u) <= 3.0e38f);
                        if (nonfin && p.bad) atomicAdd(p.bad, 1u);
                        u = nonfin ? __builtin_nanf("") : tanhf(u);
                    }
                    if (p.yf) p.yf[row * p.ldyf + co + e] = u;
                    if (p.ya) p.ya[row * p.ldya + co + e] = (unsigned short)vf2bf(u > 0.f ? u : u * p.slope);
                }
            }
        }
    }
}

template <int MT, int NT, int WT, int WC, int CK, bool X3, bool H2 = false>
static hipError_t vlaunch_x(const VConvParams& p, hipStream_t stream) {
    constexpr int PITCH = CK * 2 + 16, TT = 32 * MT * WT, CO_T = 32 * NT * WC;
    const int rows = TT + p.K * p.dil;   // incl. one spare tap for the activation-fragment prefetch past the last step
    size_t lds = (size_t)rows * PITCH * (X3 ? 2 : 1);
    const size_t ep = (size_t)WT * 32 * (CO_T * 4 + 16);
    if (ep > lds) lds = ep;
    auto kern = vconv_kernel<MT, NT, WT, WC, CK, X3, H2>;
    static size_t configured_dev[64] = {};   // per device: hipFuncSetAttribute is per device
    int cur_dev = 0;
    if (lds > 65536) (void)hipGetDevice(&cur_dev);
    size_t& configured = configured_dev[cur_dev & 63];
    if (lds > 65536 && lds > configured) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured = lds;
    }
    if (p.C_out_pad % CO_T || p.C_in_pad % CK) return hipErrorInvalidValue;
    if (p.gate_H && (CO_T % 64)) return hipErrorInvalidValue;   // a tanh tile and its sigmoid partner must sit in one workgroup
    // the split-operand staging addresses an utterance's fp32 rows through ONE buffer resource with 32-bit byte offsets (num_records =
    // len * ldx * 4): an utterance beyond 2 GiB (> ~131 k mel frames in the last upsampler) would wrap silently — refused instead
    if (X3 && p.xf && (long long)p.T * p.ldx * 4 >= (1LL << 31)) return hipErrorInvalidValue;
    dim3 grid((p.T + TT - 1) / TT, p.C_out_pad / CO_T, p.B);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, p);
    return hipGetLastError();
}

template <int MT, int NT, int WT, int WC, int CK>
static hipError_t vlaunch(const VConvParams& p, hipStream_t stream) {
    if (p.xf) {
        if (p.ya) return hipErrorInvalidValue;
        if (p.h2) {   // fp16 two-product form: instantiated for the wide upsampler configuration only
            if constexpr (MT == 4 && NT == 2 && CK == 128) return vlaunch_x<MT, NT, WT, WC, CK, true, true>(p, stream);
            return hipErrorInvalidValue;
        }
        if (!p.wlo) return hipErrorInvalidValue;
        return vlaunch_x<MT, NT, WT, WC, CK, true>(p, stream);
    }
    return vlaunch_x<MT, NT, WT, WC, CK, false>(p, stream);
}

hipError_t vconv_launch(const VConvParams& p, hipStream_t stream) {
    const int ci = p.C_in_pad, co = p.C_out_pad;
    if ((p.gate_H || p.split) && !p.xf) return hipErrorInvalidValue;   // the WaveNet epilogue exists on the split-operand path only
    // the prior flow's conditioning convolution (192 -> 2,048 channels over T_mel / 4 = 111-185 rows per utterance): 64-row tiles, three workgroups
    // per CU (128-row tiles: 960 workgroups for 512 slots at B = 60, 8 at B = 1): decode -24 us / -8 us (round 5)
    if (p.xf && !p.gate_H && !p.split && !p.small_tiles && co % 256 == 0 && ci == 192) return vlaunch_x<2, 2, 1, 4, 64, true>(p, stream);
    if (co % 256 == 0) {
        // ups.1 (256 -> 8 x 128 channels, two C_in chunks): 64-row tiles, three workgroups per CU instead of two 128-row ones — its workgroups
        // have half the contraction per epilogue of ups.0's, and a third resident workgroup covers more of the staging / epilogue phases:
        // 762 -> 735 us, same bits (round 5; ups.0 with four chunks loses on the same form: 183 -> 323 us, and keeps the 128-row tiles)
        if (p.xf && p.small_tiles && !p.h2 && ci == 256) return vlaunch_x<2, 2, 1, 4, 128, true>(p, stream);
        // (conv_pre — one 128-channel chunk, 720 workgroups for 512 slots — on the same form: 94.8 -> 92.5 us, within noise; left on 128-row tiles)
        if (ci % 128 == 0) return vlaunch<4, 2, 1, 4, 128>(p, stream);
        if (ci % 64 == 0) return vlaunch<4, 2, 1, 4, 64>(p, stream);
        return vlaunch<4, 2, 1, 4, 32>(p, stream);
    }
#ifdef VC_DEC_CK192   // experiment: the whole C_in = 192 of a decoder WaveNet layer as ONE chunk (one staging + barrier per tile instead of three)
    if (p.xf && (p.gate_H || p.split) && co % 128 == 0 && ci == 192 && (VC_DEC_CK192 >= 2 || p.K == 1)) return vlaunch_x<2, 1, 1, 4, 192, true>(p, stream);
#endif
    // the decoder WaveNet layers (gate / res-skip epilogue; 192 -> 384 channels, k = 5 / 1): 64-row tiles, four workgroups per CU.  On 128-row
    // tiles a B = 60 batch was 1,080 workgroups for 768 slots (two rounds for 1.4 rounds of work) and one sentence 12 workgroups:
    // decode 1.165 -> 1.117 ms at B = 60, 0.515 -> 0.446 ms at B = 1 (round 5, same contraction order: same bits)
    if (p.xf && (p.gate_H || p.split) && co % 128 == 0 && ci % 64 == 0) return vlaunch_x<2, 1, 1, 4, 64, true>(p, stream);
    if (p.xf && p.small_tiles && co % 128 == 0 && ci % 128 == 0) return vlaunch_x<2, 1, 1, 4, 128, true>(p, stream);
    if (p.xf && p.small_tiles && co % 64 == 0 && co % 128 && ci % 64 == 0) return vlaunch_x<2, 1, 2, 2, 64, true>(p, stream);
    if (co % 128 == 0) {
        if (ci % 128 == 0) return vlaunch<4, 1, 1, 4, 128>(p, stream);
        if (ci % 64 == 0) return vlaunch<4, 1, 1, 4, 64>(p, stream);
        return vlaunch<4, 1, 1, 4, 32>(p, stream);
    }
    if (co % 64 == 0) {
        // the last res_skip layer of the decoder WaveNet (192 -> 192, 1x1): 128-row tiles instead of 256 (decode -17 us / -5 us; round 5)
        if (p.xf && (p.gate_H || p.split) && ci % 64 == 0) return vlaunch_x<2, 1, 2, 2, 64, true>(p, stream);
        // the strided g_pre_net (768 -> 192 channels over T_mel / 4 rows as a 3-tap convolution over 4-frame groups): 128-row tiles and
        // 128-channel chunks — on the generic form it was ONE 256-row tile per utterance (T_mel / 4 = 111-185 rows: half of it empty) walking
        // twelve 64-channel chunks in sequence: 63 us at B = 1.  decode -25 us (B = 60), -19 us (B = 1); round 5
        if (p.xf && p.in_half && ci % 128 == 0) return vlaunch_x<2, 1, 2, 2, 128, true>(p, stream);
        if (ci % 64 == 0) return vlaunch<4, 1, 2, 2, 64>(p, stream);
        return vlaunch<4, 1, 2, 2, 32>(p, stream);
    }
    if (co % 32 == 0) {
        if (ci % 64 == 0) return vlaunch<4, 1, 4, 1, 64>(p, stream);
        return vlaunch<4, 1, 4, 1, 32>(p, stream);
    }
    return hipErrorInvalidValue;
}

// mel fp32 [rows][C] -> bf16 [rows][C_pad] (zero padded channels), no activation
__global__ void f32_to_bf16_pad_kernel(const float* x, unsigned short* y, long long rows, int C, int C_pad) {
    const long long n = rows * C_pad;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / C_pad;
        const int c = (int)(i % C_pad);
        y[i] = c < C ? (unsigned short)vf2bf(x[r * C + c]) : 0;
    }
}
hipError_t f32_to_bf16_pad_launch(const float* x, unsigned short* y, long long rows, int C, int C_pad, hipStream_t s) {
    const long long n = rows * C_pad;
    const int blocks = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(f32_to_bf16_pad_kernel, dim3(blocks), dim3(256), 0, s, x, y, rows, C, C_pad);
    return hipGetLastError();
}

} // namespace dtts
