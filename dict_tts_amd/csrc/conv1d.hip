// Implicit-GEMM 1-D convolution on gfx950 MFMA, channels-last.  See conv1d.h for the contract.
//
// GEMM view: D[t][co] = sum_{tap, ci} X[t*stride + tap*dil - pad][ci] * W[co][ci][tap]
//   A operand = activations (rows = output time), staged once per C_in chunk into LDS with the halo rows of
//               all taps, so that every tap is a row-shifted read of the same tile;
//   B operand = weights, pre-packed on the host in MFMA fragment order ([tap][k-group][co-tile][lane][16 B]) so
//               that one wave-wide 16 B/lane load is 1 KiB contiguous (L2-resident, shared by all blocks).
// Fragment maps (cdna_hip_programming.md §3): 32x32x16 bf16: A[i=l&31][k=8*(l>>5)+e], B[k=8*(l>>5)+e][j=l&31];
// 32x32x2 f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; C/D: j=l&31, i=(r&3)+8*(r>>2)+4*(l>>5).
// For fp32 each lane reads 4 consecutive channels (one ds_read_b128) and spends them on 4 successive
// 32x32x2 MFMAs; the weight packing uses the same k permutation, so the contraction is unchanged.
#include "conv1d.h"
#include "tune_env.h"
#include <type_traits>
#include <cstdlib>

namespace dtts {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ unsigned f2bf(float f) {  // round-to-nearest-even fp32 -> bf16 bits (hardware convert)
    const __bf16 h = (__bf16)f;
    return (unsigned)__builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float bf2f(unsigned h) { return __uint_as_float(h << 16); }

template <int ENGINE, int MT, int NT, int WT, int WC, int CK>
__global__ __launch_bounds__(256) void conv1d_cl_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ES = (ENGINE == ENG_F32) ? 4 : 2;
    constexpr int KG = (ENGINE == ENG_F32) ? 8 : 16;  // channels per k-group (one 16-B fragment per lane)
    constexpr int PITCH = CK * ES + 16;               // +16 B: conflict-free ds_read_b128 across 16 rows
    constexpr int TT = 32 * MT * WT;
    constexpr int CO_T = 32 * NT * WC;
    constexpr int NKG = CK / KG;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wt = wave % WT, wc = wave / WT;
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * TT;
    const int ct0 = blockIdx.y * (CO_T / 32) + wc * NT;  // first packed co-tile of this wave
    const int NCT = p.C_out_pad >> 5;
    const int in_len = p.in_lens ? p.in_lens[b] : p.T_in;
    const int out_len = p.out_lens ? p.out_lens[b] : p.T_out;
    if (t0 >= out_len && !p.zero_masked) return;
    const int rows = (TT - 1) * p.stride + (p.K - 1) * p.dil + 1;
    const int in0 = t0 * p.stride - p.pad;
    const int NG = p.C_in_pad / KG;
    constexpr int pitch = PITCH;
    char* lds_lo = smem + (size_t)rows * pitch;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const float* xb = p.x + (long long)b * p.x_bstride + p.x_coff;
    const bool live = t0 < out_len;

    // ---- stage X[in0 .. in0+rows) x [ci0, ci0 + width) into LDS columns [col0, col0 + width) (pre-activation, zero padding, conversion):
    // batches of U independent 16 B loads in flight per thread, then conversion + LDS writes
    auto stage = [&](int ci0, int width, int col0) {
        constexpr int U = 4;
        constexpr int PIECES = CK / 4;
        const int total = rows * PIECES;
        for (int base = tid; base < total; base += 256 * U) {
            f32x4 vv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + u * 256;
                const int r = idx / PIECES, c4 = idx % PIECES;
                const int t = in0 + r, ci = ci0 + c4 * 4;
                vv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (idx < total && t >= 0 && t < in_len && ci < p.C_in) vv[u] = *(const f32x4*)(xb + (long long)t * p.ldx + ci);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + u * 256;
                if (idx >= total) continue;
                const int r = idx / PIECES, c4 = idx % PIECES + col0 / 4;
                f32x4 v = vv[u];
                if (p.pre_act) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.pre_slope;
                }
                if constexpr (ENGINE == ENG_F32) {
                    *(f32x4*)(smem + r * pitch + c4 * 16) = v;
                } else {
                    unsigned h0 = f2bf(v[0]), h1 = f2bf(v[1]), h2 = f2bf(v[2]), h3 = f2bf(v[3]);
                    *(uint2*)(smem + r * pitch + c4 * 8) = make_uint2(h0 | (h1 << 16), h2 | (h3 << 16));
                    if constexpr (ENGINE == ENG_BF16X3) {
                        unsigned l0 = f2bf(v[0] - bf2f(h0)), l1 = f2bf(v[1] - bf2f(h1));
                        unsigned l2 = f2bf(v[2] - bf2f(h2)), l3 = f2bf(v[3] - bf2f(h3));
                        *(uint2*)(lds_lo + r * pitch + c4 * 8) = make_uint2(l0 | (l1 << 16), l2 | (l3 << 16));
                    }
                }
            }
        }
    };
    if (live) {
        for (int ci0 = 0; ci0 < p.C_in_pad; ci0 += CK) {
            stage(ci0, CK, 0);
            __syncthreads();
            // ---- contraction over taps and k-groups of this chunk.  Steps s = tap * NKG + kg; weight fragments
            // run PF steps ahead in a register ring, activation fragments one step ahead (double buffer); the
            // sched_barriers keep those prefetches above the MFMAs of the current step.
            {
                constexpr int R = NKG < 4 ? NKG : 4, PF = R - 1;
                const int g0 = ci0 / KG;
                const int S = p.K * NKG;
                const size_t tap_stride = (size_t)NG * NCT * 64, kg_stride = (size_t)NCT * 64;
                const uint4* wh = (const uint4*)p.w_hi + (size_t)g0 * NCT * 64 + lane;
                const uint4* wl = (const uint4*)p.w_lo + (size_t)g0 * NCT * 64 + lane;
                int ctc[NT];
#pragma unroll
                for (int n = 0; n < NT; ++n) ctc[n] = ct0 + n < NCT ? ct0 + n : NCT - 1;
                // 32-bit element offsets (a packed layer is far below 2^31 uint4): one s_mul / s_add per prefetch
                const int tap_stride_i = (int)tap_stride, kg_stride_i = (int)kg_stride;
                auto load_w = [&](uint4 (&dh)[NT], uint4 (&dl)[NT], int s) {
                    const int sc = s < S ? s : S - 1;
                    const int off = (sc / NKG) * tap_stride_i + (sc % NKG) * kg_stride_i;
#pragma unroll
                    for (int n = 0; n < NT; ++n) {
                        dh[n] = wh[off + ctc[n] * 64];
                        if constexpr (ENGINE == ENG_BF16X3) dl[n] = wl[off + ctc[n] * 64];
                    }
                };
                const int abase = ((wt * MT) * 32 + (lane & 31)) * p.stride * pitch + (lane >> 5) * (KG / 2) * ES;
                auto load_x = [&](uint4 (&dh)[MT], uint4 (&dl)[MT], int s) {
                    const int sc = s < S ? s : S - 1;
                    const int off = abase + (sc / NKG) * p.dil * pitch + (sc % NKG) * KG * ES;
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        dh[m] = *(const uint4*)(smem + off + m * 32 * p.stride * pitch);
                        if constexpr (ENGINE == ENG_BF16X3) dl[m] = *(const uint4*)(lds_lo + off + m * 32 * p.stride * pitch);
                    }
                };
                uint4 rh[R][NT], rl[R][NT], xh[2][MT], xl[2][MT];
#pragma unroll
                for (int s = 0; s < PF; ++s) load_w(rh[s], rl[s], s);
                load_x(xh[0], xl[0], 0);
                for (int tap = 0; tap < p.K; ++tap) {
#pragma unroll
                    for (int kg = 0; kg < NKG; ++kg) {
                        const int s = tap * NKG + kg;
                        load_w(rh[(kg + PF) % R], rl[(kg + PF) % R], s + PF);
                        load_x(xh[(kg + 1) & 1], xl[(kg + 1) & 1], s + 1);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int m = 0; m < MT; ++m)
#pragma unroll
                            for (int n = 0; n < NT; ++n) {
                                // (a co-tile past the layer's last one computes on a clamped copy; the epilogue drops it)
                                const uint4 bh = rh[kg % R][n], bl = rl[kg % R][n];
                                if constexpr (ENGINE == ENG_F32) {
                                    const f32x4 a = *(const f32x4*)&xh[kg & 1][m];
                                    const f32x4 w = *(const f32x4*)&bh;
#pragma unroll
                                    for (int q = 0; q < 4; ++q)
                                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], w[q], acc[m][n], 0, 0, 0);
                                } else {
                                    const bf16x8 a = *(const bf16x8*)&xh[kg & 1][m];
                                    const bf16x8 w = *(const bf16x8*)&bh;
                                    if constexpr (ENGINE == ENG_BF16X3) {
                                        const bf16x8 a2 = *(const bf16x8*)&xl[kg & 1][m];
                                        const bf16x8 w2 = *(const bf16x8*)&bl;
                                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, w, acc[m][n], 0, 0, 0);
                                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, w2, acc[m][n], 0, 0, 0);
                                    }
                                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, w, acc[m][n], 0, 0, 0);
                                }
                            }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            __syncthreads();
        }
    }

    // ---- epilogue
    const int col = lane & 31;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            if (p.gate_H && (n & 1)) continue;
            const int ct = ct0 + n;
            if (ct >= NCT) continue;
            int co, co_s = 0;
            if (p.gate_H) {
                co = (ct >> 1) * 32 + col;   // tanh channel; sigmoid channel = gate_H + co
                co_s = p.gate_H + co;
            } else {
                co = ct * 32 + col;
            }
            if (co >= (p.gate_H ? p.gate_H : p.C_out)) continue;
            const float bias = p.bias ? p.bias[co] : 0.f;
            const float bias_s = (p.gate_H && p.bias) ? p.bias[co_s] : 0.f;
            const int s = (co >= p.split) ? 1 : 0;
            const ConvSeg& sg = p.seg[s];
            const int cs = co - (s ? p.split : 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int t = t0 + (wt * MT + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (t >= p.T_out) continue;
                const long long row = (long long)b * p.y_bstride_rows + t;
                float v;
                if (t >= out_len) {
                    if (!p.zero_masked) continue;
                    v = 0.f;
                } else {
                    v = acc[m][n][r] + bias;
                    if (p.gate_H) {
                        float u = acc[m][n + (NT > 1 ? 1 : 0)][r] + bias_s;
                        if (p.cond) {
                            const float* c = p.cond + row * p.ld_cond + p.cond_coff;
                            v += c[co];
                            u += c[co_s];
                        }
                        v = tanhf(v) * (1.f / (1.f + expf(-u)));
                    }
                    if (sg.res) v += sg.res[row * sg.ld_res + sg.coff_res + cs];
                    if (sg.res2) v += sg.res2[row * sg.ld_res2 + sg.coff_res2 + cs];
                    if (p.out_div != 1.f) v = v / p.out_div;
                    if (p.out_mul != 1.f) v = v * p.out_mul;
                    if (p.post_act == 1) v = fmaxf(v, 0.f);
                    else if (p.post_act == 2) v = tanhf(v);
                    else if (p.post_act == 3) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));  // F.gelu (erf form)
                }
                sg.y[row * sg.ld + sg.coff + cs] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// conv1d_short_kernel: the same convolution for FEW ROWS (<= 256 tiles of 32 output rows over the whole batch: the word encoders at
// T_w ~ 27, the duration predictor, B = 1, one long text).  There the generic kernel is bound by per-kernel round trips and by one
// wave's serial MFMA chain, not by the matrix rate (tools/c1d_phase_prof.py; LABNOTES (qq)-(ss)):
//   * one 32-row tile per workgroup, staged WHOLE (all C_in) in one batch of loads; (tap, k-group) is one step sequence per wave;
//   * KS = 2 / 4: the workgroup's waves each sum 1/KS of the k-groups of every tap, partial sums meet in LDS in a fixed order;
//   * ENG_BF16X6: fp32-grade products from three bf16 pieces per operand (6 MFMAs per 16 channels instead of 8 fp32 ones per 8);
//   * the epilogue's operands are requested above the contraction; every optional epilogue step runs over all 16 rows of the lane
//     under one uniform branch.
// C1D_PROF = 1 (make prof): per-workgroup cycle stamps at start / tile staged / contraction done / stores issued.
#ifndef C1D_PROF
#define C1D_PROF 0
#endif
#if C1D_PROF
__device__ unsigned long long c1d_prof[4][2048][8];
extern "C" __attribute__((visibility("default"))) int dtts_debug_c1d_prof(unsigned long long* host) { return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(c1d_prof), sizeof(c1d_prof)); }
#define STAMP(i) if (pslot >= 0 && threadIdx.x == 0) { c1d_prof[pslot][pwg][i] = __builtin_readcyclecounter(); }
#else
#define STAMP(i)
#endif
// PP: contraction PARTS per wave.  The contraction of an output tile is always summed as P = KS * PP partial sums — part j = the k-groups
// [j NG / P, (j + 1) NG / P) of every tap, combined ((p0 + p1) + p2) + p3 — where P depends on the LAYER only (4 if its k-group count
// divides by 4, else 2, else 1): how many waves share the parts (KS = 4 / 2 / 1, chosen per launch from the grid size and the CU count)
// does not change a single bit of the result, so an utterance gets the same log-durations alone and inside any batch.
template <int ENGINE, int NT, int WC, int KS, int U, int PP>
__global__ __launch_bounds__(256, U > 8 ? 1 : NT > 1 ? (ENGINE == ENG_BF16X6 ? 1 : 2) : 3) void conv1d_short_kernel(const ConvParams p) {
    constexpr int MT = 1, WT = 1;   // one 32-row time tile per workgroup
    static_assert(WC * KS == 4, "four waves: co-tile groups x contraction splits");
    extern __shared__ __attribute__((aligned(16))) char smem[];
#if C1D_PROF
    const int pslot = (p.K == 5 && p.C_out == 768) ? 0 : (p.K == 1 && p.C_out == 576) ? 1 : (p.K == 1 && p.C_in == 768 && p.T_out <= 64) ? 2 : (p.K == 1 && p.C_out == 192 && p.C_in == 192) ? 3 : -1;
    const int pwg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (pslot >= 0 && threadIdx.x == 0) { c1d_prof[pslot][pwg][6] = __builtin_amdgcn_s_memrealtime(); c1d_prof[pslot][pwg][5] = 0; c1d_prof[pslot][pwg][3] = 0; }
    STAMP(0)
#endif
    constexpr int ES = (ENGINE == ENG_F32) ? 4 : 2;
    constexpr int KG = (ENGINE == ENG_F32) ? 8 : 16;  // channels per k-group (one 16-B fragment per lane)
    // operand planes: fp32 values as they are, or bf16 pieces x = p0 + p1 (+ p2), p_i = bf16(x - p_0 - .. - p_{i-1}) (exact remainders)
    constexpr int NP = ENGINE == ENG_BF16X6 ? 3 : ENGINE == ENG_BF16X3 ? 2 : 1;
    constexpr int TT = 32 * MT * WT;
    constexpr int CO_T = 32 * NT * WC;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wt = wave % WT, wc = (wave / WT) % WC;
    const int ks = __builtin_amdgcn_readfirstlane(wave / (WT * WC));   // which part of the contraction this wave sums (KS > 1)
    const int b = blockIdx.z;
    const int t0 = blockIdx.x * TT;
    const int ct0 = blockIdx.y * (CO_T / 32) + wc * NT;  // first packed co-tile of this wave
    const int NCT = p.C_out_pad >> 5;
    const int in_len = p.in_lens ? p.in_lens[b] : p.T_in;
    const int out_len = p.out_lens ? p.out_lens[b] : p.T_out;
    const int rows = (TT - 1) * p.stride + (p.K - 1) * p.dil + 1;
    const int in0 = t0 * p.stride - p.pad;
    const int NG = p.C_in_pad / KG;
    const int pitch = p.C_in_pad * ES + 16;   // +16 B: conflict-free ds_read_b128 across 16 rows ((C_in_pad * ES + 16) mod 256 == 16 for the widths in use)
    const int plane = rows * pitch;   // bytes between the planes of the staged tile

    // accumulator sets: one per part when the parts of other waves arrive through LDS (KS > 1); a single wave (KS = 1) needs only two —
    // the running total ((p0 + p1) + ..) and the part in progress
    constexpr int NSET = KS == 1 ? (PP > 1 ? 2 : 1) : PP;
    f32x16 accp[NSET][MT][NT];
#pragma unroll
    for (int pp = 0; pp < NSET; ++pp)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) accp[pp][m][n][r] = 0.f;
    f32x16 (&acc)[MT][NT] = accp[0];   // the combined sum ends up in part 0's registers

    const float* xb = p.x + (long long)b * p.x_bstride + p.x_coff;
    const bool live = t0 < out_len;
#if C1D_PROF
    if (pslot >= 0 && threadIdx.x == 0) c1d_prof[pslot][pwg][5] = 1 + live;
#endif

    // ---- stage X[in0 .. in0 + rows) x [0, C_in_pad) into LDS (pre-activation, zero padding, conversion / split into planes): U independent
    // 16 B loads in flight per thread (the launcher picks U so that the whole tile is ONE batch where the registers allow), then the LDS
    // writes.  The rows are requested before the utterance's length is known (up to the padded T_in, masked afterwards): kernel arguments
    // -> lengths -> rows would be one more round trip with nothing to hide it.
    {
        const int PIECES = p.C_in_pad / 4;
        const int total = rows * PIECES;
        // piece idx = r * PIECES + c4 walked incrementally (256 pieces per hop): one division per thread, not two per piece
        // (one wave per SIMD in the short-sequence configurations: every instruction of this loop is on the critical path)
        const int dr = 256 / PIECES, dc = 256 % PIECES;
        int r = tid / PIECES, c4 = tid - r * PIECES;
        for (int base = tid; base < total; base += 256 * U) {
            f32x4 vv[U];
            int ru[U], cu[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                ru[u] = r;
                cu[u] = c4;
                const int t = in0 + r, ci = c4 * 4;
                vv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (r < rows && t >= 0 && t < p.T_in && ci < p.C_in) vv[u] = *(const f32x4*)(xb + (long long)t * p.ldx + ci);
                r += dr;
                c4 += dc;
                if (c4 >= PIECES) {
                    c4 -= PIECES;
                    ++r;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (ru[u] >= rows) continue;
                const int r = ru[u], c4 = cu[u];
                f32x4 v = vv[u];
                if (in0 + r >= in_len) v = f32x4{0.f, 0.f, 0.f, 0.f};
                if (p.pre_act) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.pre_slope;
                }
                if constexpr (ENGINE == ENG_F32) {
                    *(f32x4*)(smem + r * pitch + c4 * 16) = v;
                } else {
#pragma unroll
                    for (int pl = 0; pl < NP; ++pl) {
                        const unsigned h0 = f2bf(v[0]), h1 = f2bf(v[1]), h2 = f2bf(v[2]), h3 = f2bf(v[3]);
                        *(uint2*)(smem + pl * plane + r * pitch + c4 * 8) = make_uint2(h0 | (h1 << 16), h2 | (h3 << 16));
                        v = f32x4{v[0] - bf2f(h0), v[1] - bf2f(h1), v[2] - bf2f(h2), v[3] - bf2f(h3)};
                    }
                }
            }
        }
    }
    // ---- the epilogue's operands (bias, residual rows) of output tile (m, n): requested ABOVE the contraction, all 16 rows in flight
    // at once (one wave per SIMD: nothing else would hide the round trip in the epilogue)
    struct EpiOperands {
        bool valid;
        int co;
        float bias, bias_s;
        float r1[16];   // (the second residual is rare on this kernel's layers: fetched in the epilogue, its 16 registers are worth a wave of occupancy)
    };
    auto epi_fetch = [&](int m, int n, EpiOperands& eo) {
        eo.valid = false;
        if (KS > 1 && ks > 0) return;
        if (p.gate_H && (n & 1)) return;
        const int ct = ct0 + n;
        if (ct >= NCT) return;
        const int co = (p.gate_H ? (ct >> 1) : ct) * 32 + (lane & 31);   // gated: tanh channel; its sigmoid channel = gate_H + co
        if (co >= (p.gate_H ? p.gate_H : p.C_out)) return;
        eo.valid = true;
        eo.co = co;
        eo.bias = p.bias ? p.bias[co] : 0.f;
        eo.bias_s = (p.gate_H && p.bias) ? p.bias[p.gate_H + co] : 0.f;
        const int s = (co >= p.split) ? 1 : 0;
        const ConvSeg& sg = p.seg[s];
        const int cs = co - (s ? p.split : 0);
        const int tb = t0 + (wt * MT + m) * 32 + 4 * (lane >> 5);
        const long long row0 = (long long)b * p.y_bstride_rows;
#pragma unroll
        for (int r = 0; r < 16; ++r) eo.r1[r] = 0.f;
        if (sg.res) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int t = tb + (r & 3) + 8 * (r >> 2);
                if (t < p.T_out && t < out_len) eo.r1[r] = sg.res[(row0 + t) * sg.ld_res + sg.coff_res + cs];
            }
        }
    };
    EpiOperands early[MT][NT];
    __syncthreads();
    // a tile beyond its utterance leaves after the staging
    if (t0 >= out_len && !p.zero_masked) return;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) epi_fetch(m, n, early[m][n]);
    STAMP(1)
    // one step's MFMAs: weight fragments bw[plane][n] x activation fragments aw[plane][m] into every (m, n) accumulator.  Split
    // operands: every product of pieces i, j with i + j < NP, smallest first (x3: 3 products ~ 2^-16; x6: 6 products ~ 2^-24, the
    // fp32 MFMA's accuracy at 6 x 8 passes per 16 channels instead of 8 x 16)
    auto mma = [&](f32x16 (&acc)[MT][NT], const uint4 (&bw)[NP][NT], const uint4 (&aw)[NP][MT]) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                // (a co-tile past the layer's last one computes on a clamped copy; the epilogue drops it)
                if constexpr (ENGINE == ENG_F32) {
                    const f32x4 a = *(const f32x4*)&aw[0][m];
                    const f32x4 w = *(const f32x4*)&bw[0][n];
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], w[q], acc[m][n], 0, 0, 0);
                } else {
#pragma unroll
                    for (int d = NP - 1; d >= 0; --d)
#pragma unroll
                        for (int i = d; i >= 0; --i) {
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)&aw[i][m], *(const bf16x8*)&bw[d - i][n], acc[m][n], 0, 0, 0);
                        }
                }
            }
    };
    // ---- short-sequence contraction: the tile is staged whole, so the steps (tap, k-group) run as ONE sequence per wave (the
    // weight ring is never restarted), and the KS waves of an output tile each sum 1/KS of the k-groups of every tap — the
    // fp32 MFMA chain of one wave (64 cycles per 32x32x2) is what bounds these kernels, not the matrix rate.  Partial sums
    // meet in LDS in a fixed order (ks = 0 + 1 + 2 + 3).
    if (live) {
        constexpr int R = 4, PF = 3;
        const int NGW = NG / (KS * PP);               // k-groups per tap and PART (the launcher checks NG % (KS * PP) == 0)
        const int S = p.K * NGW;
        const int kg_stride_i = NCT * 64, tap_wrap_w = (NG - NGW) * NCT * 64;
        int ctc[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) ctc[n] = ct0 + n < NCT ? ct0 + n : NCT - 1;
        constexpr bool SEQ = KS == 1 && PP > 1;       // one wave sums all parts: each finished part is folded into the total at once
        // (SEQ: a real loop — four unrolled copies of the contraction would cost the registers the two sets save; every part
        // accumulates in set 1 and is added to the total in set 0, 0 + p0 being exact)
#pragma unroll(SEQ ? 1 : PP)
        for (int pp = 0; pp < PP; ++pp) {
        const int part = ks * PP + pp;                // this wave's parts: ks * PP .. ks * PP + PP - 1
        constexpr int SEQ_SET = 1;
        if (SEQ && pp > 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) accp[1][m][n][r] = 0.f;
        }
        const uint4* wp[3] = {(const uint4*)p.w_hi + (size_t)part * NGW * NCT * 64 + lane, (const uint4*)p.w_lo + (size_t)part * NGW * NCT * 64 + lane,
                              (const uint4*)p.w_lo2 + (size_t)part * NGW * NCT * 64 + lane};
        // two cursors over the part's step sequence (weights run PF steps ahead, activations one), advanced by adds; both stop on
        // the last step
        int wo = 0, wg = 0, ws = 0;
        auto load_w = [&](uint4 (&d)[NP][NT]) {
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                for (int n = 0; n < NT; ++n) d[pl][n] = wp[pl][wo + ctc[n] * 64];
            if (ws + 1 < S) {
                ++ws;
                wo += kg_stride_i;
                if (++wg == NGW) {
                    wg = 0;
                    wo += tap_wrap_w;
                }
            }
        };
        const int abase = ((wt * MT) * 32 + (lane & 31)) * p.stride * pitch + (lane >> 5) * (KG / 2) * ES + part * NGW * KG * ES;
        const int tap_wrap_x = p.dil * pitch - NGW * KG * ES;
        int xo = 0, xg = 0, xs = 0;
        auto load_x = [&](uint4 (&d)[NP][MT]) {
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                for (int m = 0; m < MT; ++m) d[pl][m] = *(const uint4*)(smem + pl * plane + abase + xo + m * 32 * p.stride * pitch);
            if (xs + 1 < S) {
                ++xs;
                xo += KG * ES;
                if (++xg == NGW) {
                    xg = 0;
                    xo += tap_wrap_x;
                }
            }
        };
        uint4 rw[R][NP][NT], xa[2][NP][MT];
#pragma unroll
        for (int j = 0; j < PF; ++j) load_w(rw[j]);
        load_x(xa[0]);
        for (int s = 0; s < S; s += R) {
#pragma unroll
            for (int j = 0; j < R; ++j) {
                load_w(rw[(j + PF) % R]);
                load_x(xa[(j + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                if (s + j < S) {
                    if constexpr (SEQ) mma(accp[SEQ_SET], rw[j], xa[j & 1]);
                    else mma(accp[pp], rw[j], xa[j & 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (SEQ) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) accp[0][m][n][r] += accp[1][m][n][r];
        }
        }   // (parts of this wave)
        // ---- combine the P = KS * PP partial sums in the fixed order ((p0 + p1) + p2) + p3: this wave's own parts first ...
        if (KS > 1 && ks == 0) {
#pragma unroll
            for (int pp = 1; pp < PP; ++pp)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int r = 0; r < 16; ++r) accp[0][m][n][r] += accp[pp][m][n][r];
        }
        if constexpr (KS > 1) {   // ... then the other waves' parts, one by one, through LDS
            __syncthreads();   // every wave is done with the staged rows: their LDS takes the partial sums
            float* red = (float*)smem;
            const int tile = wave % (WT * WC);
            if (ks > 0) {
#pragma unroll
                for (int pp = 0; pp < PP; ++pp)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int n = 0; n < NT; ++n)
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                red[((((((ks - 1) * PP + pp) * (WT * WC) + tile) * MT + m) * NT + n) * 16 + r) * 64 + lane] = accp[pp][m][n][r];
            }
            __syncthreads();
            if (ks == 0) {
#pragma unroll
                for (int k2 = 1; k2 < KS; ++k2)
#pragma unroll
                    for (int pp = 0; pp < PP; ++pp)
#pragma unroll
                        for (int m = 0; m < MT; ++m)
#pragma unroll
                            for (int n = 0; n < NT; ++n)
#pragma unroll
                                for (int r = 0; r < 16; ++r)
                                    acc[m][n][r] += red[((((((k2 - 1) * PP + pp) * (WT * WC) + tile) * MT + m) * NT + n) * 16 + r) * 64 + lane];
            }
        }
    }
    if (KS > 1 && ks > 0) return;

    // ---- epilogue
    STAMP(2)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const EpiOperands& eo = early[m][n];
            if (!eo.valid) continue;
            const int co = eo.co, co_s = p.gate_H + co;
            const int s = (co >= p.split) ? 1 : 0;
            const ConvSeg& sg = p.seg[s];
            const int cs = co - (s ? p.split : 0);
            const float bias = eo.bias, bias_s = eo.bias_s;
            // each optional step applied to all 16 rows under ONE uniform branch (straight-line bodies; masked rows compute on
            // whatever the accumulator holds and are replaced by zero or dropped at the store)
            const int tb = t0 + (wt * MT + m) * 32 + 4 * (lane >> 5);
            const long long row0 = (long long)b * p.y_bstride_rows + tb;
            const int t_end = out_len < p.T_out ? out_len : p.T_out;     // rows below it are computed, rows in [t_end, T_out) zeroed or kept
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[m][n][r] + bias;
            if (p.gate_H) {
                float u[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) u[r] = acc[m][n + (NT > 1 ? 1 : 0)][r] + bias_s;
                if (p.cond) {
                    const float* c = p.cond + row0 * p.ld_cond + p.cond_coff;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dt = (r & 3) + 8 * (r >> 2);
                        if (tb + dt < t_end) {
                            v[r] += c[(long long)dt * p.ld_cond + co];
                            u[r] += c[(long long)dt * p.ld_cond + co_s];
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = tanhf(v[r]) * (1.f / (1.f + expf(-u[r])));
            }
            if (sg.res) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] += eo.r1[r];
            }
            if (sg.res2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dt = (r & 3) + 8 * (r >> 2);
                    if (tb + dt < t_end) v[r] += sg.res2[(row0 + dt) * sg.ld_res2 + sg.coff_res2 + cs];
                }
            }
            if (p.out_div != 1.f) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = v[r] / p.out_div;
            }
            if (p.out_mul != 1.f) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = v[r] * p.out_mul;
            }
            if (p.post_act == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
            } else if (p.post_act == 2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = tanhf(v[r]);
            } else if (p.post_act == 3) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = 0.5f * v[r] * (1.f + erff(v[r] * 0.70710678118654752f));  // F.gelu (erf form)
            }
            float* yp = sg.y + row0 * sg.ld + sg.coff + cs;
            const int t_store = p.zero_masked ? p.T_out : t_end;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dt = (r & 3) + 8 * (r >> 2);
                if (tb + dt < t_store) yp[(long long)dt * sg.ld] = tb + dt < t_end ? v[r] : 0.f;
            }
        }
    }
    STAMP(3)
#if C1D_PROF
    if (pslot >= 0 && threadIdx.x == 0) c1d_prof[pslot][pwg][7] = __builtin_amdgcn_s_memrealtime();
#endif
}

template <int ENGINE, int MT, int NT, int WT, int WC, int CK>
static hipError_t launch_cfg(const ConvParams& p, hipStream_t stream) {
    constexpr int ES = (ENGINE == ENG_F32) ? 4 : 2;
    constexpr int PITCH = CK * ES + 16;
    constexpr int TT = 32 * MT * WT, CO_T = 32 * NT * WC;
    const int rows = (TT - 1) * p.stride + (p.K - 1) * p.dil + 1;
    const size_t lds = (size_t)rows * PITCH * (ENGINE == ENG_BF16X3 ? 2 : 1);
    auto kern = conv1d_cl_kernel<ENGINE, MT, NT, WT, WC, CK>;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {   // above the default dynamic-LDS limit (the strided g_pre_net on this kernel: ~140 KB): opt in, per device
        static bool configured_dev[64] = {};
        int cur_dev = 0;
        (void)hipGetDevice(&cur_dev);
        if (!configured_dev[cur_dev & 63]) {
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            configured_dev[cur_dev & 63] = true;
        }
    }
    dim3 grid((p.T_out + TT - 1) / TT, (p.C_out_pad + CO_T - 1) / CO_T, p.B);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, p);
    return hipGetLastError();
}

// LDS of one short-sequence tile: 32 output rows' receptive field x the whole input width (+ the partial sums of a split contraction)
static size_t short_lds(const ConvParams& p, int es, int planes, int red_tiles) {
    const size_t rows = (size_t)31 * p.stride + (size_t)(p.K - 1) * p.dil + 1;
    const size_t tile = rows * ((size_t)p.C_in_pad * es + 16) * planes, red = (size_t)red_tiles * 16 * 64 * sizeof(float);
    return tile > red ? tile : red;
}

static int cu_count() {
    static const int n_cu = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    return n_cu;
}

template <int ENGINE, int NT, int WC, int KS, int U, int PP>
static hipError_t launch_short_u(const ConvParams& p, hipStream_t stream) {
    constexpr int ES = (ENGINE == ENG_F32) ? 4 : 2;
    constexpr size_t LDS_CU = 160 * 1024;
    const size_t lds = short_lds(p, ES, ENGINE == ENG_BF16X6 ? 3 : ENGINE == ENG_BF16X3 ? 2 : 1, (KS - 1) * PP * WC * NT);
    auto kern = conv1d_short_kernel<ENGINE, NT, WC, KS, U, PP>;
    static bool configured_dev[64] = {};   // per device: hipFuncSetAttribute is per device
    int cur_dev = 0;
    (void)hipGetDevice(&cur_dev);
    if (!configured_dev[cur_dev & 63]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_CU);
        if (e != hipSuccess) return e;
        configured_dev[cur_dev & 63] = true;
    }
    dim3 grid((p.T_out + 31) / 32, (p.C_out_pad + 32 * NT * WC - 1) / (32 * NT * WC), p.B);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, p);
    return hipGetLastError();
}

template <int ENGINE, int NT, int WC, int KS, int PP = 1>
static hipError_t launch_short(const ConvParams& p, hipStream_t stream) {
    // 16 B pieces of the staged tile per thread: 8 in flight at 4 workgroups per CU (<= 128 VGPRs), or 24 for the wide tiles (LDS
    // leaves one workgroup per CU anyway)
    const size_t pieces = ((size_t)31 * p.stride + (size_t)(p.K - 1) * p.dil + 1) * (p.C_in_pad / 4);
    return pieces <= 256 * 8 ? launch_short_u<ENGINE, NT, WC, KS, 8, PP>(p, stream) : launch_short_u<ENGINE, NT, WC, KS, 24, PP>(p, stream);
}

// Few rows (the T_w ~ 27 encoder at B = 60, one long text at B = 1: <= 256 tiles of 32 rows): bound by the serial MFMA chain of one wave
// and by round trips nothing hides -> conv1d_short_kernel while the 32-row tile fits the LDS whole.  (Gated layers keep two co-tiles per wave: the tanh tile and its
// sigmoid partner meet in the epilogue.)  Returns false when the layer does not qualify.
template <int ENGINE>
static bool launch_short_policy(const ConvParams& p, hipStream_t stream, hipError_t* err) {
    constexpr int ES = (ENGINE == ENG_F32) ? 4 : 2, KG = (ENGINE == ENG_F32) ? 8 : 16, NP = ENGINE == ENG_BF16X6 ? 3 : ENGINE == ENG_BF16X3 ? 2 : 1;
    const size_t lds = short_lds(p, ES, NP, 0);
    // the three-piece engine IS the layer's arithmetic (fixed when the weights were packed): it takes every problem size whose 32-row
    // tile fits the LDS; the exact-fp32 MFMA engine keeps the few-rows limit (beyond it the generic kernel computes the same products)
    if (lds > 150 * 1024 || (ENGINE != ENG_BF16X6 && (long long)p.B * ((p.T_out + 31) / 32) > 256)) return false;
    if (p.gate_H) {
        *err = launch_short<ENGINE, 2, 4, 1>(p, stream);                           // 32 t x 256 co
        return true;
    }
    // The contraction is summed in P parts fixed by the LAYER (see the kernel); the parts are spread over 4 / 2 / 1 waves (each output
    // tile's chain 4x / 2x shorter, 4x / 2x the workgroups) as far as ALL workgroups stay co-resident: a second round of workgroups
    // costs more than the shorter chains give.  That choice depends on the grid and the device — the result does not.
    static const int ks_env = [] { const char* e = ablate_env("DTTS_C1D_KS"); return e ? atoi(e) : 0; }();   // A/B override
    const int n_cu = cu_count();
    const int NG = p.C_in_pad / KG;
    const int P = NG % 4 == 0 ? 4 : (NG % 2 == 0 ? 2 : 1);
    const size_t per_cu_lds = (160 * 1024) / lds;
    const bool wide = ((size_t)31 * p.stride + (size_t)(p.K - 1) * p.dil + 1) * (p.C_in_pad / 4) > 256 * 8;
    const size_t tiles = (size_t)((p.T_out + 31) / 32) * p.B;
    int ks = 1;
    for (int k = P; k >= 2 && ks == 1; k >>= 1) {
        // waves per SIMD the kernel's registers allow (k < P: the wave carries P / k accumulator sets)
        const size_t occ = wide ? 2 : (k == P ? (k == 4 ? 3 : (ENGINE == ENG_F32 ? 4 : 3)) : (ENGINE == ENG_F32 ? 3 : 2));
        const size_t wgs = tiles * ((p.C_out_pad + 32 * (4 / k) - 1) / (32 * (4 / k)));
        if (wgs <= (size_t)n_cu * (occ < per_cu_lds ? occ : per_cu_lds)) ks = k;
    }
    if ((ks_env == 1 || ks_env == 2 || ks_env == 4) && P % ks_env == 0) ks = ks_env;
    if (P == 4) {
        if (ks == 4) *err = launch_short<ENGINE, 1, 1, 4, 1>(p, stream);         // 32 t x 32 co, one part per wave
        else if (ks == 2) *err = launch_short<ENGINE, 1, 2, 2, 2>(p, stream);    // 32 t x 64 co, two parts per wave
        else *err = launch_short<ENGINE, 1, 4, 1, 4>(p, stream);                 // 32 t x 128 co, all four parts in one wave
    } else if (P == 2) {
        if (ks == 2) *err = launch_short<ENGINE, 1, 2, 2, 1>(p, stream);
        else *err = launch_short<ENGINE, 1, 4, 1, 2>(p, stream);
    } else {
        *err = launch_short<ENGINE, 1, 4, 1, 1>(p, stream);
    }
    return true;
}

template <int ENGINE, int CK>
static hipError_t launch_engine(const PackedConv& L, const ConvParams& p, hipStream_t stream) {
    if constexpr (ENGINE == ENG_F32) {
        hipError_t err = hipSuccess;
        if (L.x6[0]) {   // the same weights as three bf16 pieces: fp32-grade products at 2.7x the fp32 MFMA rate
            ConvParams q = p;
            q.w_hi = L.x6[0];
            q.w_lo = L.x6[1];
            q.w_lo2 = L.x6[2];
            if (launch_short_policy<ENG_BF16X6>(q, stream, &err)) return err;
        }
        if (launch_short_policy<ENG_F32>(p, stream, &err)) return err;
        if (p.T_out <= 64 && !p.gate_H) return launch_cfg<ENGINE, 1, 1, 1, 4, CK>(p, stream);   // 32 t x 128 co
        if (p.T_out <= 64) return launch_cfg<ENGINE, 1, 2, 1, 4, CK>(p, stream);                // 32 t x 256 co
        return launch_cfg<ENGINE, 1, 2, 4, 1, CK>(p, stream);                                    // 128 t x 64 co
    } else {
        if (L.C_out_pad <= 32) return launch_cfg<ENGINE, 2, 1, 4, 1, CK>(p, stream);   // 256 t x 32 co
        if (L.C_out_pad <= 64) return launch_cfg<ENGINE, 2, 2, 4, 1, CK>(p, stream);   // 256 t x 64 co
        if (L.C_out_pad <= 128) return launch_cfg<ENGINE, 2, 2, 2, 2, CK>(p, stream);  // 128 t x 128 co
        return launch_cfg<ENGINE, 2, 4, 2, 2, CK>(p, stream);                           // 128 t x 256 co
    }
}

hipError_t conv1d_launch(const PackedConv& L, ConvParams p, hipStream_t stream) {
    p.w_hi = L.w_hi;
    p.w_lo = L.w_lo;
    p.w_lo2 = nullptr;
    p.bias = L.bias;
    p.C_in = L.C_in;
    p.C_in_pad = L.C_in_pad;
    p.C_out = L.C_out;
    p.C_out_pad = L.C_out_pad;
    p.K = L.K;
    p.dil = L.dil;
    p.stride = L.stride;
    p.pad = L.pad;
    p.gate_H = L.gate_H;
    if (p.out_div == 0.f) p.out_div = 1.f;
    if (p.out_mul == 0.f) p.out_mul = 1.f;
    switch (L.engine) {
        case ENG_F32:
            return L.CK == 32 ? launch_engine<ENG_F32, 32>(L, p, stream) : launch_engine<ENG_F32, 64>(L, p, stream);
        case ENG_BF16:
            return L.CK == 32 ? launch_engine<ENG_BF16, 32>(L, p, stream) : launch_engine<ENG_BF16, 64>(L, p, stream);
        default:
            return L.CK == 32 ? launch_engine<ENG_BF16X3, 32>(L, p, stream)
                              : launch_engine<ENG_BF16X3, 64>(L, p, stream);
    }
}

size_t packed_elems(const PackedConv& L) { return (size_t)L.K * L.C_in_pad * L.C_out_pad; }

} // namespace dtts
