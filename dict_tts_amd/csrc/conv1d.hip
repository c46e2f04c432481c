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

namespace dtts {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ unsigned f2bf(float f) {  // round-to-nearest-even fp32 -> bf16 bits (hardware convert)
    const __bf16 h = (__bf16)f;
    return (unsigned)__builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float bf2f(unsigned h) { return __uint_as_float(h << 16); }

// FULL: the whole input width C_in_pad is staged ONCE (LDS rows of C_in_pad elements) and the CK-chunks are contracted from it back to
// back in the same order — bit-identical to the chunk-by-chunk form, without its per-chunk staging round trip and two barriers.  For the
// short-sequence configurations (one workgroup per CU, one wave per SIMD: nothing else hides those latencies).
template <int ENGINE, int MT, int NT, int WT, int WC, int CK, bool FULL = false>
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
    const int pitch = FULL ? p.C_in_pad * ES + 16 : PITCH;   // (C_in_pad * ES + 16) mod 256 == 16 as well for the widths in use: same bank spread
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
        const int PIECES = FULL ? width / 4 : CK / 4;
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
        if constexpr (FULL) {
            stage(0, p.C_in_pad, 0);
            __syncthreads();
        }
        for (int ci0 = 0; ci0 < p.C_in_pad; ci0 += CK) {
            if constexpr (!FULL) {
                stage(ci0, CK, 0);
                __syncthreads();
            }
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
                const int abase = ((wt * MT) * 32 + (lane & 31)) * p.stride * pitch + (lane >> 5) * (KG / 2) * ES + (FULL ? ci0 * ES : 0);
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
            if constexpr (!FULL) __syncthreads();
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

template <int ENGINE, int MT, int NT, int WT, int WC, int CK, bool FULL = false>
static hipError_t launch_cfg(const ConvParams& p, hipStream_t stream) {
    constexpr int ES = (ENGINE == ENG_F32) ? 4 : 2;
    constexpr int PITCH = CK * ES + 16;
    constexpr int TT = 32 * MT * WT, CO_T = 32 * NT * WC;
    const int rows = (TT - 1) * p.stride + (p.K - 1) * p.dil + 1;
    size_t lds = (size_t)rows * (FULL ? p.C_in_pad * ES + 16 : PITCH) * (ENGINE == ENG_BF16X3 ? 2 : 1);
    auto kern = conv1d_cl_kernel<ENGINE, MT, NT, WT, WC, CK, FULL>;
    static size_t configured_dev[64] = {};   // per device: hipFuncSetAttribute is per device
    int cur_dev = 0;
    if (lds > 65536) (void)hipGetDevice(&cur_dev);
    size_t& configured = configured_dev[cur_dev & 63];
    if (lds > 65536 && lds > configured) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured = lds;
    }
    dim3 grid((p.T_out + TT - 1) / TT, (p.C_out_pad + CO_T - 1) / CO_T, p.B);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, p);
    return hipGetLastError();
}

template <int ENGINE, int CK>
static hipError_t launch_engine(const PackedConv& L, const ConvParams& p, hipStream_t stream) {
    if constexpr (ENGINE == ENG_F32) {
        // short sequences (the T_w ~ 27 encoder, B = 1): latency-bound by the serial fp32 MFMA chain of one wave (64 cycles per
        // 32x32x2 MFMA); one co-tile per wave halves that chain at twice the workgroups (bit-identical: the order over K is unchanged)
        // (gated layers keep two co-tiles per wave: the tanh tile and its sigmoid partner meet in the epilogue)
        // (FULL: the input tile is staged once for all its CK-chunks, while the rows fit the LDS)
        const bool full = p.C_in_pad > CK && ((size_t)(31 * p.stride + (p.K - 1) * p.dil + 1) * (p.C_in_pad * 4 + 16) <= 150 * 1024);
        if (p.T_out <= 64 && !p.gate_H) return full ? launch_cfg<ENGINE, 1, 1, 1, 4, CK, true>(p, stream) : launch_cfg<ENGINE, 1, 1, 1, 4, CK>(p, stream);  // 32 t x 128 co
        if (p.T_out <= 64) return full ? launch_cfg<ENGINE, 1, 2, 1, 4, CK, true>(p, stream) : launch_cfg<ENGINE, 1, 2, 1, 4, CK>(p, stream);   // 32 t x 256 co
        return launch_cfg<ENGINE, 1, 2, 4, 1, CK>(p, stream);                      // 128 t x 64 co
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
