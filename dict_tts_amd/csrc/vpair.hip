// One ResBlock1 iteration  x <- x + c2(leaky_relu(c1(leaky_relu(x))))  (modules/hifigan/hifigan.py:51-58) in ONE
// kernel, for the C = 128 stage (bf16 MFMA).  Per-convolution, that stage is HBM-bound: 2.2 KB per row and iteration
// (bf16 copies in/out, the bf16 xt round trip, fp32 residual in/out).  Here a workgroup
//   * reads the fp32 stream once (tile + halo of d(k-1)/2 + (k-1)/2 rows), applies leaky_relu + bf16 rounding while
//     staging it into LDS (the same rounding point as everywhere else),
//   * runs c1 for 128 rows of xt, barriers, overwrites the LDS tile with bf16(leaky_relu(xt)) (zero outside the
//     utterance = the reference's zero padding), runs c2 for 128 - (k-1) valid output rows,
//   * re-reads the centre rows of x for the residual (L2-hot: this block staged them a moment ago) in the coalesced
//     epilogue and writes only the fp32 result (or accumulates into the stage sum xs for the last iteration).
// HBM bytes per row and iteration: ~0.8 KB in + 0.5 KB out.  4 waves over output channels, 3 workgroups per CU.
#include "vpair.h"
#include "tune_env.h"

#include "rb_common.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace dtts {

// TT = 128: 3 workgroups per CU; TT = 256: every weight fragment feeds 8 MFMAs instead of 4 (half the weight stream
// through the texture path, half the halo), 2 workgroups per CU when the LDS tile allows
// C = 256 (NT = 2 co-tiles per wave): the stage-1 ResBlocks; 128-row tiles only.
// WT = 2 (C = 128): the four waves as 2 (time) x 2 (output channels), two co-tiles per wave — every activation fragment read from LDS feeds two
// MFMAs instead of one (half the ds_read_b128 traffic; twice the weight fragments through the texture path, as at C = 256).
// X16 (round 6; EL_F16 only): the INPUT stream x is fp16 — iterations 1 and 2 of a ResBlock, whose predecessor stored its result with
// p.y16.  fp16(x) is exactly what the fp32 stream's staging computed as the convolution operand, so c1's operands keep their bits; what
// changes is the residual add (x16 + xt instead of x32 + xt).  The staging loads 8 channels per thread (half the bytes, half the
// accesses), needs no conversion (leaky_relu on the packed pairs as they arrive) and writes 16 bytes per LDS access; the epilogue
// re-reads 8 bytes per four channels and widens them in registers.
template <int C, int TT, int EL, bool GUARD, int WT = 1, bool X16 = false>
__global__ __launch_bounds__(256, (C == 128 && TT == 128 && WT == 1) ? 3 : 2) void vpair_kernel(const VPairParams p) {
    static_assert(!X16 || EL == EL_F16, "the 16-bit stream is fp16");
    const int mode = p.mode;
    constexpr bool PS = !(C == 128 && TT == 128 && WT == 1);   // persistent workgroups (below); not the 3-per-CU configuration, which loses 9 % with them
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int WC = 4 / WT, TW = TT / WT;     // waves over the output channels; rows of a time-wave
    constexpr int MT = (TW == 192 || TW == 96) ? 3 : (TW == 64 ? 2 : 4), NT = C / (32 * WC), MH = TW / (32 * MT), MTT = MT * MH;   // TW = 192: two passes of 3 row tiles
    constexpr int PITCH = C * 2 + 16, NKG = C / 16, NCT = C / 32;
#ifndef VP_XA1
#define VP_XA1 1
#endif
    constexpr bool XA1 = VP_XA1 != 0;   // one activation-fragment set in the contractions (rb_common.h)
    constexpr int EP = C * 4 + 16, F4 = C / 4;
    static_assert(NCT == WC * NT && (NT == 1 || MH == 1) && WT * WC == 4, "4 waves: WT over time x WC over the output channels");
    const int tid0 = threadIdx.x;
    const int h1 = p.dil * (p.K - 1) / 2, h2 = (p.K - 1) / 2;
    const int TTe = TT - 2 * h2;             // valid output rows per tile
    const int S = DTTS_DBG(p, 1) ? 0 : p.K * NKG;
    // ---- persistent workgroups (as in rblock.hip): the valid tiles of the batch, ceil(len_b / TTe) per utterance, are numbered
    // through and workgroup w takes tiles w, w + G, ...; the table of the per-utterance tile counts' prefix sums and the lengths
    // live in LDS behind the tile.  A tile's stores drain while the next tile is staged, and no workgroup is launched per tile.
    int* pre = (int*)(smem + p.pre_off);
    int total = 1, b = 0;
    if constexpr (PS) {
        for (int i = tid0; i < p.B; i += 256) {
            const int l = p.lens ? p.lens[i] : p.T;
            pre[p.B + 1 + i] = (l + TTe - 1) / TTe;
            pre[2 * p.B + 1 + i] = l;
        }
        __syncthreads();
        for (int i = tid0; i <= p.B; i += 256) {
            int a = 0;
            for (int u = 0; u < i; ++u) a += pre[p.B + 1 + u];
            pre[i] = a;
        }
        __syncthreads();
        total = pre[p.B];
    }
#pragma unroll 1
    for (int j = PS ? blockIdx.x : 0; j < total;) {
    // the next tile: static (j + G) or, with p.tile_ctr, the next unclaimed tile of the launch (rblock.hip: dynamic tile claiming) — the
    // atomic is issued here, its result is broadcast through LDS behind the barrier that ends the tile
    unsigned claim = 0;
    if (PS && p.tile_ctr && tid0 == 0) claim = atomicAdd(p.tile_ctr, 1u);
    const int j_static = j + (int)gridDim.x;
    // (the thread index passes through an opaque move every tile: everything derived from it is then recomputed per tile — a few
    // VALU instructions — instead of being hoisted out of the tile loop by hipcc and spilled to scratch for lack of registers)
    int tid = tid0;
    if constexpr (PS) asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wv / WT, wt = wv % WT;
    int len, t0;
    if constexpr (PS) {
        while (pre[b + 1] <= j) ++b;             // the utterance index only moves forward
        b = __builtin_amdgcn_readfirstlane(b);
        // (readfirstlane: a length in a VGPR would put every buffer resource below in VGPRs: a waterfall loop around each buffer access)
        len = __builtin_amdgcn_readfirstlane(pre[2 * p.B + 1 + b]);
        t0 = __builtin_amdgcn_readfirstlane((j - pre[b]) * TTe);
    } else {   // one tile per workgroup: grid (tiles, utterances)
        b = blockIdx.y;
        len = __builtin_amdgcn_readfirstlane(p.lens ? p.lens[b] : p.T);
        t0 = blockIdx.x * TTe;
        if (t0 >= len) return;
    }
    const long long brow = (long long)b * p.T;

#ifdef DTTS_ABLATE
    unsigned long long tq[6];
#define VP_STAMP(i) tq[i] = __builtin_amdgcn_s_memtime()
#else
#define VP_STAMP(i)
#endif
    VP_STAMP(0);
    int n_ovf = 0;
    uint4 ring[4][NT];
    const size_t wlane = (size_t)wc * NT * 64 + lane;   // the wave's first co-tile
    rb_preload<NT>(ring, p.w1 + wlane, NCT * 64);   // c1's first weights fly while the tile is staged

    // ---- stage bf16(leaky_relu(x)) for rows [t0 - h2 - h1, t0 - h2 + TT + h1) ; zero outside the utterance.
    // Buffer loads over the utterance [0, len) x C: an out-of-range row (t < 0 wraps to a huge unsigned offset,
    // t >= len exceeds num_records) returns zeros = the reference's zero padding, with no per-access compare; a thread
    // keeps its column and walks rows in steps of 8, so each access costs one v_add for its address.
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    constexpr int XB = X16 ? 2 : 4;                 // bytes per element of the input stream
    const char* xu = (const char*)p.x + brow * C * XB;
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)xu, 0, len * C * XB, 0x00020000);
    constexpr int RSTEP = 256 / F4;                 // rows between two accesses of a thread (8)
    const int c4 = tid % F4, r0 = tid / F4;
    const int a0 = t0 - h2 - h1;
    const int arows = TT + 2 * h1;
    if constexpr (X16) {
        if (!DTTS_DBG(p, 4)) {
            constexpr int F8 = C / 8, RS8 = 256 / F8;   // 8 channels (16 bytes) per access; rows between two accesses of a thread (16 / 8)
#ifndef VP_U16
#define VP_U16 (RS8 == 16 ? 10 : 12)
#endif
            // independent loads in flight per thread and batch.  (All 19 / 23 accesses of a tile in ONE batch — one exposed round trip instead of
            // two — shortens wave 0's staging phase and not the launch: LABNOTES round 6 (b).)
            constexpr int U = VP_U16;
            const int c8 = tid % F8, r8 = tid / F8;
            const int nk = (arows + RS8 - 1) / RS8;
            const int voff0 = ((a0 + r8) * C + c8 * 8) * 2;
            char* lrow = smem + r8 * PITCH + c8 * 16;
            const int rlim = p.tile_rows - r8;      // the last pass may reach past the tile's LDS rows (the tile table lives there)
            const _Float16 hs = (_Float16)0.1f;
            const f16x2_t slope2 = {hs, hs};
            for (int kb = 0; kb < nk; kb += U) {
                u32x4 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u)   // (an access past the tile's rows is sent out of range: zeros, no memory traffic)
                    v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, kb + u < nk ? voff0 + (kb + u) * (RS8 * C * 2) : (int)0x80000000, 0, 0);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (kb + u >= nk || (kb + u) * RS8 >= rlim) continue;
                    u32x4 r;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned w = v[u][e];   // (a scalar copy first: __builtin_bit_cast applied to a vector-element lvalue reads element 0 whatever the index)
                        const f16x2_t hv = __builtin_bit_cast(f16x2_t, w);
                        if constexpr (GUARD) {   // census: a stored value beyond the fp16 range arrives as +-inf (only the rows this tile outputs)
                            const int rr = r8 + (kb + u) * RS8 - h1 - h2;
                            if (rr >= 0 && rr < TTe)
                                n_ovf += (__builtin_fabsf((float)hv[0]) > 65504.f ? 1 : 0) + (__builtin_fabsf((float)hv[1]) > 65504.f ? 1 : 0);
                        }
                        r[e] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(hv, hv * slope2));
                    }
                    *(u32x4*)(lrow + (kb + u) * (RS8 * PITCH)) = r;
                }
            }
        }
    } else if (!DTTS_DBG(p, 4)) {
#ifndef VP_U
#define VP_U 12
#endif
        constexpr int U = VP_U;                     // independent loads in flight per thread and batch
        const int nk = (arows + RSTEP - 1) / RSTEP;
        const int voff0 = ((a0 + r0) * C + c4 * 4) * 4;
        char* lrow = smem + r0 * PITCH + c4 * 8;
        for (int kb = 0; kb < nk; kb += U) {
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, voff0 + (kb + u) * (RSTEP * C * 4), 0, 0);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (kb + u >= nk) continue;
                const f32x4 f = __builtin_bit_cast(f32x4, v[u]);
                if constexpr (GUARD) {   // census: only the rows this tile outputs (rows outside the utterance arrive as zeros)
                    const int rr = r0 + (kb + u) * RSTEP - h1 - h2;
                    n_ovf += (rr >= 0 && rr < TTe) ? ovf4(f, 0.1f) : 0;
                }
                *(uint2*)(lrow + (kb + u) * (RSTEP * PITCH)) = act4<EL>(f, 0.1f);
            }
        }
    }
    // this lane's bias quads (channel of accumulator slot 4q+e of co-tile n: (wc * NT + n) * 32 + 8q + 4 (lane >> 5) + e)
    f32x4 bb[NT][4];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) bb[n][q] = *(const f32x4*)(p.b1 + (wc * NT + n) * 32 + 8 * q + 4 * (lane >> 5));
    __syncthreads();
    VP_STAMP(1);

    // ---- c1: xt rows r = 0..127  <->  global t0 - h2 + r ; reads staged rows r + tap * d
    f32x16 acc[MTT][NT];
    f32x16 cinit[NT];   // bias pattern of this lane's 16 channel slots: the C operand of every tile's first MFMA
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) cinit[n][4 * q + e] = bb[n][q][e];
    auto load_b2 = [&]() {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q) bb[n][q] = *(const f32x4*)(p.b2 + (wc * NT + n) * 32 + 8 * q + 4 * (lane >> 5));
    };
    if (NT == 1) load_b2();   // lands while c1 runs; at C = 256 its 32 registers do not fit beside c1's (spills): fetched after c1 there
    const int xlane = (wt * TW + (lane & 31)) * PITCH + (lane >> 5) * 16;
    rb_contract<EL, MT, NT, NKG, PITCH, true, MH, XA1>(acc, ring, smem, xlane, p.w1 + wlane, S, p.dil * PITCH, 0, &cinit);
    if (NT != 1) load_b2();
    rb_preload<NT>(ring, p.w2 + wlane, NCT * 64);
    VP_STAMP(2);
    __syncthreads();   // every wave is done reading the x tile
    // ---- bf16(leaky_relu(xt)) overwrites it (rows 0..127), zero outside the utterance
#pragma unroll
    for (int m = 0; m < (DTTS_DBG(p, 8) ? 0 : MTT); ++m) {
        const int r = wt * TW + m * 32 + (lane & 31);
        const int t = t0 - h2 + r;
        const bool inb = t >= 0 && t < len;
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v4 = {acc[m][n][4 * q], acc[m][n][4 * q + 1], acc[m][n][4 * q + 2], acc[m][n][4 * q + 3]};
                uint2 pk = act4<EL>(v4, 0.1f);
                if constexpr (GUARD) n_ovf += (inb && r >= h2 && r < h2 + TTe) ? ovf4(v4, 0.1f) : 0;
                if (!inb) pk = make_uint2(0, 0);
                *(uint2*)(smem + r * PITCH + ((wc * NT + n) * 32 + 8 * q + 4 * (lane >> 5)) * 2) = pk;
            }
    }
    __syncthreads();
    VP_STAMP(3);
    // ---- c2: output rows o = 0..127 <-> global t0 + o (valid for o < TTe) ; reads xt rows o + tap
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) cinit[n][4 * q + e] = bb[n][q][e];
    rb_contract<EL, MT, NT, NKG, PITCH, true, MH, XA1>(acc, ring, smem, xlane, p.w2 + wlane, S, PITCH, 0, &cinit);
    VP_STAMP(4);
    __syncthreads();   // the xt tile is dead: the staging buffer of the epilogue aliases it

    if constexpr (GUARD) {
        if (n_ovf) atomicAdd(p.ovf, (unsigned long long)n_ovf);   // (output rows only: every in-utterance row exactly once per launch)
    }
    if (DTTS_DBG(p, 2)) {
        if (acc[0][0][0] == 123.456f) p.y[0] = 1.f;
        if constexpr (!PS) break;
        j = j_static;
        continue;
    }
    // ---- epilogue: 32-row slabs through LDS, whole rows out; residual x re-read (L2), xs accumulated per mode.
    // Buffer loads / stores again: rows >= len are dropped by the range check, the garbage rows o >= TTe of the last
    // slab are sent out of range explicitly.  The reads of slab m+1 are issued before slab m is processed.
    constexpr int PER = WT * 32 * F4 / 256;     // a pass moves one 32-row slab of every time-wave
    float* yu = p.y + brow * C;
    // (p.y16, mode 1 only: the result leaves as fp16 — the next iteration's X16 input; same rows, half the bytes)
    const auto rs_y = p.y16 ? __builtin_amdgcn_make_buffer_rsrc((void*)((char*)p.y + brow * C * 2), 0, len * C * 2, 0x00020000)
                            : __builtin_amdgcn_make_buffer_rsrc((void*)yu, 0, len * C * 4, 0x00020000);
    const auto rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ya ? p.ya + brow * C : (unsigned short*)yu), 0, len * C * 2, 0x00020000);
    const int eoff0 = (t0 * C + c4 * 4) * 4;
    auto eoff = [&](int m, int u) {                 // byte offset of (slab m, access u) or out of range
        const int sr = u * RSTEP + r0;              // staged row: slab m of time-wave sr / 32
        const int o = (sr >> 5) * TW + m * 32 + (sr & 31);
        return o < TTe ? eoff0 + o * (C * 4) : (int)0x80000000;
    };
    // The residual rows (and, modes 2 / 3, the stage sum) of a slab are requested XD - 1 (SD - 1) slabs ahead: one slab at C = 128, none at
    // C = 256 (registers).  Round 6 stamps had this epilogue at 34 k cycles of a 90 k-cycle tile at C = 128 — a chain of exposed round trips; a
    // lookahead of 3 - 5 slabs (-DVP_XD / -DVP_SD; an fp16 row costs half the registers) shortens wave 0's phase by 27 - 40 % and leaves every
    // launch where it was (the co-resident workgroup fills the gaps either way): the shallow ring stays.  LABNOTES round 6 (b).
#ifndef VP_XD
#define VP_XD (NT == 1 ? 2 : 1)
#endif
#ifndef VP_SD
#define VP_SD (NT == 1 ? 2 : 1)
#endif
    constexpr int XD = VP_XD < MTT ? VP_XD : MTT, SD = VP_SD < MTT ? VP_SD : MTT;
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
    typedef typename std::conditional<X16, u32x2, u32x4>::type xrow_t;
    xrow_t xin[XD][PER];
    u32x4 sold[SD][PER];
    auto fetch_x = [&](int m, xrow_t (&xi)[PER]) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int off = eoff(m, u);
            if constexpr (X16) xi[u] = __builtin_amdgcn_raw_buffer_load_b64(rs_x, off == (int)0x80000000 ? off : off >> 1, 0, VP_XI_AUX);
            else xi[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, VP_XI_AUX);
        }
    };
    auto fetch_s = [&](int m, u32x4 (&so)[PER]) {
        if (mode < 2) return;
#pragma unroll
        for (int u = 0; u < PER; ++u) so[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, eoff(m, u), 0, VP_LD_AUX);
    };
#pragma unroll
    for (int m = 0; m < XD - 1; ++m) fetch_x(m, xin[m]);
#pragma unroll
    for (int m = 0; m < SD - 1; ++m) fetch_s(m, sold[m]);
#pragma unroll
    for (int m = 0; m < MTT; ++m) {
        if (m) __syncthreads();
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[m][n][4 * q + e];
                *(f32x4*)(smem + (wt * 32 + (lane & 31)) * EP + ((wc * NT + n) * 32 + 8 * q + 4 * (lane >> 5)) * 4) = v;
            }
        if (m + XD - 1 < MTT) fetch_x(m + XD - 1, xin[(m + XD - 1) % XD]);
        if (m + SD - 1 < MTT) fetch_s(m + SD - 1, sold[(m + SD - 1) % SD]);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int off = eoff(m, u);
            f32x4 xr;
            if constexpr (X16) {
                const unsigned w0 = xin[m % XD][u][0], w1 = xin[m % XD][u][1];   // (scalar copies: see the staging loop)
                const f16x2_t g0 = __builtin_bit_cast(f16x2_t, w0), g1 = __builtin_bit_cast(f16x2_t, w1);
                xr = f32x4{(float)g0[0], (float)g0[1], (float)g1[0], (float)g1[1]};
            } else xr = __builtin_bit_cast(f32x4, xin[m % XD][u]);
            f32x4 o = *(const f32x4*)(smem + (r0 + u * RSTEP) * EP + c4 * 16) + xr;   // x = xt + x
            if (mode >= 2) o += __builtin_bit_cast(f32x4, sold[m % SD][u]);                                                 // xs += x
            if (mode == 3) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = o[e] / p.div;
            }
            if (p.y16) {   // (no saturation: a value beyond the fp16 range is stored as +-inf, reaches conv_post as a non-finite sum: the always-on detector)
                const u32x2 pk = {pack2<EL_F16>(o[0], o[1]), pack2<EL_F16>(o[2], o[3])};
                __builtin_amdgcn_raw_buffer_store_b64(pk, rs_y, off == (int)0x80000000 ? off : off >> 1, 0, VP_ST_AUX);
            } else if (!(mode == 3 && p.ya && p.drop_y))   // the stage's consumers read only the bf16 copy
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs_y, off, 0, VP_ST_AUX);
            if (mode == 3 && p.ya) {
                const u32x2 pk = {pack2bf(lrelu(o[0], p.slope), lrelu(o[1], p.slope)), pack2bf(lrelu(o[2], p.slope), lrelu(o[3], p.slope))};
                __builtin_amdgcn_raw_buffer_store_b64(pk, rs_a, off == (int)0x80000000 ? off : off >> 1, 0, VP_ST_AUX);
            }
        }
    }
#ifdef DTTS_ABLATE
    if (p.stats) {
        asm volatile("s_waitcnt vmcnt(0)");   // the stores' acknowledgements are part of the epilogue's time here
        VP_STAMP(5);
        if (tid == 0) {
            for (int i = 0; i < 5; ++i) atomicAdd(p.stats + i, tq[i + 1] - tq[i]);
            atomicAdd(p.stats + 5, 1ull);
        }
    }
#endif
    if constexpr (!PS) break;
    if (p.tile_ctr && tid0 == 0) pre[3 * p.B + 1] = (int)gridDim.x + (int)claim;
    __syncthreads();   // the epilogue's staging rows alias the tile the next iteration stages into
    j = p.tile_ctr ? __builtin_amdgcn_readfirstlane(pre[3 * p.B + 1]) : j_static;
    }   // (tiles of this workgroup)
}

bool vpair_supported(int C, int K, int dil) {
    return (C == 128 || C == 256) && (K & 1) && K >= 3 && K <= 11 && dil >= 1 && dil <= 5;
}

template <int CC, int TT, int EL, bool GUARD = false, int WT = 1, bool X16 = false>
static hipError_t vpair_launch_tt(const VPairParams& p, hipStream_t stream) {
    if constexpr (EL == EL_F16 && !X16) {
        if (p.x16) return vpair_launch_tt<CC, TT, EL, GUARD, WT, true>(p, stream);
    }
    if ((p.x16 && !X16) || (p.y16 && (EL != EL_F16 || p.mode != 1))) return hipErrorInvalidValue;
    if ((long long)p.T * CC * 4 >= (1LL << 31)) return hipErrorInvalidValue;   // 32-bit byte offsets inside an utterance's buffer resource
    constexpr int PITCH = CC * 2 + 16;
    const int h1 = p.dil * (p.K - 1) / 2, h2 = (p.K - 1) / 2;
    const int TTe = TT - 2 * h2;
    // staged rows, rounded up to the staging loop's step, or one spare tap for c1's activation prefetch, whichever reaches further;
    // the xt phase needs TT + (K-1) + 1 rows, the epilogue 32 fp32 rows
    constexpr int RSTEP = 256 / (CC / 4);
    size_t rows = (size_t)TT + 2 * h1 + std::max(p.dil + 1, RSTEP);
    if (rows < (size_t)TT + p.K + 1) rows = TT + p.K + 1;
    size_t lds = rows * PITCH;
    const size_t ep = (size_t)WT * 32 * (CC * 4 + 16);
    if (ep > lds) lds = ep;
    VPairParams q = p;
    q.tile_rows = (int)rows;
    q.pre_off = (int)lds;                          // tile table: prefix sums [B + 1], counts [B], lengths [B]
    constexpr bool PS = !(CC == 128 && TT == 128 && WT == 1);
    if (PS) lds += (size_t)(3 * p.B + 2) * sizeof(int);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if constexpr (EL == EL_F16 && !GUARD) {
        if (p.ovf) return vpair_launch_tt<CC, TT, EL, true, WT, X16>(p, stream);
    }
    auto kern = vpair_kernel<CC, TT, EL, GUARD, WT, X16>;
    // per device (hipFuncSetAttribute is per device; a process may hold contexts on several GPUs)
    static bool configured_dev[64] = {};
    int cur_dev = 0;
    (void)hipGetDevice(&cur_dev);
    bool& configured = configured_dev[cur_dev & 63];
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        configured = true;
    }
    if (!PS) {
        hipLaunchKernelGGL(kern, dim3((p.T + TTe - 1) / TTe, p.B), dim3(256), lds, stream, q);
        return hipGetLastError();
    }
    // persistent workgroups: as many as are resident at once (LDS, the kernel's register bound), never more than there can be tiles
    static int cus_dev[64] = {};
    int& cus = cus_dev[cur_dev & 63];
    if (!cus) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, cur_dev) != hipSuccess) return hipErrorInvalidDevice;
        cus = prop.multiProcessorCount;
    }
    const int per_cu = std::max(1, std::min((int)(160 * 1024 / lds), (CC == 128 && TT == 128 && WT == 1) ? 3 : 2));
    const long long max_tiles = (long long)p.B * ((p.T + TTe - 1) / TTe);
    const int grid = (int)std::min<long long>((long long)std::max(1, cus - cu_reserve()) * per_cu, max_tiles);
    if (grid <= 0) return hipSuccess;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, q);
    return hipGetLastError();
}

#ifndef VP_TT96
#define VP_TT96 1
#endif
// CUs of the current device (cached per device)
static int vpair_cus() {
    static int cus_dev[64] = {};
    int cur_dev = 0;
    (void)hipGetDevice(&cur_dev);
    int& cus = cus_dev[cur_dev & 63];
    if (!cus) {
        hipDeviceProp_t prop;
        cus = hipGetDeviceProperties(&prop, cur_dev) == hipSuccess ? prop.multiProcessorCount : 256;
    }
    return cus;
}

template <int EL>
static hipError_t vpair_launch_el(const VPairParams& p, int C, hipStream_t stream) {
    // small batches (B = 1: one sentence): the default tiles would leave most CUs without one and the launch takes as long as ONE tile
    // -> half-size tiles (more halo rows recomputed, twice the weight stream per row, but twice the CUs at work)
    static const bool small_ok = [] { const char* e = ablate_env("DTTS_VP_SMALL"); return !e || atoi(e) != 0; }();
    auto tiles_of = [&](int tt) { return (long long)p.B * ((p.T + (tt - (p.K - 1)) - 1) / (tt - (p.K - 1))); };
    if (C == 256) {
#ifdef VP_FORCE64    // experiment: 64-row tiles at C = 256 whatever the tile count
        return vpair_launch_tt<256, 64, EL>(p, stream);
#endif
        if (small_ok && 2 * tiles_of(128) <= vpair_cus()) return vpair_launch_tt<256, 64, EL>(p, stream);
        // 128-row tiles, or 96-row ones where only those leave room for TWO workgroups per CU (one workgroup = one wave per SIMD exposes
        // every latency of the memory phases: k = 7 with dilation 5, k = 11 with dilation 3)
        auto lds_of = [&](int tt) { return ((size_t)tt + (size_t)p.dil * (p.K - 1) + std::max(p.dil + 1, 4)) * (256 * 2 + 16) + (size_t)(3 * p.B + 2) * sizeof(int); };
        if (2 * lds_of(128) > 160 * 1024 && 2 * lds_of(96) <= 160 * 1024 && VP_TT96) return vpair_launch_tt<256, 96, EL>(p, stream);
        return vpair_launch_tt<256, 128, EL>(p, stream);
    }
    // 256-row tiles while two workgroups still fit a CU's 160 KB of LDS (all but k = 11 with dilation 5)
    const size_t rows256 = (size_t)256 + (size_t)p.dil * (p.K - 1) + std::max(p.dil + 1, 8);
    const bool big = (rows256 * (128 * 2 + 16) + (size_t)(3 * p.B + 2) * sizeof(int)) * 2 <= 160 * 1024;
#ifdef VP_FORCE128   // experiment (small grids: long form, B = 1): 128-row tiles at C = 128 whatever the tile count
    return vpair_launch_tt<128, 128, EL>(p, stream);
#endif
    if (small_ok && 2 * tiles_of(256) <= vpair_cus()) return vpair_launch_tt<128, 128, EL>(p, stream);
#ifdef VP_NT2   // experiment: 2 x 2 waves, two co-tiles per wave
    if (big) return vpair_launch_tt<128, 256, EL, false, 2>(p, stream);
    return vpair_launch_tt<128, 192, EL, false, 2>(p, stream);
#endif
    if (big) return vpair_launch_tt<128, 256, EL>(p, stream);
    // k = 11 with dilation 5: 192-row tiles (two workgroups per CU, persistent) instead of 128-row ones (three, one tile each)
    return vpair_launch_tt<128, 192, EL>(p, stream);
}

hipError_t vpair_launch(const VPairParams& p, int C, hipStream_t stream) {
    if (!vpair_supported(C, p.K, p.dil)) return hipErrorInvalidValue;
    return p.el == EL_F16 ? vpair_launch_el<EL_F16>(p, C, stream) : vpair_launch_el<EL_BF16>(p, C, stream);
}

} // namespace dtts
