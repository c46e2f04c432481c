// Fused HifiGAN ResBlock1 (modules/hifigan/hifigan.py:27-58), 16-bit MFMA: every k at C = 64 / 32, k = 3 at C = 128 / 256.
//
// Unfused, a ResBlock is six convolutions that each stream the whole activation through HBM; at C <= 64 their
// arithmetic intensity (C*k/2 FLOP/B) is far below the MFMA/HBM ridge, i.e. the vocoder's last two stages (36 % of
// its FLOPs) are HBM-bound.  Here one workgroup keeps a time tile resident for all six convolutions:
//   * the fp32 residual stream x lives in REGISTERS in MFMA accumulator layout (D[co][t]),
//   * the bf16 leaky_relu copy that feeds the next convolution lives in ONE LDS buffer (A and the intermediate
//     xt time-share it: conv reads -> barrier -> overwrite -> barrier),
//   * weights stream from L2 through a 4-deep register ring (22..90 KB per conv, shared by every workgroup),
//   * the tile carries a halo of 6*(k-1) rows per side (sum of the six receptive half-widths) that is recomputed;
//     rows outside the utterance are forced to zero after every activation = the reference's zero padding.
// HBM traffic per ResBlock drops from ~9 passes to: read x once, read-modify-write the stage accumulator once.
// PS = 1 (all but C = 32): persistent workgroups walk the batch's valid tiles, the next tile's x is fetched straight into the
// residual registers (accumulator layout, no LDS transposition) slab by slab as the epilogue releases them.
#include "rblock.h"
#include "tune_env.h"
#include "rb_common.h"

#include <algorithm>
#include <type_traits>
#include <cstdlib>

namespace dtts {

#ifdef RB_STAMP   // per-phase clock stamps (tools/rb_stamps.py; a variant build, never the release library): sums over the tiles of every workgroup,
                  // wave 0 (table 0) and the last wave (table 1), one row per (C, k): s_memrealtime ticks of 10 ns
__device__ unsigned long long rb_stamps[2][12][16];
extern "C" __attribute__((visibility("default"))) int dtts_debug_rb_stamps(unsigned long long* host, int reset) {
    hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(rb_stamps), sizeof(rb_stamps));
    if (e == hipSuccess && reset) {
        static unsigned long long z[2][12][16];
        e = hipMemcpyToSymbol(HIP_SYMBOL(rb_stamps), z, sizeof z);
    }
    return (int)e;
}
#define RB_T(k) do { const unsigned long long _n = __builtin_amdgcn_s_memrealtime(); tq[k] += _n - t_last; t_last = _n; } while (0)
#else
#define RB_T(k)
#endif

// TB (two LDS activation buffers, experiment of round 5): leaky_relu(x) and leaky_relu(xt) live in SEPARATE buffers, so the rewrite after a
// contraction needs no write-after-read barrier (nobody reads the buffer it writes): two workgroup barriers per iteration instead of four.
template <int C, int MT, int NT, int WT, int WC, int EL, int PS, bool GUARD, bool TB = false>
__global__ __launch_bounds__(64 * WT * WC, (64 * WT * WC <= 256) ? 2 : 1) void rblock_kernel(const RBlockParams p) {
    static_assert(WC * NT * 32 == C, "channel tiling must cover C");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int THREADS = 64 * WT * WC;
    constexpr int W = 32 * MT * WT;
    constexpr int PITCH = C * 2 + 16;
    constexpr int NKG = C / 16;
    // ONE activation-fragment set (rb_common.h: XA1) in the 640-row C = 64 instantiation: with the double buffer it sat at 256 VGPRs with 13 of them
    // spilled; the single set (a row tile's next fragment is read right behind the MFMAs that consumed the current one and lands while the other row
    // tiles' MFMAs run) takes 249 and spills nothing.  Same instruction order per accumulator: same bits; -0.7 % on the vocoder (LABNOTES round 5 (Y)).
#ifndef RB_XA1
#define RB_XA1 1
#endif
#ifndef RB_XA1_ALL
#define RB_XA1_ALL 1   // every C >= 64 instantiation (C = 32: neutral / +2 %, keeps the double buffer)
#endif
    constexpr bool XA1 = RB_XA1 && ((C == 64 && MT == 5) || (RB_XA1_ALL && C >= 64));
#ifndef RB_XA1_32
#define RB_XA1_32 1
#endif
    constexpr bool XA1_32 = RB_XA1_32 != 0;   // the same in the C = 32 contraction (rb2_contract)
    constexpr int EP = C * 4 + 16;                 // fp32 staging row
    constexpr int F4 = C / 4, SROWS = WT * 32;
    constexpr size_t ACT_BYTES = (size_t)(W + 2 * RB_GUARD) * PITCH;
    char* act = smem;                                    // leaky_relu(x)
    char* act2 = TB ? smem + ACT_BYTES : smem;           // leaky_relu(xt): its own buffer (TB) or the same one, time-shared
    char* stage = TB ? act2 + RB_GUARD * PITCH : smem + ACT_BYTES;   // epilogue transposition; TB: over the xt buffer's tile rows (dead after the last contraction, rewritten whole by the next tile; its guard bands stay zero)

    // per-thread coordinates; PS refreshes them through an opaque move at every tile (see the tile loop)
    int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int wt = wave % WT, wc = wave / WT;
    const int H = 6 * (p.K - 1);
    const int TT = W - 2 * H;
    // fused conv_post (p.wav): the tile's TT valid rows give TT - (PK - 1) output samples, so tiles step by that and start
    // (PK - 1) / 2 rows early
    constexpr int PK = 7, PH = (PK - 1) / 2;
    const int TTo = p.wav ? TT - 2 * PH : TT;

    // ---- persistent workgroups.  The valid tiles of the batch (ceil(len_b / TTo) per utterance) are numbered through, workgroup w
    // takes tiles w, w + G, w + 2G, ...: a table of the per-utterance tile counts' prefix sums lives in LDS.  While a tile is
    // computed the NEXT tile's residual stream is already on its way into registers, so the exposed HBM round trip and the
    // LDS transposition of a per-tile x load disappear from every tile but the workgroup's first.
    int* pre = (int*)(smem + p.pre_off);
    // zero the guard bands (once; the fused conv_post's fp32 output tile aliases them: again after every tile there)
    auto zero_guard_bands = [&](int t) {
        for (int idx = t; idx < 2 * RB_GUARD * (PITCH / 16); idx += THREADS) {
            const int r = idx / (PITCH / 16), c = idx % (PITCH / 16);
            const int row = r < RB_GUARD ? r : W + r;
            *(uint4*)(act + row * PITCH + c * 16) = make_uint4(0, 0, 0, 0);
            if constexpr (TB) *(uint4*)(act2 + row * PITCH + c * 16) = make_uint4(0, 0, 0, 0);
        }
    };
    zero_guard_bands(tid);
    int total = 0, j = blockIdx.x;
    if constexpr (PS) {
        for (int i = tid; i < p.B; i += THREADS) {
            const int l = p.lens ? p.lens[i] : p.T;
            pre[p.B + 1 + i] = (l + TTo - 1) / TTo;
            pre[2 * p.B + 1 + i] = l;                  // (the lengths too: no global load between two tiles)
        }
        __syncthreads();
        for (int i = tid; i <= p.B; i += THREADS) {
            int a = 0;
            for (int u = 0; u < i; ++u) a += pre[p.B + 1 + u];
            pre[i] = a;
        }
        __syncthreads();
        total = pre[p.B];
        if (j >= total) return;
    }
    const int G = gridDim.x;

    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    constexpr int RPP = THREADS / F4;              // rows one cooperative access of the workgroup covers (32 at C <= 64, 16 at C >= 128)
    constexpr int PER = SROWS / RPP;               // accesses per staging pass of SROWS = 32 rows per time-wave
    static_assert(THREADS % F4 == 0 && SROWS % RPP == 0, "row-coalesced staging");
    int c4 = tid % F4, r0 = tid / F4;   // r0 in [0, RPP)
    // staged row s = r0 + RPP * u of pass m <-> tile row: 32-row slab m of time-wave s / 32
    auto tile_row = [&](int m, int u) { const int sr = r0 + RPP * u; return ((sr >> 5) * MT + m) * 32 + (sr & 31); };

    // tile j -> (utterance, first output row): the utterance index only ever moves forward
    auto locate = [&](int jj, int& bb) {
        while (pre[bb + 1] <= jj) ++bb;
        bb = __builtin_amdgcn_readfirstlane(bb);
    };
    // the residual stream of a tile, fp32, straight into accumulator layout (lane & 31 = row, 4 consecutive channels per 16 B access).
    // Buffer loads over the utterance [0, len) x C return zeros for rows outside it (t < 0 wraps to a huge unsigned offset) = the zero padding.
    auto load_x = [&](f32x16 (&d)[NT], int m, int bb, int base, int ln) {   // 32-row slab m of this wave
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (long long)bb * p.T * C), 0, ln * C * 4, 0x00020000);
        const int o0 = ((base + wt * MT * 32 + (lane & 31)) * C + wc * NT * 32 + 4 * (lane >> 5)) * 4;
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                u32x4 v = u32x4{0u, 0u, 0u, 0u};
                if (!DTTS_DBG(p, 4)) v = __builtin_amdgcn_raw_buffer_load_b128(rs, o0 + (m * 32 * C + n * 32 + 8 * q) * 4, 0, RB_X_AUX);
                const f32x4 f = __builtin_bit_cast(f32x4, v);
#pragma unroll
                for (int e = 0; e < 4; ++e) d[n][4 * q + e] = f[e];
            }
    };
    // (readfirstlane: a length in a VGPR would put every buffer resource below in VGPRs: a waterfall loop around each buffer access)
    auto len_of = [&](int bb) { return __builtin_amdgcn_readfirstlane(pre[2 * p.B + 1 + bb]); };

    int b = 0, len, t0;
    if constexpr (PS) {
        locate(j, b);
        len = len_of(b);
        t0 = (j - pre[b]) * TTo - (p.wav ? PH : 0);
    } else {   // one tile per workgroup: grid (tiles, utterances)
        b = blockIdx.y;
        // (readfirstlane: hipcc loads lens[b] with a vector load — the kernel also stores through other pointers, so no scalar load)
        len = __builtin_amdgcn_readfirstlane(p.lens ? p.lens[b] : p.T);
        t0 = blockIdx.x * TTo - (p.wav ? PH : 0);
        if (t0 + (p.wav ? PH : 0) >= len) return;
    }
    f32x16 xr[MT][NT];
    if constexpr (PS) {
#pragma unroll
        for (int m = 0; m < MT; ++m) load_x(xr[m], m, b, t0 - H, len);
    } else {
        // one tile per workgroup (C = 32): coalesced whole-row loads, transposed into accumulator layout through the staging buffer
        // (the 32 B per row and instruction of the direct form cost 3 % at C = 32).  All MT*PER 16 B loads of a thread are issued
        // before the first LDS round trip: one exposed HBM latency per tile.
        const auto rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (long long)b * p.T * C), 0, len * C * 4, 0x00020000);
        const int g0 = ((t0 - H + r0) * C + c4 * 4) * 4;
        u32x4 ld[MT][PER];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                ld[m][u] = u32x4{0u, 0u, 0u, 0u};
                if (!DTTS_DBG(p, 4)) ld[m][u] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, g0 + (tile_row(m, u) - r0) * (C * 4), 0, RB_X_AUX);
            }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (m) __syncthreads();
#pragma unroll
            for (int u = 0; u < PER; ++u) *(u32x4*)(stage + (r0 + RPP * u) * EP + c4 * 16) = ld[m][u];
            __syncthreads();
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = *(const f32x4*)(stage + (wt * 32 + (lane & 31)) * EP + ((wc * NT + n) * 32 + 8 * q + 4 * (lane >> 5)) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) xr[m][n][4 * q + e] = v[e];
                }
        }
    }

    const int kg_stride = (C / 32) * 64;
    // work items of a workgroup: (tile, ResBlock r) — r runs over the launch's p.nrb ResBlocks on the SAME tile before the next tile
    int r = 0;

#ifdef RB_STAMP
    unsigned long long tq[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_last = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll 1
    for (;;) {
    if constexpr (PS) {
        // the thread index passes through an opaque move every tile: everything derived from it is recomputed per tile (a few VALU
        // instructions) instead of being hoisted out of the tile loop by hipcc and spilled to scratch for lack of registers
        tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        wt = wave % WT, wc = wave / WT;
        c4 = tid % F4, r0 = tid / F4;
    }
    const int xlane = (RB_GUARD + wt * MT * 32 + (lane & 31)) * PITCH + (lane >> 5) * 16;
    const size_t wlane = (size_t)(wc * NT) * 64 + lane;
    if constexpr (PS) r = __builtin_amdgcn_readfirstlane(r);
    else r = 0;                                      // (one tile per workgroup: one ResBlock per launch)
    const RBlockParams::Set& R = p.rb[PS ? r : 0];
    const int Kr = R.K;                              // this ResBlock's kernel size (the tile's halo follows the launch's largest)
    // C = 32 (two k-groups per tap): the packs' zero padding to a multiple of four steps would be 25 / 12.5 / 8 % of the MFMAs at k = 3 / 7 / 11:
    // those configurations run the real steps only (rb2_contract, uniform exit at tap boundaries)
    constexpr bool REAL_STEPS = (NKG < 4);
#ifdef RB_WINO_PROBE   // TIMING PROBE ONLY (wrong results): k = 3 at C >= 128 with 2/3 of the k-steps = the MFMA / LDS-read count of Winograd F(2,3)
    const int S = (C >= 128 && Kr == 3) ? 2 * NKG : (REAL_STEPS ? Kr : R.Kp) * NKG;
#else
    const int S = DTTS_DBG(p, 1) ? 0 : (REAL_STEPS ? Kr : R.Kp) * NKG;   // k-steps (packed taps are zero padded so that Kp * NKG % 4 == 0)
#endif
    const bool last_rb = !PS || r + 1 == p.nrb;
    // what the epilogue does with the stage sum: one ResBlock per launch: p.mode; all of the stage's: write, accumulate.., finish
    const int mode = (!PS || p.nrb == 1 || r == 0) ? p.mode : (last_rb ? p.last_mode : 1);
    const bool wav_now = p.wav && last_rb;           // (the launcher accepts p.wav only on a launch that ends its stage: last_mode 2)
    t0 = __builtin_amdgcn_readfirstlane(t0);
    const int base_t = t0 - H;  // global time of local row 0
    const long long brow = (long long)b * p.T;
    // the workgroup's next tile
    // the workgroup's next tile: static (j + G), or — p.tile_ctr — DYNAMIC: the next unclaimed tile of the launch, taken from a
    // device counter (tiles 0 .. G-1 are the workgroups' first tiles, the counter hands out G, G+1, ...).  Workgroups slowed down by
    // whatever shares their CU (the next batch's text->mel kernels on the other stream, a neighbour's phase) then simply take fewer
    // tiles instead of setting the launch's finish time.  One lane issues the atomic at the top of the tile; its result is not
    // needed before the epilogue (the x prefetch), where it is broadcast through LDS behind a barrier that exists anyway.
    unsigned claim = 0;
    if (PS && p.tile_ctr && tid == 0 && last_rb) claim = atomicAdd(p.tile_ctr, 1u);
    int jn = j + G;
    bool has_next = PS && jn < total;
    int bn = b, lenn = len, t0n = 0;
    auto plan_next = [&]() {
        if (!last_rb) {                              // the same tile again, for the stage's next ResBlock
            has_next = true;
            jn = j;
            bn = b;
            lenn = len;
            t0n = t0;
            return;
        }
        has_next = PS && jn < total;
        bn = b;
        if (has_next) {
            locate(jn, bn);
            lenn = len_of(bn);
            t0n = (jn - pre[bn]) * TTo - (p.wav ? PH : 0);
        }
    };
    if (!(PS && p.tile_ctr) || !last_rb) plan_next();
    // the stage sum: [B][T][C] like x — or, p.s_private (all ResBlocks in one launch WITH the fused conv_post, whose tiles overlap by
    // 2 PH rows): a private strip of TT rows per tile, so that no two workgroups read-modify-write the same rows
    const auto rs_s = p.s_private ? __builtin_amdgcn_make_buffer_rsrc((void*)p.S, 0, p.s_private, 0x00020000)
                                  : __builtin_amdgcn_make_buffer_rsrc((void*)(p.S + brow * C), 0, len * C * 4, 0x00020000);

    // bf16(leaky_relu(v + bias, 0.1)) of this wave's tiles -> LDS activation buffer, zero outside the utterance.
    // bias: this lane's 4 channel quads per co-tile, loaded into registers BEFORE the contraction it follows.
    auto load_bias = [&](f32x4 (&bb)[NT][4], const float* bias) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q) bb[n][q] = *(const f32x4*)(bias + (wc * NT + n) * 32 + 8 * q + 4 * (lane >> 5));
    };
    const bool all_inb = base_t >= 0 && base_t + W <= len;   // block-uniform: no row of the tile needs masking
    int n_ovf = 0;
    // MASKED = false: a tile wholly inside its utterance (block-uniform, most tiles) — no row needs the zero select: 2 of the ~11 VALU
    // instructions per four values less, in the phase that is VALU-bound (LABNOTES round 4 (C))
    auto write_act_impl = [&](char* dst, const f32x16 (&v)[MT][NT], auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int row = (wt * MT + m) * 32 + (lane & 31);
            const int t = base_t + row;
            const bool inb = !MASKED || (t >= 0 && t < len);
            // range guard: only the rows this tile OUTPUTS are counted.  Every in-utterance row is an output row of exactly one tile
            // and carries the exact activation there at each of the six stages, so the count is a census; halo rows (recomputed,
            // increasingly inexact towards the tile edge, their results discarded) are another tile's output rows.
            const bool counted = inb && row >= H && row < H + TT;
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co = (wc * NT + n) * 32 + 8 * q + 4 * (lane >> 5);
                    f32x4 v4 = {v[m][n][4 * q], v[m][n][4 * q + 1], v[m][n][4 * q + 2], v[m][n][4 * q + 3]};
#if defined(RB_WINO_PROBE) && RB_WINO_PROBE >= 2   // (the output transform: one fp32 add per value, as two packed adds per four)
                    if constexpr (C >= 128) {
                        f32x2_t lo2 = {v4[0], v4[1]}, hi2 = {v4[2], v4[3]};
                        asm volatile("v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %2" : "+v"(lo2), "+v"(hi2) : "v"(f32x2_t{0.f, 0.f}));
                        v4 = f32x4{lo2[0], lo2[1], hi2[0], hi2[1]};
                    }
#endif
                    uint2 pk = act4<EL>(v4, 0.1f);
                    if constexpr (GUARD) n_ovf += counted ? ovf4(v4, 0.1f) : 0;
                    if constexpr (MASKED) {
                        if (!inb) pk = make_uint2(0, 0);
                    }
                    *(uint2*)(dst + (RB_GUARD + row) * PITCH + co * 2) = pk;
                }
        }
    };
    auto write_act = [&](char* dst, const f32x16 (&v)[MT][NT]) {
        if (DTTS_DBG(p, 8)) return;
        if (all_inb) write_act_impl(dst, v, std::false_type{});
        else write_act_impl(dst, v, std::true_type{});
    };

    uint4 ring[4][NT];
    f32x4 bb[NT][4];   // one live bias set
    RB_T(10);                                            // (tile bookkeeping, and — the first tile — the wait for x)
    rb_preload<NT>(ring, R.w1[0] + wlane, kg_stride);   // in flight during the first activation write
    load_bias(bb, R.b1[0]);
    write_act(act, xr);
    RB_T(0);
    __syncthreads();
    RB_T(1);

    f32x16 acc[MT][NT];
#pragma unroll 1
    for (int it = 0; it < 3; ++it) {
        // conv1: the first MFMA of every tile takes the bias pattern as its C operand (no accumulator init pass)
        f32x16 cinit[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) cinit[n][4 * q + e] = bb[n][q][e];
        load_bias(bb, R.b2[it]);       // lands while conv1 runs
        const int d = R.dil[it];
        if constexpr (REAL_STEPS) {
            if (S) rb2_contract<EL, MT, NT, NKG, PITCH, 4, true, XA1_32>(acc, ring, act, xlane - ((Kr - 1) / 2) * d * PITCH, R.w1[it] + wlane, S, d * PITCH, cinit);
            else rb_contract<EL, MT, NT, NKG, PITCH, true, 1, XA1>(acc, ring, act, 0, R.w1[it] + wlane, 0, 0, kg_stride, &cinit);
        } else
            rb_contract<EL, MT, NT, NKG, PITCH, true, 1, XA1>(acc, ring, act, xlane - ((Kr - 1) / 2) * d * PITCH, R.w1[it] + wlane, S, d * PITCH, kg_stride, &cinit);
        rb_preload<NT>(ring, R.w2[it] + wlane, kg_stride);   // next conv's first weights fly during barrier + write
        RB_T(2);
        if constexpr (!TB) __syncthreads();   // every wave is done reading A (TB: xt has its own buffer, last read before the previous barrier)
        RB_T(3);
        write_act(act2, acc);          // xt (16-bit, activated): overwrites A, or goes to its own buffer
        RB_T(4);
        __syncthreads();
        RB_T(5);
        // conv2 accumulates straight into the residual registers: x = x + b2 + W2 * xt
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) xr[m][n][4 * q + e] += bb[n][q][e];
        if (it < 2) load_bias(bb, R.b1[it + 1]);
        if constexpr (REAL_STEPS) {
            if (S) rb2_contract<EL, MT, NT, NKG, PITCH, 4, false, XA1_32>(xr, ring, act2, xlane - ((Kr - 1) / 2) * PITCH, R.w2[it] + wlane, S, PITCH, cinit);
        } else
            rb_contract<EL, MT, NT, NKG, PITCH, false, 1, XA1>(xr, ring, act2, xlane - ((Kr - 1) / 2) * PITCH, R.w2[it] + wlane, S, PITCH, kg_stride);
        if (it < 2) rb_preload<NT>(ring, R.w1[it + 1] + wlane, kg_stride);
        if (PS && p.tile_ctr && last_rb && it == 2 && tid == 0) pre[3 * p.B + 1] = G + (int)claim;   // the claimed tile, for everyone (read behind the barrier)
        RB_T(6);
        if (!TB || it == 2) __syncthreads();   // every wave is done reading xt (TB: A is rewritten, not xt; the barrier stays in front of the epilogue)
        RB_T(7);
        if (it < 2) {
            write_act(act, xr);
            RB_T(0);
            __syncthreads();
            RB_T(1);
        }
    }

    if (PS && p.tile_ctr && last_rb) {
        jn = __builtin_amdgcn_readfirstlane(pre[3 * p.B + 1]);
        plan_next();
    }
    if (DTTS_DBG(p, 2)) {
        if (xr[0][0][0] == 123.456f) p.S[0] = 1.f;
        if (has_next) {
#pragma unroll
            for (int m = 0; m < MT; ++m) load_x(xr[m], m, bn, t0n - H, lenn);
        }
    } else {
    // ---- epilogue: rows [H, H+TT) of the tile leave as whole rows through the fp32 staging buffer; the old
    // accumulator values (xs += ...) of all MT passes are fetched up front.  Buffer ops: rows >= len are dropped by the
    // range check, halo rows are sent out of range explicitly.
    const auto rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Sa ? p.Sa + brow * C : (unsigned short*)(p.S + brow * C)), 0,
                                                        len * C * 2, 0x00020000);
    // WAVE-PRIVATE transposition: each wave stages its own 32 rows x its own CW channels and reads the same block back as whole 128-byte
    // row segments (8 NT lanes per row): no workgroup barrier inside the epilogue (round 3: 2 MT - 1 of them), the waves drift through their
    // slabs and HBM round trips independently.  (The LDS serves one wave's accesses in order.)
    constexpr int CW = NT * 32, F4W = CW / 4, RPI = 64 / F4W, NRD = 32 / RPI;
    const int er = lane / F4W, ec = lane % F4W;                          // lane octet (one row per octet and access), 16-byte chunk of the wave's row segment
    const int goffw = (p.s_private ? (j * TT - H) * C : base_t * C) * 4 + (wc * CW + ec * 4) * 4;
    // Which of the slab's 32 rows an octet reads in access u.  Consecutive rows (u * 8 + er: rounds 3 - 5) put the 16-lane groups of a ds_read_b128
    // ({0-3, 12-15, 20-27}, ...) on overlapping 16-byte slots of the 256-byte bank row — 3-way at C >= 64, 2-way at C = 32: 12 / 8 LDS cycles per read instead
    // of 4, a quarter of the kernels' bank-conflict cycles (round 6: tools/lds_conflict_attrib.sh; the model reproduces the counter to 1 %).  Rows
    // base + {0, 16, 8, 24} for the four octets of a half-wave (base = 2 u + half) spread every group over all 16 slots for each row pitch in use (9 / 17 / 33 / 65
    // slots): conflict-free; an octet still moves one whole 128-byte row segment, so the global accesses coalesce as before.  Same values, same order per row.
    static_assert(F4W == 8 && RPI == 8 && NRD == 4, "the conflict-free row order below is for 8 lanes per row, 8 rows per access");
    // (C = 32 keeps consecutive rows: a row is 128 bytes there, so an access of 8 consecutive rows is ONE contiguous kilobyte of the stage sum — worth more than
    // the 4 LDS cycles: +1.1 % on both C = 32 launches with the permuted order, same box)
    const int erow_slab_base = C == 32 ? er : (er >> 2) + ((er & 1) ? 16 : 0) + ((er & 2) ? 8 : 0);
    auto srow = [&](int u) { return (C == 32 ? 8 : 2) * u + erow_slab_base; };                // row of the 32-row slab
    auto erow = [&](int m, int u) { return (wt * MT + m) * 32 + srow(u); };                   // local tile row of (slab m, access u)
    auto eoff = [&](int m, int u) {
        const int row = erow(m, u);
        return (row >= H && row < H + TT) ? goffw + row * (C * 4) : (int)0x80000000;
    };
    u32x4 sold[MT][NRD];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int u = 0; u < NRD; ++u) {
            sold[m][u] = u32x4{0u, 0u, 0u, 0u};
            // (all ResBlocks in one launch: the sum was written by THIS workgroup a moment ago and sits in L2 / Infinity Cache: a cached read)
            if (mode >= 1) sold[m][u] = p.nrb > 1 ? __builtin_amdgcn_raw_buffer_load_b128(rs_s, eoff(m, u), 0, 0)
                                                  : __builtin_amdgcn_raw_buffer_load_b128(rs_s, eoff(m, u), 0, VP_LD_AUX);
        }
    RB_T(12);                                            // (epilogue: plan + stage-sum loads issued)
    // fused conv_post: the stage output leaky_relu(xs / num_kernels) stays in LDS as an fp32 tile ([TT rows][C], rows outside
    // the utterance zero = conv_post's zero padding) instead of going to HBM; the transposition buffer moves behind it
    constexpr int OP = C * 4;                      // otile row pitch (bytes)
    char* otile = smem;
    // (TB: both activation buffers are dead in the epilogue; the staging rows follow the output tile inside them)
    char* estage = wav_now ? smem + ((TB || (size_t)TT * OP > ACT_BYTES) ? (size_t)TT * OP : ACT_BYTES) : stage;
    char* stg = estage + (wt * 32) * EP + (wc * CW) * 4;                 // this wave's block of the staging buffer
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = xr[m][n][4 * q + e];
                *(f32x4*)(stg + (lane & 31) * EP + (n * 32 + 8 * q + 4 * (lane >> 5)) * 4) = v;
            }
        // slab m of the residual registers is free: the NEXT tile's slab m starts its trip into them (no second register set, and
        // the loads are younger than the stage sum fetched above, so nothing below waits for them)
        if (has_next) load_x(xr[m], m, bn, t0n - H, lenn);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        RB_T(13);                                        // (slab: accumulators -> staging rows, next tile's loads issued)
#pragma unroll
        for (int u = 0; u < NRD; ++u) {
            const int off = eoff(m, u);
            f32x4 o = *(const f32x4*)(stg + srow(u) * EP + ec * 16);
            o += __builtin_bit_cast(f32x4, sold[m][u]);                // xs += resblock(x)  (hifigan.py:133-135); zeros in mode 0
            if (wav_now) {
                const int row = erow(m, u);                            // local tile row
                if (row >= H && row < H + TT) {
                    const int t = base_t + row;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (t >= 0 && t < len) ? lrelu(o[e] / p.div, p.slope) : 0.f;
                    *(f32x4*)(otile + (size_t)(row - H) * OP + (wc * CW + ec * 4) * 4) = o;
                }
                continue;
            }
            if (mode == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = o[e] / p.div;
            }
            if (!(mode == 2 && p.Sa && p.drop_S)) {   // the stage's consumers read only the bf16 copy: the fp32 sum can stay unwritten
                // an intermediate sum of a fused launch is re-read by this workgroup's next ResBlock: write-back cached; final results stream out
                if (p.nrb > 1 && !last_rb) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs_s, off, 0, 0);
                else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs_s, off, 0, VP_ST_AUX);
            }
            if (mode == 2 && p.Sa) {
                typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
                const u32x2 pk = {pack2bf(lrelu(o[0], p.slope), lrelu(o[1], p.slope)), pack2bf(lrelu(o[2], p.slope), lrelu(o[3], p.slope))};
                __builtin_amdgcn_raw_buffer_store_b64(pk, rs_a, off == (int)0x80000000 ? off : off >> 1, 0, VP_ST_AUX);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                               // slab m + 1 reuses this wave's staging block
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        RB_T(14);                                        // (slab: rows read back, + stage sum, stored)
    }
    if constexpr (C == 32) if (wav_now) {   // (the launcher rejects p.wav for other widths)
        // ---- wav[t] = tanh(b + sum_{tap, c} w[c][tap] * otile[t + tap - 3][c])   (conv_post + tanh, hifigan.py:139-141) in exact fp32:
        // 8 lanes per output sample (4 channels each, 7 taps), partial sums joined by three xor-shuffles
        __syncthreads();
        const int q8 = tid & 7, rr = tid >> 3;                         // channel quad, row within a pass of THREADS / 8 rows
        f32x4 wq[PK];
#pragma unroll
        for (int k = 0; k < PK; ++k) wq[k] = *(const f32x4*)(p.post_w + k * C + q8 * 4);
        const float pb = p.post_b[0];
        float* wb = p.wav + brow;
        // each 8-lane group slides over PR consecutive outputs (PR + PK - 1 row reads instead of PR * PK; PR odd: neighbouring groups start
        // 128 B apart modulo the 256 B of the LDS banks); every output is still summed tap by tap in the order of the one-output form
        constexpr int PR = 3;
        for (int o0 = 0; o0 < TTo; o0 += (THREADS / 8) * PR) {
            const int ob = o0 + rr * PR;                               // first output of this group: otile rows ob .. ob + PR + PK - 2
            float a[PR];
#pragma unroll
            for (int i = 0; i < PR; ++i) a[i] = 0.f;
            if (ob < TTo) {
#pragma unroll
                for (int j = 0; j < PR + PK - 1; ++j) {
                    const int row = ob + j < TTo + PK - 1 ? ob + j : TTo + PK - 2;   // (rows past the tile feed discarded outputs only)
                    const f32x4 v = *(const f32x4*)(otile + (size_t)row * OP + q8 * 16);
#pragma unroll
                    for (int i = 0; i < PR; ++i) {
                        const int k = j - i;
                        if (k >= 0 && k < PK) {   // explicit operations: the same rounding sequence for every output, whatever its place in a group / tile
                            const float d = __builtin_fmaf(v[3], wq[k][3], __builtin_fmaf(v[2], wq[k][2], __builtin_fmaf(v[1], wq[k][1], __fmul_rn(v[0], wq[k][0]))));
                            a[i] = __fadd_rn(a[i], d);
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < PR; ++i) {
                a[i] += __shfl_xor(a[i], 1, 64);
                a[i] += __shfl_xor(a[i], 2, 64);
                a[i] += __shfl_xor(a[i], 4, 64);
            }
            // ALWAYS-ON overflow detector (every instantiation, every call): an fp16 operand that overflowed anywhere upstream is +-inf, every
            // sum it enters is inf / NaN from there on (the fp32 residual stream never recovers), so it arrives HERE as a non-finite
            // pre-tanh value.  tanh would turn +-inf into a plausible +-1: the sample is poisoned with NaN instead and counted.
            // tanh(x) = 1 - 2 / (e^{2x} + 1) on the hardware exp2 / rcp, evaluated by every lane (libm's tanhf ran its ~45 instructions for
            // the one live lane in eight): absolute error <= 3e-7 (1 / 100 of an int16 step), saturates correctly at +-1.
            float pre[PR], th[PR];
            int nf = 0;   // non-finite SAMPLES of this group (the same unit as vconv.hip's post_tanh detector: dtts_vocoder_nonfinite counts samples)
#pragma unroll
            for (int i = 0; i < PR; ++i) {
                pre[i] = a[i] + pb;
                th[i] = __builtin_fmaf(-2.f, __builtin_amdgcn_rcpf(__fadd_rn(__builtin_amdgcn_exp2f(pre[i] * 2.885390081777927f), 1.f)), 1.f);   // (2 log2 e)
            }
            if (q8 == 0) {
#pragma unroll
                for (int i = 0; i < PR; ++i) {
                    const int o = ob + i, t = t0 + PH + o;
                    const bool nonfin = !(__builtin_fabsf(pre[i]) <= 3.0e38f);
                    if (o < TTo && t < len) {
                        wb[t] = nonfin ? __builtin_nanf("") : th[i];
                        nf += nonfin ? 1 : 0;
                    }
                }
                if (nf && p.bad) atomicAdd(p.bad, (unsigned)nf);   // (never on a healthy call)
            }
        }
    }
    }   // (epilogue)
    RB_T(8);                                             // epilogue (+ the fused conv_post)
#ifdef RB_STAMP
    tq[11] += 1;
#endif
    if constexpr (GUARD) {
        if (n_ovf) atomicAdd(p.ovf, (unsigned long long)n_ovf);
    }
    if (!has_next) break;
    r = last_rb ? 0 : r + 1;
    if (wav_now) {
        // the fp32 output tile aliases the activation buffer AND its guard bands: conv_post's reads are over behind this barrier, then
        // the bands are zero again before the next tile's first convolution reads them (the barrier after its first write_act orders
        // both).  Without it a workgroup's 2nd+ tile ran its outermost halo rows on fp32 bit patterns read as 16-bit operands: the
        // results of those rows are discarded (H covers the six receptive fields), but the range guard counted their Inf / huge values.
        __syncthreads();
        zero_guard_bands(threadIdx.x);
    }
    j = jn;
    b = bn;
    len = lenn;
    t0 = t0n;
    }   // (tiles of this workgroup)
#ifdef RB_STAMP
    {
        const int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0 && (w == 0 || w == WT * WC - 1)) {
            const int slot = (C == 32 ? 0 : C == 64 ? 1 : C == 128 ? 2 : 3) * 3 + (p.K == 3 ? 0 : p.K == 7 ? 1 : 2);
            for (int k = 0; k < 16; ++k) atomicAdd(&rb_stamps[w == 0 ? 0 : 1][slot][k], tq[k]);
        }
    }
#endif
}

template <int C, int MT, int NT, int WT, int WC, int EL, int PS, bool GUARD = false, bool TB = false>
static hipError_t rb_launch_cfg(const RBlockParams& p, hipStream_t stream) {
    constexpr int W = 32 * MT * WT, PITCH = C * 2 + 16, EP = C * 4 + 16;
    const int H = 6 * (p.K - 1), TT = W - 2 * H;
    if (TT < 32) return hipErrorInvalidValue;
    if ((long long)p.T * C * 4 >= (1LL << 31)) return hipErrorInvalidValue;   // 32-bit byte offsets inside an utterance's buffer resource
    constexpr size_t ACT = (size_t)(W + 2 * RB_GUARD) * PITCH;
    size_t lds = TB ? std::max(2 * ACT, ACT + (size_t)RB_GUARD * PITCH + (size_t)WT * 32 * EP) : ACT + (size_t)WT * 32 * EP;   // TB: the staging rows lie over the xt buffer
    int TTo = TT;
    if (p.wav) {   // fused conv_post (7 taps): the fp32 output tile may be larger than the activation tile it replaces
        if (C != 32 || (p.nrb == 1 && p.mode != 2) || !p.post_w || !p.post_b) return hipErrorInvalidValue;
        lds = TB ? std::max(2 * ACT, (size_t)TT * C * 4 + (size_t)WT * 32 * EP) : std::max((size_t)TT * C * 4, ACT) + (size_t)WT * 32 * EP;
        TTo = TT - 6;
    }
    RBlockParams q = p;
    q.pre_off = (int)lds;                          // tile table: prefix sums [B + 1], counts [B], lengths [B]
    if (PS) lds += (size_t)(3 * p.B + 2) * sizeof(int);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if constexpr (EL == EL_F16 && !GUARD) {
        if (p.ovf) return rb_launch_cfg<C, MT, NT, WT, WC, EL, PS, true, TB>(p, stream);
    }
    auto kern = rblock_kernel<C, MT, NT, WT, WC, EL, PS, GUARD, TB>;
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
    constexpr int THREADS = 64 * WT * WC;
    if (!PS) {
        hipLaunchKernelGGL(kern, dim3((p.T + TTo - 1) / TTo, p.B), dim3(THREADS), lds, stream, q);
        return hipGetLastError();
    }
    // persistent workgroups: as many as are resident at once (LDS / thread limits), never more than there can be tiles
    static int cus_dev[64] = {};
    int& cus = cus_dev[cur_dev & 63];
    if (!cus) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, cur_dev) != hipSuccess) return hipErrorInvalidDevice;
        cus = prop.multiProcessorCount;
    }
    const int per_cu = std::max(1, std::min({(int)(160 * 1024 / lds), 2048 / THREADS, THREADS <= 256 ? 2 : 1}));
    const long long max_tiles = (long long)p.B * ((p.T + TTo - 1) / TTo);
    const int grid = (int)std::min<long long>((long long)std::max(1, cus - cu_reserve()) * per_cu, max_tiles);
    if (grid <= 0) return hipSuccess;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), lds, stream, q);
    return hipGetLastError();
}

// whole-ResBlock fusion pays where the halo 6 (k - 1) is small against the tile that fits: every k at C <= 64 (512-row tiles), k = 3
// at C = 128 (256 rows) and C = 256 (128 rows); the larger kernels of the wide stages run per iteration (vpair.hip)
bool rblock_supported(int C, int K) {
    if (!(K & 1) || K < 3 || K > 11) return false;
    return C == 32 || C == 64 || ((C == 128 || C == 256) && K == 3);
}

long long rblock_private_rows(int C, int Kmax, int B, int T) {
    if (C != 32) return 0;
    const int W = 1024, TT = W - 12 * (Kmax - 1), TTo = TT - 6;    // rb_launch_cfg<32, 4, 1, 8, 1, ., 1> with the fused conv_post
    if (TTo < 32) return 0;
    return (long long)B * ((T + TTo - 1) / TTo) * TT;
}

int rblock_padded_taps(int C, int K) {
    const int nkg = C / 16;
    int kp = K;
    while ((kp * nkg) % 4) ++kp;
    return kp;
}

// Persistent workgroups everywhere but at C = 32 (same box: C = 64 -5 %, C = 128 -5 %, C = 256 -17 %; C = 32, two 4-wave workgroups
// per CU, +1 %: that one keeps one tile per workgroup).
template <int EL>
static hipError_t rb_launch_el(const RBlockParams& p, int C, hipStream_t stream) {
    // C = 32: k >= 7 on 1024-row tiles (8 waves over time, one persistent workgroup per CU: the 6 (k - 1)-row halo costs 12 % of a
    // k = 11 tile instead of 23 %; -8 % on that launch), k = 3 on 512-row tiles, two 4-wave workgroups per CU, one tile each (the
    // 1024-row form is 14 % slower there).  DTTS_RB32=0: 512-row tiles for every k (round 2).
    static const int rb32 = ablate_env("DTTS_RB32") ? atoi(ablate_env("DTTS_RB32")) : 1;
    // small batches (B = 1: one sentence): when the default tiles leave more than half of the CUs without one, the launch takes as long
    // as ONE tile -> half-size tiles (more halo recomputed, but twice the CUs at work)
    static const bool small_ok = [] { const char* e = ablate_env("DTTS_RB_SMALL"); return !e || atoi(e) != 0; }();
    static int cus_dev[64] = {};
    int cur_dev = 0;
    (void)hipGetDevice(&cur_dev);
    int& cus = cus_dev[cur_dev & 63];
    if (!cus) {
        hipDeviceProp_t prop;
        cus = hipGetDeviceProperties(&prop, cur_dev) == hipSuccess ? prop.multiProcessorCount : 256;
    }
    auto few = [&](int W) {   // tiles of W rows (valid: W - 12 (k - 1), the fused conv_post 6 less): at most half the CUs get one
        const int tt = W - 12 * (p.K - 1) - (p.wav ? 6 : 0);
        return small_ok && tt >= 32 && 2 * (long long)p.B * ((p.T + tt - 1) / tt) <= cus;
    };
    if (p.nrb < 1 || p.nrb > 3) return hipErrorInvalidValue;
    if (p.nrb > 1) {   // every ResBlock of the stage in one launch: the persistent full-size configurations only
        if (C == 32) return rb_launch_cfg<32, 4, 1, 8, 1, EL, 1>(p, stream);
        if (C == 64) return rb_launch_cfg<64, 4, 1, 4, 2, EL, 1>(p, stream);
        return hipErrorInvalidValue;
    }
    // (experiment, tune bit 7) C = 32 without the fused conv_post: two phase-shifted groups per workgroup (rblock2.hip)
#ifdef DTTS_ABLATE   // (rblock2.hip is compiled into the ablation library only)
    if (p.pingpong && rblock2_supported(C, p.K, p.wav != nullptr) && !few(512)) return rblock2_launch(p, C, stream);
#endif
#if defined(RB_TB32)   // experiment: two activation buffers at C = 32, k >= 7: 768-row tiles (MT = 3), two barriers per iteration instead of four
    if (C == 32 && rb32 && p.K >= 7 && !few(768)) return rb_launch_cfg<32, 3, 1, 8, 1, EL, 1, false, true>(p, stream);
#endif
#if defined(RB_X32) && RB_X32 == 1   // experiment: 12 waves (3 per SIMD) over 1152 rows at C = 32, k >= 7
    if (C == 32 && rb32 && p.K >= 7 && !few(1152)) {   // (with the fused conv_post the 1152-row output tile does not fit the LDS: falls through)
        const hipError_t e = rb_launch_cfg<32, 3, 1, 12, 1, EL, 1>(p, stream);
        if (e != hipErrorInvalidValue) return e;
    }
#elif defined(RB_X32) && RB_X32 == 2 // experiment: 12 waves (3 per SIMD) over 768 rows, MT = 2
    if (C == 32 && rb32 && p.K >= 7 && !few(768)) return rb_launch_cfg<32, 2, 1, 12, 1, EL, 1>(p, stream);
#endif
    if (C == 32 && rb32 && p.K >= 7 && !few(1024)) return rb_launch_cfg<32, 4, 1, 8, 1, EL, 1>(p, stream);
    if (C == 64 && few(512)) return rb_launch_cfg<64, 4, 1, 2, 2, EL, 1>(p, stream);     // 256-row tile, 4 waves
    if (C == 128 && few(256)) return rb_launch_cfg<128, 4, 1, 1, 4, EL, 1>(p, stream);   // 128-row tile, 4 waves
    if (C == 256 && few(128)) return rb_launch_cfg<256, 2, 1, 1, 8, EL, 1>(p, stream);   // 64-row tile
    if (C == 32) return rb_launch_cfg<32, 4, 1, 4, 1, EL, 0>(p, stream);      // 512-row tile, 4 waves over time
    // C = 64, k >= 7: 640-row tiles (MT = 5; the halo 12 (k - 1) is 11 / 19 % of the tile instead of 14 / 23 %: -4.8 % at k = 11, nothing at k = 3
    // where 13 spilled registers cost what the halo gives); tune bit 14: 512-row tiles for every k (round 3)
#if defined(RB_X64) && RB_X64 <= 2   // experiment: 12 waves (3 per SIMD: 6 time x 2 channel), 576-row tiles, 168 registers: RB_X64 = 1 for k >= 7, 2 for every k
    if (C == 64 && (p.K >= 7 || RB_X64 >= 2) && !p.small_tile) return rb_launch_cfg<64, 3, 1, 6, 2, EL, 1>(p, stream);
#elif defined(RB_X64)                // experiment: 16 waves (4 per SIMD: 8 time x 2 channel), 512-row tiles, 128 registers, every k
    if (C == 64 && !p.small_tile) return rb_launch_cfg<64, 2, 1, 8, 2, EL, 1>(p, stream);
#endif
#if defined(RB_X128)  // experiment: 12 waves (3 time x 4 channel), 288-row tiles at C = 128 (k = 3)
    if (C == 128) return rb_launch_cfg<128, 3, 1, 3, 4, EL, 1>(p, stream);
#endif
    if (C == 64 && p.K >= 7 && !p.small_tile) return rb_launch_cfg<64, 5, 1, 4, 2, EL, 1>(p, stream);
    if (C == 64) return rb_launch_cfg<64, 4, 1, 4, 2, EL, 1>(p, stream);      // 512-row tile, 8 waves (4 time x 2 channel)
    if (C == 128) return rb_launch_cfg<128, 4, 1, 2, 4, EL, 1>(p, stream);    // 256-row tile, 8 waves (2 time x 4 channel)
    if (C == 256) return rb_launch_cfg<256, 4, 1, 1, 8, EL, 1>(p, stream);    // 128-row tile, 8 waves over channels
    return hipErrorInvalidValue;
}

hipError_t rblock_launch(const RBlockParams& p, int C, hipStream_t stream) {
    return p.el == EL_F16 ? rb_launch_el<EL_F16>(p, C, stream) : rb_launch_el<EL_BF16>(p, C, stream);
}

} // namespace dtts
