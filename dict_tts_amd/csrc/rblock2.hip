// Fused HifiGAN ResBlock1 (modules/hifigan/hifigan.py:27-58), PHASE-SHIFTED two-group form ("ping-pong") for the narrow stages.
//
// rblock.hip keeps one time tile per workgroup resident for all six convolutions; its eight waves run in barrier lockstep, so the two
// waves of every SIMD contend for the matrix pipe during a contraction and leave it idle during every activation rewrite, x load and
// epilogue (MFMA busy 27-46 % at C <= 64, profiles/r03_f_pmc_util.md).  The matrix pipe is per SIMD and a partner wave's MFMAs come
// straight out of one's own stream, while VALU / LDS / VMEM work beside another wave's MFMAs is nearly free (MI355X_MICROARCH.md,
// "Two waves per SIMD").  Here a workgroup is TWO groups of GW waves (one wave of each group per SIMD), each group with its OWN tile,
// its own LDS region and its own residual registers, running the same phase sequence
//     N0 [epilogue of the previous tile + x of the next one -> activation tile]  M1 [conv1]  N2 [rewrite]  M3 [conv2]  N4 ...  M11
// with every phase closed by ONE workgroup barrier — and group 1 started one barrier later.  So whenever group 0 is in a matrix phase
// (M) group 1 is in a memory / VALU phase (N) and vice versa: each SIMD's matrix pipe always belongs to exactly one wave, and the
// rewrites, the epilogue's HBM round trips and the next tile's x load hide under the partner's contraction.  12 barriers per tile
// (rblock.hip: 12 + 2 MT - 1), none inside a phase: the epilogue transposes each wave's accumulators through a wave-private LDS
// region (its own rows of the dead activation tile).
// Tiles are handed out in PAIRS (2q, 2q + 1 -> group 0, 1): both groups run the same number of rounds, an odd last tile's partner
// runs a zero-length dummy (buffer loads return zeros, stores are dropped).
#include "rblock.h"
#include "rb_common.h"

#include <algorithm>

namespace dtts {

template <int C, int MT, int NT, int WT, int WC, int EL, bool GUARD>
__global__ __launch_bounds__(128 * WT * WC, 1) void rblock2_kernel(const RBlockParams p) {
    static_assert(WC * NT * 32 == C, "channel tiling must cover C");
    static_assert(WC == 1, "the wave-private epilogue staging lives in the wave's own activation rows: one wave per row strip");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int GW = WT * WC;                    // waves per group
    constexpr int GT = 64 * GW;                    // threads per group
    constexpr int W = 32 * MT * WT;                // rows of a group's tile
    constexpr int PITCH = C * 2 + 16;
    constexpr int NKG = C / 16;
    constexpr int CW = NT * 32;                    // channels a wave owns
    constexpr int EPW = CW * 4 + 16;               // fp32 staging row of one wave
    constexpr int F4W = CW / 4;                    // 16-byte chunks per staged row
    constexpr int RPI = 64 / F4W;                  // rows one wave-wide 16 B/lane access covers
    constexpr int NRD = 32 / RPI;                  // accesses per 32-row slab
    constexpr size_t GSZ = (size_t)(W + 2 * RB_GUARD) * PITCH;
    static_assert((size_t)MT * 32 * PITCH >= (size_t)32 * EPW, "a wave's rows must hold its 32-row fp32 staging slab");

    int tid = threadIdx.x % GT, lane = tid & 63, wave = tid >> 6;
    const int grp = __builtin_amdgcn_readfirstlane((int)threadIdx.x / GT);
    char* act = smem + (size_t)grp * GSZ;
    int wt = wave % WT, wc = wave / WT;
    const int H = 6 * (p.K - 1);
    const int TT = W - 2 * H;

    int* pre = (int*)(smem + p.pre_off);
    for (int idx = tid; idx < 2 * RB_GUARD * (PITCH / 16); idx += GT) {   // zero this group's guard bands (never written again)
        const int r = idx / (PITCH / 16), c = idx % (PITCH / 16);
        const int row = r < RB_GUARD ? r : W + r;
        *(uint4*)(act + row * PITCH + c * 16) = make_uint4(0, 0, 0, 0);
    }
    for (int i = threadIdx.x; i < p.B; i += 2 * GT) {
        const int l = p.lens ? p.lens[i] : p.T;
        pre[p.B + 1 + i] = (l + TT - 1) / TT;
        pre[2 * p.B + 1 + i] = l;
    }
    __syncthreads();
    for (int i = threadIdx.x; i <= p.B; i += 2 * GT) {
        int a = 0;
        for (int u = 0; u < i; ++u) a += pre[p.B + 1 + u];
        pre[i] = a;
    }
    __syncthreads();
    const int total = pre[p.B];
    const int npairs = (total + 1) >> 1;
    int q = blockIdx.x;                            // this round's pair of tiles
    if (q >= npairs) return;                       // (workgroup-uniform)
    const int G = gridDim.x;

    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    // tile j of the launch -> (utterance, first output row, length); j >= total: the zero-length dummy partner of an odd last tile
    int b = 0, len = 0, t0 = 0;
    auto place = [&](int jj, int& bb, int& ln, int& tt) {
        if (jj < total) {
            while (pre[bb + 1] <= jj) ++bb;        // (the utterance index only moves forward)
            bb = __builtin_amdgcn_readfirstlane(bb);
            ln = __builtin_amdgcn_readfirstlane(pre[2 * p.B + 1 + bb]);
            tt = __builtin_amdgcn_readfirstlane((jj - pre[bb]) * TT);
        } else {
            ln = 0;
            tt = 0;
        }
    };
    // the residual stream of a tile, fp32, straight into accumulator layout (rblock.hip: load_x); rows outside [0, len) arrive as zeros
    auto load_x = [&](f32x16 (&d)[NT], int m, int bb, int base, int ln) {
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (long long)bb * p.T * C), 0, ln * C * 4, 0x00020000);
        const int o0 = ((base + wt * MT * 32 + (lane & 31)) * C + wc * CW + 4 * (lane >> 5)) * 4;
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, o0 + (m * 32 * C + n * 32 + 8 * qq) * 4, 0, RB_X_AUX);
                const f32x4 f = __builtin_bit_cast(f32x4, v);
#pragma unroll
                for (int e = 0; e < 4; ++e) d[n][4 * qq + e] = f[e];
            }
    };

    place(2 * q + grp, b, len, t0);
    f32x16 xr[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) load_x(xr[m], m, b, t0 - H, len);

    const int kg_stride = (C / 32) * 64;
    const int S = p.K * NKG;                       // real k-steps (the packs' zero padding to a multiple of four steps is skipped)
    constexpr int RD = 8;                          // weight ring: fragments run 7 steps ahead
    // group 1 runs HALF A ROUND (7 of the 14 slots) behind group 0 from here on: its matrix phases fall on group 0's memory phases and
    // each group's long memory span (epilogue | next x | first rewrite) lies beside the partner's M - N - M
    if (grp == 1) {
#pragma unroll 1
        for (int i = 0; i < 7; ++i) __syncthreads();
    }

#ifdef DTTS_ABLATE
    // per-phase cycle stamps of each group's wave 0 (DTTS_RB_STATS=1): slots 0 write_act(x) 1 barrier-after-N 2 conv1 3 barrier-after-M
    // 4 rewrite(xt) 5 conv2 6 rewrite(x) 7 epilogue 8 tiles
    unsigned long long tsum[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
#define RB2_T0() tlast = __builtin_amdgcn_s_memtime()
#define RB2_T(slot) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tsum[slot] += now_ - tlast; tlast = now_; } while (0)
#else
#define RB2_T0()
#define RB2_T(slot)
#endif
#pragma unroll 1
    for (;;) {
        // the thread index passes through an opaque move every tile (rblock.hip): everything derived from it is recomputed per tile
        // instead of being hoisted out of the tile loop and spilled
        tid = threadIdx.x % GT;
        asm volatile("" : "+v"(tid));
        lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        wt = wave % WT, wc = wave / WT;
        const int xlane = (RB_GUARD + wt * MT * 32 + (lane & 31)) * PITCH + (lane >> 5) * 16;
        const size_t wlane = (size_t)(wc * NT) * 64 + lane;
        t0 = __builtin_amdgcn_readfirstlane(t0);
        const int base_t = t0 - H;                 // global time of local row 0
        const long long brow = (long long)b * p.T;
        // the workgroup's next PAIR: static (q + G) or, with p.tile_ctr, the next unclaimed pair of the launch.  Lane 0 of the workgroup
        // issues the atomic at the top of group 0's tile; the result is broadcast through LDS (written before the barrier that closes
        // group 0's last contraction, read by group 0 right behind it and by group 1 one barrier later; rewritten 12 barriers later)
        unsigned claim = 0;
        if (p.tile_ctr && threadIdx.x == 0) claim = atomicAdd(p.tile_ctr, 1u);
        int qn = q + G;

        auto load_bias = [&](f32x4 (&bb)[NT][4], const float* bias) {
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) bb[n][qq] = *(const f32x4*)(bias + (wc * NT + n) * 32 + 8 * qq + 4 * (lane >> 5));
        };
        const bool all_inb = base_t >= 0 && base_t + W <= len;   // group-uniform: no row of the tile needs masking
        int n_ovf = 0;
        auto write_act = [&](const f32x16 (&v)[MT][NT]) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int row = (wt * MT + m) * 32 + (lane & 31);
                const int t = base_t + row;
                const bool inb = all_inb || (t >= 0 && t < len);
                const bool counted = inb && row >= H && row < H + TT;   // range guard: the rows this tile outputs (rblock.hip)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        const int co = (wc * NT + n) * 32 + 8 * qq + 4 * (lane >> 5);
                        const f32x4 v4 = {v[m][n][4 * qq], v[m][n][4 * qq + 1], v[m][n][4 * qq + 2], v[m][n][4 * qq + 3]};
                        uint2 pk = act4<EL>(v4, 0.1f);
                        if constexpr (GUARD) n_ovf += counted ? ovf4(v4, 0.1f) : 0;
                        if (!all_inb && !inb) pk = make_uint2(0, 0);
                        *(uint2*)(act + (RB_GUARD + row) * PITCH + co * 2) = pk;
                    }
            }
        };

        // ---- N0 (its first part, the previous tile's epilogue, is at the bottom of the loop)
        uint4 ring[RD][NT];
        f32x4 bb[NT][4];
        RB2_T0();
        rb2_preload<NT, RD>(ring, p.rb[0].w1[0] + wlane, kg_stride);
        load_bias(bb, p.rb[0].b1[0]);
        write_act(xr);
        RB2_T(0);
        __syncthreads();
        RB2_T(1);

        // epilogue addressing (used from the last contraction on)
        bool has_next = false;
        int bn = b, lenn = 0, t0n = 0;
        const auto rs_s = __builtin_amdgcn_make_buffer_rsrc((void*)(p.S + brow * C), 0, len * C * 4, 0x00020000);
        const auto rs_sold = __builtin_amdgcn_make_buffer_rsrc((void*)(p.S + brow * C), 0, p.mode >= 1 ? len * C * 4 : 0, 0x00020000);
        const auto rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Sa ? p.Sa + brow * C : (unsigned short*)(p.S + brow * C)), 0,
                                                            len * C * 2, 0x00020000);
        char* stg = act + (size_t)(RB_GUARD + wt * MT * 32) * PITCH;
        const int er = lane / F4W, ec = lane % F4W;                       // row within an access, 16-byte chunk of the row
        auto eoff = [&](int m, int u) {                                   // byte offset of (slab m, access u) in the utterance, or out of range
            const int row = (wt * MT + m) * 32 + u * RPI + er;
            return (row >= H && row < H + TT) ? ((base_t + row) * C + wc * CW + ec * 4) * 4 : (int)0x80000000;
        };

        f32x16 acc[MT][NT];
#pragma unroll 1
        for (int it = 0; it < 3; ++it) {
            // ---- M: conv1 (the first MFMA of every tile takes the bias pattern as its C operand)
            f32x16 cinit[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq)
#pragma unroll
                    for (int e = 0; e < 4; ++e) cinit[n][4 * qq + e] = bb[n][qq][e];
            load_bias(bb, p.rb[0].b2[it]);
            const int d = p.rb[0].dil[it];
            rb2_contract<EL, MT, NT, NKG, PITCH, RD, true>(acc, ring, act, xlane - ((p.K - 1) / 2) * d * PITCH, p.rb[0].w1[it] + wlane, S, d * PITCH, cinit);
            rb2_preload<NT, RD>(ring, p.rb[0].w2[it] + wlane, kg_stride);
            RB2_T(2);
            __syncthreads();
            RB2_T(3);
            // ---- N: xt (16-bit, activated) overwrites the tile
            write_act(acc);
            RB2_T(4);
            __syncthreads();
            RB2_T(1);
            // ---- M: conv2 accumulates straight into the residual registers: x = x + b2 + W2 * xt
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq)
#pragma unroll
                        for (int e = 0; e < 4; ++e) xr[m][n][4 * qq + e] += bb[n][qq][e];
            if (it < 2) load_bias(bb, p.rb[0].b1[it + 1]);
            rb2_contract<EL, MT, NT, NKG, PITCH, RD, false>(xr, ring, act, xlane - ((p.K - 1) / 2) * PITCH, p.rb[0].w2[it] + wlane, S, PITCH, cinit);
            if (it < 2) rb2_preload<NT, RD>(ring, p.rb[0].w1[it + 1] + wlane, kg_stride);
            if (p.tile_ctr && it == 0 && threadIdx.x == 0) pre[3 * p.B + 1] = G + (int)claim;
            if (it == 2) break;                    // (the barrier that closes the last contraction follows the loop)
            RB2_T(5);
            __syncthreads();
            RB2_T(3);
            if (it == 0) {   // the next pair is known: place this group's next tile
                if (p.tile_ctr) qn = __builtin_amdgcn_readfirstlane(pre[3 * p.B + 1]);
                has_next = qn < npairs;            // workgroup-uniform
                if (has_next) place(2 * qn + grp, bn, lenn, t0n);
            }
            // ---- N
            write_act(xr);
            RB2_T(6);
            __syncthreads();
            RB2_T(1);
        }
        // the stage sum of this tile's rows (read-modify-written by the epilogue) starts its trip now: it is younger than every weight
        // fragment, so no contraction step waited for it, and it lands while this group waits for the partner's phase
        u32x4 sold[MT][NRD];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int u = 0; u < NRD; ++u) {
                sold[m][u] = __builtin_amdgcn_raw_buffer_load_b128(rs_sold, eoff(m, u), 0, VP_LD_AUX);   // (mode 0: an empty resource, zeros)
            }
        // ... and so does the NEXT tile's residual stream, into the accumulator registers conv1 no longer needs (in-order VMEM returns: issued
        // any earlier, these HBM round trips would sit in front of the contractions' weight fragments)
#pragma unroll
        for (int m = 0; m < MT; ++m) load_x(acc[m], m, bn, t0n - H, lenn);   // (no next tile: zero length, zeros without traffic)
        RB2_T(5);
        __syncthreads();
        RB2_T(3);
        if constexpr (GUARD) {
            if (n_ovf) atomicAdd(p.ovf, (unsigned long long)n_ovf);
        }

        // ---- N0 of the next round, first part: this tile's epilogue, wave-private (its stage-sum loads are already in flight).  Rows [H, H + TT) leave as whole rows of the wave's
        // CW channels through a 32-row fp32 staging slab in the wave's OWN rows of the (now dead) activation tile: no other wave reads or
        // writes those rows outside a contraction, so no barrier is needed.
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = xr[m][n][4 * qq + e];
                    *(f32x4*)(stg + (lane & 31) * EPW + (n * 32 + 8 * qq + 4 * (lane >> 5)) * 4) = v;
                }
            // (the LDS serves one wave's accesses in order: the reads below see the writes above without a barrier)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int u = 0; u < NRD; ++u) {
                const int off = eoff(m, u);
                f32x4 o = *(const f32x4*)(stg + (u * RPI + er) * EPW + ec * 16);
                o += __builtin_bit_cast(f32x4, sold[m][u]);                // xs += resblock(x)  (hifigan.py:133-135); zeros in mode 0
                if (p.mode == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = o[e] / p.div;
                }
                if (!(p.mode == 2 && p.Sa && p.drop_S))
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs_s, off, 0, VP_ST_AUX);
                if (p.mode == 2 && p.Sa) {
                    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
                    const u32x2 pk = {pack2bf(lrelu(o[0], p.slope), lrelu(o[1], p.slope)), pack2bf(lrelu(o[2], p.slope), lrelu(o[3], p.slope))};
                    __builtin_amdgcn_raw_buffer_store_b64(pk, rs_a, off == (int)0x80000000 ? off : off >> 1, 0, VP_ST_AUX);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();                              // slab m + 1 reuses the staging rows
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
#ifdef DTTS_ABLATE
        asm volatile("s_waitcnt vmcnt(0)");   // (the stores' acknowledgements and the next x count as epilogue time here)
        RB2_T(7);
        tsum[8] += 1;
        if (!has_next && p.stats && tid == 0)
            for (int i = 0; i < 9; ++i) atomicAdd(p.stats + grp * 9 + i, tsum[i]);
#endif
        __syncthreads();                           // closes the epilogue slot
        // ---- slot 13: the next tile's residual stream has had the partner's M - N - M to arrive: into the residual registers
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) xr[m][n] = acc[m][n];
        asm volatile("" ::: "memory");
        __syncthreads();
        if (!has_next) break;
        q = qn;
        b = bn;
        len = lenn;
        t0 = t0n;
    }
    if (grp == 0) {                                // matches group 1's leading barriers
#pragma unroll 1
        for (int i = 0; i < 7; ++i) __syncthreads();
    }
}

template <int C, int MT, int NT, int WT, int WC, int EL, bool GUARD = false>
static hipError_t rb2_launch_cfg(const RBlockParams& p, hipStream_t stream) {
    constexpr int W = 32 * MT * WT, PITCH = C * 2 + 16;
    const int H = 6 * (p.K - 1), TT = W - 2 * H;
    if (TT < 32 || p.wav) return hipErrorInvalidValue;
    size_t lds = 2 * (size_t)(W + 2 * RB_GUARD) * PITCH;
    RBlockParams q = p;
    q.pre_off = (int)lds;                          // tile table: prefix sums [B + 1], counts [B], lengths [B], the claimed pair
    lds += (size_t)(3 * p.B + 2) * sizeof(int);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if constexpr (EL == EL_F16 && !GUARD) {
        if (p.ovf) return rb2_launch_cfg<C, MT, NT, WT, WC, EL, true>(p, stream);
    }
    auto kern = rblock2_kernel<C, MT, NT, WT, WC, EL, GUARD>;
    static bool configured_dev[64] = {};
    static int cus_dev[64] = {};
    int cur_dev = 0;
    (void)hipGetDevice(&cur_dev);
    if (!configured_dev[cur_dev & 63]) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        configured_dev[cur_dev & 63] = true;
    }
    int& cus = cus_dev[cur_dev & 63];
    if (!cus) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, cur_dev) != hipSuccess) return hipErrorInvalidDevice;
        cus = prop.multiProcessorCount;
    }
    const long long max_pairs = ((long long)p.B * ((p.T + TT - 1) / TT) + 1) / 2;
    const int grid = (int)std::min<long long>((long long)cus, max_pairs);   // one 8-wave workgroup per CU
    if (grid <= 0) return hipSuccess;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(128 * WT * WC), lds, stream, q);
    return hipGetLastError();
}

// the configurations the two-group form covers (rblock.hip's launcher asks first): C = 32, every k, without the fused conv_post
bool rblock2_supported(int C, int K, bool wav) { return C == 32 && !wav && (K & 1) && K >= 3 && K <= 11; }

hipError_t rblock2_launch(const RBlockParams& p, int C, hipStream_t stream) {
    if (!rblock2_supported(C, p.K, p.wav != nullptr)) return hipErrorInvalidValue;
    // two groups of 4 waves, a 512-row tile each (2 x 47 KB of LDS)
    return p.el == EL_F16 ? rb2_launch_cfg<32, 4, 1, 4, 1, EL_F16>(p, stream) : rb2_launch_cfg<32, 4, 1, 4, 1, EL_BF16>(p, stream);
}

} // namespace dtts
