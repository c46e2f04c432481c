// The FVAE prior flow in reverse (z_p -> z_q; modules/dict_tts/fvae_semantics.py:112-113, modules/commons/glow_modules /
// ResidualCouplingBlock.forward(reverse=True): for each block, last to first: Flip, then the mean-only coupling layer
//     h = pre(x0) ; out = WN(h, g) ; m = post(out) ; x1 = x1 - m
// with WN (modules/commons/wavenet.py:54-78) = n_layers x [ in_layer (k = 3) + cond -> tanh * sigmoid -> res_skip (1x1) ]) as ONE
// kernel in exact fp32.  Launch by launch that is 11 kernels per block — pre, cond, 4 x (in, res_skip), post — each a few GFLOP on
// T_mel/4 rows x 64 channels, i.e. 44 latency-bound launches per batch; here a workgroup takes a chunk of 128 rows (of which
// 128 - 2 * n_blocks * n_layers are valid: every k = 3 convolution eats one halo row per side) through ALL blocks:
//   * z (16 channels), the WaveNet state h (64) and the gated activations (64) live in LDS as fp32 rows; wave w owns rows
//     and each pair of waves owns a row tile of 32 rows (one wave per channel half: a tanh co-tile, its sigmoid partner, the res and
//     skip co-tiles of the same 32 channels), so the gate and the skip accumulator are wave-local; two barriers per layer;
//   * contractions on v_mfma_f32_32x32x2_f32 (D[co][t]: A = weights, B = LDS rows); the K order is chosen so that a lane's four
//     consecutive steps read four consecutive channels: one ds_read_b128 of the state and one 16 B weight fragment per co-tile
//     feed 16 MFMAs.  Weights stream from L2 in fragment order (context.hip: flowstack_pack);
//   * the conditioning (cond_layer(g) of all blocks: one [B*T4, hidden] x [hidden, 2048] convolution, launched once before) enters as
//     the accumulators' initial value together with the bias;
//   * pre (8 -> 64) and post (64 -> 8) on the VALU; rows outside [0, T4) are forced to zero after every update = the
//     convolutions' zero padding (the reference runs the flow with x_mask = 1 over the padded batch).
#include "flowstack.h"
#include "rb_common.h"

#include <algorithm>
#include <cstring>

namespace dtts {

typedef __attribute__((ext_vector_type(16))) float fs16;
typedef __attribute__((ext_vector_type(4))) float fs4;

namespace {
constexpr int W = 128;                 // rows per chunk (4 waves x 32)
constexpr int HP = FS_H + 4;           // LDS row pitch of h / acts (floats): 272 B rows, conflict-free ds_read_b128 over 16 rows
constexpr int ZP = 16;                 // z row (floats)
}

// X3: the two convolutions of every layer on v_mfma_f32_32x32x16_bf16 with bf16 hi / lo split operands (Wlo*Xhi + Whi*Xlo + Whi*Xhi: 16-bit
// significand products, fp32 accumulation — the decoder WaveNet's arithmetic, 5.3x the fp32-MFMA rate); the fp32 rows in LDS are split
// on the fly (8 values per lane and fragment).  X3 = false: exact fp32 MFMA (dtts_config.decoder_fp32).
// 8 waves: wave (rt, ch) owns row tile rt = rows 32 rt .. 32 rt + 31 and the channel half ch: in_layer co-tiles {ch (tanh), ch + 2 (its
// sigmoid partner)}, res_skip co-tiles {ch (res), ch + 2 (skip)}, i.e. channels 32 ch .. 32 ch + 31 of h, of the activations and of the
// skip sum.  Two waves per SIMD: one wave's LDS / L2 / transcendental latencies run under the other's MFMAs.
template <bool X3>
__global__ __launch_bounds__(512, 1) void flowstack_kernel(const FlowStackParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* zt = (float*)smem_raw;                       // [W][ZP]
    float* hb = zt + W * ZP;                            // [W + 2][HP], row 0 and W + 1 are zero guards
    float* ab = hb + (W + 2) * HP;                      // [W][HP]
    float* wp = ab + W * HP;                            // pre / post weights of the current block: FS_PRE + FS_POST floats
    float* pm = wp + FS_PRE + FS_POST;                  // [W][FS_HALF]: channel half 1's partial sums of post()
    float* rb = pm + W * FS_HALF;                       // [layers][2 * FS_H]: the current block's res_skip biases (an L2 round trip per layer otherwise)
    float* ib = rb + p.layers * 2 * FS_H;               // [layers][2 * FS_H]: its in_layer biases

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const int rt = wave & 3, ch = wave >> 2;
    const int halo = p.n_flows * p.layers;
    const int RC = W - 2 * halo;
    const int b = blockIdx.y;
    const int t_base = blockIdx.x * RC - halo;
    const int row_l = 32 * rt + (lane & 31);
    const int t = t_base + row_l;
    const bool inb = t >= 0 && t < p.T4;
    const long long grow = (long long)b * p.T4 + t;    // global row (valid when inb)

    // ---- z tile (zeros outside the sequence), guard rows
    for (int i = tid; i < W * ZP / 4; i += 512) {
        const int r = i / (ZP / 4), c = i % (ZP / 4);
        const int tt = t_base + r;
        fs4 v = {0.f, 0.f, 0.f, 0.f};
        if (tt >= 0 && tt < p.T4 && c * 4 < p.Z) v = *(const fs4*)(p.z_in + ((long long)b * p.T4 + tt) * p.Z + c * 4);
        *(fs4*)(zt + r * ZP + c * 4) = v;
    }
    for (int i = tid; i < 2 * HP; i += 512) hb[(i < HP ? 0 : (W + 1) * HP - HP) + i] = 0.f;
    float* hrow = hb + (1 + row_l) * HP;               // this lane's row of the state
    float* arow = ab + row_l * HP;
    const int c0 = 32 * ch + 4 * half;                 // this lane's first channel of quad q = 0 (quad q: c0 + 8 q .. + 3)
    // 8 consecutive fp32 values of an LDS row -> bf16 hi / lo fragments (round-to-nearest-even both)
    auto split8 = [&](const float* src, uint4& hi, uint4& lo) {
        const fs4 v0 = *(const fs4*)src, v1 = *(const fs4*)(src + 4);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        unsigned hw[8];
        float r[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            hw[i] = rf2bf(v[i]);
            r[i] = v[i] - __builtin_bit_cast(float, hw[i] << 16);
        }
        hi = make_uint4(hw[0] | (hw[1] << 16), hw[2] | (hw[3] << 16), hw[4] | (hw[5] << 16), hw[6] | (hw[7] << 16));
        lo = make_uint4(pack2bf(r[0], r[1]), pack2bf(r[2], r[3]), pack2bf(r[4], r[5]), pack2bf(r[6], r[7]));
    };

    // this lane's conditioning of layer g = block * layers + layer (its row; tanh quad q of tile ch in [0][q], its sigmoid partner in
    // [1][q]), fetched one layer ahead as PLAIN loads — nothing consumes them before the next layer's accumulators are initialised
    // (cond + in_layer bias from LDS), so the HBM round trip runs under the gate and the 1x1 convolution.  (Adding the bias here, as
    // rounds 1-2 did, made every group of loads wait for its data on the spot: four serial round trips per layer.)
    fs4 cnd[2][4];
    auto load_cond = [&](int g) {
        if (g >= p.n_flows * p.layers) return;
        const float* cr = p.cond + grow * p.ld_cond + (size_t)g * 2 * FS_H + c0;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                cnd[m][q] = fs4{0.f, 0.f, 0.f, 0.f};
                if (inb) cnd[m][q] = *(const fs4*)(cr + 64 * m + 8 * q);
            }
    };
    load_cond(0);
    const size_t flow_floats = fs_flow_floats(p.layers);
#pragma unroll 1
    for (int f = 0; f < p.n_flows; ++f) {
        const float* wf = p.w + (size_t)f * flow_floats;
        __syncthreads();   // the previous block's readers of wp / hb / pm are done (and the z tile / guards are in place)
        for (int i = tid; i < (int)FS_PRE; i += 512) wp[i] = wf[i];
        for (int i = tid; i < (int)FS_POST; i += 512) wp[FS_PRE + i] = wf[FS_PRE + (size_t)p.layers * FS_LAYER + i];
        for (int i = tid; i < p.layers * 2 * FS_H; i += 512)
            rb[i] = wf[FS_PRE + (size_t)(i / (2 * FS_H)) * FS_LAYER + FS_IN_FRAGS + 2 * FS_H + FS_RS_FRAGS + i % (2 * FS_H)];
        for (int i = tid; i < p.layers * 2 * FS_H; i += 512) ib[i] = wf[FS_PRE + (size_t)(i / (2 * FS_H)) * FS_LAYER + FS_IN_FRAGS + i % (2 * FS_H)];
        __syncthreads();
        // ---- h = pre(x0): this lane's 16 channels of its row
        {
            const fs4 xa = *(const fs4*)(zt + row_l * ZP + p.in_coff[f]), xb = *(const fs4*)(zt + row_l * ZP + p.in_coff[f] + 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                fs4 hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int c = c0 + 8 * q + e;
                    const fs4 wa = *(const fs4*)(wp + c * FS_HALF), wb = *(const fs4*)(wp + c * FS_HALF + 4);
                    float a = wp[FS_H * FS_HALF + c];
                    a += wa[0] * xa[0]; a += wa[1] * xa[1]; a += wa[2] * xa[2]; a += wa[3] * xa[3];
                    a += wb[0] * xb[0]; a += wb[1] * xb[1]; a += wb[2] * xb[2]; a += wb[3] * xb[3];
                    hv[e] = inb ? a : 0.f;
                }
                *(fs4*)(hrow + c0 + 8 * q) = hv;
            }
        }
        fs16 skip;
#pragma unroll
        for (int i = 0; i < 16; ++i) skip[i] = 0.f;
        __syncthreads();
#pragma unroll 1
        for (int l = 0; l < p.layers; ++l) {
            const float* wl = wf + FS_PRE + (size_t)l * FS_LAYER;
            const bool last = l == p.layers - 1;
            // ---- x_in = in_layer(h) + bias + cond (tiles ch and ch + 2): the accumulators start at bias + cond
            fs16 acc[2];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const fs4 bi = *(const fs4*)(ib + l * 2 * FS_H + 64 * m + c0 + 8 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[m][4 * q + e] = bi[e] + cnd[m][q][e];
                }
            if constexpr (X3) {
                // fragment f = (tap * 4 + kb) * 4 + n: hi at uint4 index (2f) * 64 + lane, lo at (2f + 1) * 64 + lane; PFD steps ahead
                constexpr int NIT = FS_K * (FS_H / 16), PFD = 3;
                const uint4* wq = (const uint4*)wl + lane;
                uint4 ah[PFD + 1][2], al[PFD + 1][2];
#pragma unroll
                for (int s0 = 0; s0 < PFD; ++s0)
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        ah[s0][m] = wq[(2 * (s0 * 4 + ch + 2 * m)) * 64];
                        al[s0][m] = wq[(2 * (s0 * 4 + ch + 2 * m) + 1) * 64];
                    }
                uint4 xh, xl;
                split8(hrow - HP + 8 * half, xh, xl);
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    if (it + PFD < NIT) {
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            ah[(it + PFD) % (PFD + 1)][m] = wq[(2 * ((it + PFD) * 4 + ch + 2 * m)) * 64];
                            al[(it + PFD) % (PFD + 1)][m] = wq[(2 * ((it + PFD) * 4 + ch + 2 * m) + 1) * 64];
                        }
                    }
                    uint4 nh = xh, nl = xl;
                    if (it + 1 < NIT) {
                        const int tap = (it + 1) / (FS_H / 16), kb = (it + 1) % (FS_H / 16);
                        split8(hrow + (tap - 1) * HP + 16 * kb + 8 * half, nh, nl);
                    }
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        acc[m] = mfma16<EL_BF16>(al[it % (PFD + 1)][m], xh, acc[m]);
                        acc[m] = mfma16<EL_BF16>(ah[it % (PFD + 1)][m], xl, acc[m]);
                        acc[m] = mfma16<EL_BF16>(ah[it % (PFD + 1)][m], xh, acc[m]);
                    }
                    xh = nh;
                    xl = nl;
                }
            } else {
                // exact fp32: fragment (it = tap * 8 + j, n)[lane] = 4 steps; weight fragments PFD steps ahead, the state row one step ahead
                constexpr int NIT = FS_K * (FS_H / 8), PFD = 3;
                const fs4* in_frag = (const fs4*)wl + lane;
                fs4 a[PFD + 1][2];
#pragma unroll
                for (int s0 = 0; s0 < PFD; ++s0)
#pragma unroll
                    for (int m = 0; m < 2; ++m) a[s0][m] = in_frag[(s0 * 4 + ch + 2 * m) * 64];
                fs4 bv = *(const fs4*)(hrow - HP + 4 * half);
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    if (it + PFD < NIT) {
#pragma unroll
                        for (int m = 0; m < 2; ++m) a[(it + PFD) % (PFD + 1)][m] = in_frag[((it + PFD) * 4 + ch + 2 * m) * 64];
                    }
                    fs4 bn = bv;
                    if (it + 1 < NIT) {
                        const int tap = (it + 1) / (FS_H / 8), j = (it + 1) % (FS_H / 8);
                        bn = *(const fs4*)(hrow + (tap - 1) * HP + 4 * half + 8 * j);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[it % (PFD + 1)][m][e], bv[e], acc[m], 0, 0, 0);
                    bv = bn;
                }
            }
            load_cond(f * p.layers + l + 1);   // the next layer's (or block's) conditioning travels during the gate and the 1x1 convolution
            // ---- acts = tanh(x_in[:H]) * sigmoid(x_in[H:]) -> this lane's 16 channels of its row of the activation tile
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                fs4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // tanh(a) * sigmoid(g) = (1 - 2 / (e^{2a} + 1)) / (1 + e^{-g}), hardware exp2 / rcp (a few ulp; the launch-by-launch
                    // path calls tanhf / expf): both forms saturate cleanly (e^{2a} = inf -> 1, 0 -> -1)
                    const float ea = __expf(2.f * acc[0][4 * q + e]), eg = __expf(-acc[1][4 * q + e]);
                    v[e] = (1.f - 2.f * __frcp_rn(ea + 1.f)) * __frcp_rn(1.f + eg);
                }
                *(fs4*)(arow + c0 + 8 * q) = v;
            }
            // ---- res_skip(acts) (1x1): tiles ch (res; the last layer: skip) and ch + 2 (skip), bias as the initial accumulators
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    fs4 v = {0.f, 0.f, 0.f, 0.f};
                    if (m == 0 || !last) v = *(const fs4*)(rb + l * 2 * FS_H + 64 * m + c0 + 8 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[m][4 * q + e] = v[e];
                }
            __syncthreads();   // the row tile's activations are complete (both channel halves); every wave is done reading h for this layer
            if constexpr (X3) {
                constexpr int NIT = FS_H / 16;
                const uint4* wq = (const uint4*)(wl + FS_IN_FRAGS + 2 * FS_H) + lane;
                uint4 ah[NIT][2], al[NIT][2];
#pragma unroll
                for (int it = 0; it < NIT; ++it)
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const bool on = m == 0 || !last;
                        ah[it][m] = on ? wq[(2 * (it * 4 + ch + 2 * m)) * 64] : make_uint4(0, 0, 0, 0);
                        al[it][m] = on ? wq[(2 * (it * 4 + ch + 2 * m) + 1) * 64] : make_uint4(0, 0, 0, 0);
                    }
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    uint4 xh, xl;
                    split8(arow + 16 * it + 8 * half, xh, xl);
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        if (m == 1 && last) continue;
                        acc[m] = mfma16<EL_BF16>(al[it][m], xh, acc[m]);
                        acc[m] = mfma16<EL_BF16>(ah[it][m], xl, acc[m]);
                        acc[m] = mfma16<EL_BF16>(ah[it][m], xh, acc[m]);
                    }
                }
            } else {
                constexpr int NIT = FS_H / 8, PFD = 3;
                const fs4* rs_frag = (const fs4*)(wl + FS_IN_FRAGS + 2 * FS_H) + lane;
                fs4 a[PFD + 1][2];
#pragma unroll
                for (int s0 = 0; s0 < PFD; ++s0)
#pragma unroll
                    for (int m = 0; m < 2; ++m) a[s0][m] = (m == 0 || !last) ? rs_frag[(s0 * 4 + ch + 2 * m) * 64] : fs4{0.f, 0.f, 0.f, 0.f};
                fs4 bv = *(const fs4*)(arow + 4 * half);
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    if (it + PFD < NIT) {
#pragma unroll
                        for (int m = 0; m < 2; ++m)
                            a[(it + PFD) % (PFD + 1)][m] = (m == 0 || !last) ? rs_frag[((it + PFD) * 4 + ch + 2 * m) * 64] : fs4{0.f, 0.f, 0.f, 0.f};
                    }
                    fs4 bn = bv;
                    if (it + 1 < NIT) bn = *(const fs4*)(arow + 8 * (it + 1) + 4 * half);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[it % (PFD + 1)][0][e], bv[e], acc[0], 0, 0, 0);
                        if (!last) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[it % (PFD + 1)][1][e], bv[e], acc[1], 0, 0, 0);
                    }
                    bv = bn;
                }
            }
            if (!last) {
                // h = h + res ; skip += skip part   (wavenet.py:71-75)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float* hp = hrow + c0 + 8 * q;
                    fs4 v = *(const fs4*)hp;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = inb ? v[e] + acc[0][4 * q + e] : 0.f;
                    *(fs4*)hp = v;
                }
                skip += acc[1];
            } else {
                skip += acc[0];
            }
            __syncthreads();   // h is updated; every wave is done reading the activations
        }
        // ---- x1 = x1 - post(out): 8 outputs per row; this lane holds 16 of the 64 channels, lane ^ 32 another 16, the other channel
        // half's wave the remaining 32 (through LDS)
        {
            const float* wq = wp + FS_PRE;             // Wpost'[8][64], bpost'[8]
            float m[FS_HALF];
#pragma unroll
            for (int o = 0; o < FS_HALF; ++o) {
                float a = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const fs4 wv = *(const fs4*)(wq + o * FS_H + c0 + 8 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) a += wv[e] * skip[4 * q + e];
                }
                m[o] = a + __shfl_xor(a, 32, 64);
            }
            if (ch == 1 && half == 0) {
#pragma unroll
                for (int o = 0; o < FS_HALF; ++o) pm[row_l * FS_HALF + o] = m[o];
            }
            __syncthreads();
            if (ch == 0 && half == 0) {
                float* zp = zt + row_l * ZP + p.out_coff[f];
#pragma unroll
                for (int o = 0; o < FS_HALF; ++o) zp[o] += (m[o] + pm[row_l * FS_HALF + o]) + wq[FS_HALF * FS_H + o];
            }
        }
    }
    __syncthreads();
    // ---- the valid rows of the chunk leave
    for (int i = tid; i < W * ZP / 4; i += 512) {
        const int r = i / (ZP / 4), c = i % (ZP / 4);
        const int tt = t_base + r;
        if (r >= halo && r < W - halo && tt >= 0 && tt < p.T4 && c * 4 < p.Z)
            *(fs4*)(p.z_out + ((long long)b * p.T4 + tt) * p.Z + c * 4) = *(const fs4*)(zt + r * ZP + c * 4);
    }
}

bool flowstack_supported(int hidden, int kernel, int layers, int blocks, int latent) {
    return hidden == FS_H && kernel == FS_K && latent == 2 * FS_HALF && layers >= 1 && layers <= FS_MAX_LAYERS && blocks >= 1 &&
           blocks <= FS_MAX_FLOWS && 2 * blocks * layers <= W - 32;
}

static unsigned short fs_bf16(float f) {   // round-to-nearest-even
    unsigned u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static float fs_bf16_f(unsigned short h) {
    const unsigned u = (unsigned)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

void flowstack_pack(const FlowStackHostWeights& w, int layers, bool x3, std::vector<float>& out) {
    const size_t base = out.size();
    out.resize(base + fs_flow_floats(layers), 0.f);
    float* o = out.data() + base;
    for (int c = 0; c < FS_H; ++c)
        for (int i = 0; i < FS_HALF; ++i) o[c * FS_HALF + i] = w.pre[(size_t)c * FS_HALF + i];
    for (int c = 0; c < FS_H; ++c) o[FS_H * FS_HALF + c] = w.bpre[c];
    for (int l = 0; l < layers; ++l) {
        float* wl = o + FS_PRE + (size_t)l * FS_LAYER;
        const bool last = l == layers - 1;
        if (x3) {
            // split-operand form: fragment f = (tap * 4 + kb) * 4 + n (in_layer) / kb * 4 + n (res_skip): 8 bf16 hi at uint4 (2f) * 64 + lane,
            // 8 bf16 lo at (2f + 1) * 64 + lane; element i = W[co = 32 n + (lane & 31)][ci = 16 kb + 8 (lane >> 5) + i][tap]
            auto put = [&](float* dst, size_t f, int lane, int i, float v) {
                unsigned short* hi = (unsigned short*)(dst + ((2 * f) * 64 + lane) * 4);
                unsigned short* lo = (unsigned short*)(dst + ((2 * f + 1) * 64 + lane) * 4);
                hi[i] = fs_bf16(v);
                lo[i] = fs_bf16(v - fs_bf16_f(hi[i]));
            };
            const int n_out = last ? FS_H : 2 * FS_H;
            for (int tap = 0; tap < FS_K; ++tap)
                for (int kb = 0; kb < FS_H / 16; ++kb)
                    for (int n = 0; n < 4; ++n)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int i = 0; i < 8; ++i) {
                                const int co = 32 * n + (lane & 31), ci = 16 * kb + 8 * (lane >> 5) + i;
                                put(wl, ((size_t)tap * (FS_H / 16) + kb) * 4 + n, lane, i, w.in[l][((size_t)co * FS_H + ci) * FS_K + tap]);
                            }
            for (int c = 0; c < 2 * FS_H; ++c) wl[FS_IN_FRAGS + c] = w.bin[l][c];
            float* wr = wl + FS_IN_FRAGS + 2 * FS_H;
            for (int kb = 0; kb < FS_H / 16; ++kb)
                for (int n = 0; n < 4; ++n)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int i = 0; i < 8; ++i) {
                            const int co = 32 * n + (lane & 31), ci = 16 * kb + 8 * (lane >> 5) + i;
                            put(wr, (size_t)kb * 4 + n, lane, i, co < n_out ? w.rs[l][(size_t)co * FS_H + ci] : 0.f);
                        }
            for (int c = 0; c < 2 * FS_H; ++c) wr[FS_RS_FRAGS + c] = c < n_out ? w.brs[l][c] : 0.f;
            continue;
        }
        // fragment(tap, j, n, lane)[e] = W[co = 32 n + (lane & 31)][ci = 8 j + 4 (lane >> 5) + e][tap]
        for (int tap = 0; tap < FS_K; ++tap)
            for (int j = 0; j < FS_H / 8; ++j)
                for (int n = 0; n < 4; ++n)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 4; ++e) {
                            const int co = 32 * n + (lane & 31), ci = 8 * j + 4 * (lane >> 5) + e;
                            wl[((((size_t)tap * (FS_H / 8) + j) * 4 + n) * 64 + lane) * 4 + e] = w.in[l][((size_t)co * FS_H + ci) * FS_K + tap];
                        }
        for (int c = 0; c < 2 * FS_H; ++c) wl[FS_IN_FRAGS + c] = w.bin[l][c];
        float* wr = wl + FS_IN_FRAGS + 2 * FS_H;
        const int n_out = last ? FS_H : 2 * FS_H;
        for (int j = 0; j < FS_H / 8; ++j)
            for (int n = 0; n < 4; ++n)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 4; ++e) {
                        const int co = 32 * n + (lane & 31), ci = 8 * j + 4 * (lane >> 5) + e;
                        wr[(((size_t)j * 4 + n) * 64 + lane) * 4 + e] = co < n_out ? w.rs[l][(size_t)co * FS_H + ci] : 0.f;
                    }
        for (int c = 0; c < 2 * FS_H; ++c) wr[FS_RS_FRAGS + c] = c < n_out ? w.brs[l][c] : 0.f;
    }
    float* wq = o + FS_PRE + (size_t)layers * FS_LAYER;
    for (int q = 0; q < FS_HALF; ++q)
        for (int c = 0; c < FS_H; ++c) wq[q * FS_H + c] = w.post[(size_t)q * FS_H + c];
    for (int q = 0; q < FS_HALF; ++q) wq[FS_HALF * FS_H + q] = w.bpost[q];
}

hipError_t flowstack_launch(const FlowStackParams& p, hipStream_t stream) {
    const int halo = p.n_flows * p.layers, RC = W - 2 * halo;
    if (RC < 32 || p.Z != 2 * FS_HALF || p.z_in == p.z_out) return hipErrorInvalidValue;
    if (p.T4 <= 0 || p.B <= 0) return hipSuccess;
    const size_t lds = ((size_t)W * ZP + (size_t)(W + 2) * HP + (size_t)W * HP + FS_PRE + FS_POST + (size_t)W * FS_HALF + (size_t)p.layers * 4 * FS_H) * sizeof(float);
    // per device (hipFuncSetAttribute is per device; a process may hold contexts on several GPUs)
    static bool configured_dev[64] = {};
    int cur_dev = 0;
    (void)hipGetDevice(&cur_dev);
    bool& configured = configured_dev[cur_dev & 63];
    if (!configured) {
        hipError_t e = hipFuncSetAttribute((const void*)flowstack_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)flowstack_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        configured = true;
    }
    dim3 grid((p.T4 + RC - 1) / RC, p.B);
    if (p.x3) hipLaunchKernelGGL(flowstack_kernel<true>, grid, dim3(512), lds, stream, p);
    else hipLaunchKernelGGL(flowstack_kernel<false>, grid, dim3(512), lds, stream, p);
    return hipGetLastError();
}

} // namespace dtts
