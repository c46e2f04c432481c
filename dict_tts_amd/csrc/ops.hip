// Non-GEMM kernels of the Dict-TTS path.  See ops.h for the contracts and the reference lines they follow.
#include "ops.h"
#include "tune_env.h"

#include <algorithm>
#include <cstdlib>

namespace dtts {

typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---------------------------------------------------------------------------------------------------------
__global__ void embed_kernel(const int64_t* tok, const float* table, float scale, float* x, int rows, int C, int n_rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    long long id = tok[row];
    if (id < 0 || id >= n_rows) id = 0;
    const f32x4* src = (const f32x4*)(table + id * C);
    f32x4* dst = (f32x4*)(x + (long long)row * C);
    for (int c = lane; c < C / 4; c += 64) {
        f32x4 v = src[c];
        dst[c] = v * scale;
    }
}
__global__ void count_pos_kernel(const int64_t* tok, int* lens, int T) {
    const int b = blockIdx.x;
    int n = 0;
    for (int t = threadIdx.x; t < T; t += 64) n += tok[(long long)b * T + t] > 0 ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
    if (threadIdx.x == 0) lens[b] = n;
}
hipError_t embed_launch(const int64_t* tok, const float* table, float scale, float* x, int* lens, int B, int T, int C,
                        int n_rows, hipStream_t s) {
    hipLaunchKernelGGL(embed_kernel, dim3((B * T + 3) / 4), dim3(256), 0, s, tok, table, scale, x, B * T, C, n_rows);
    hipLaunchKernelGGL(count_pos_kernel, dim3(B), dim3(64), 0, s, tok, lens, T);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
__global__ void layernorm_kernel(float* x, float* y, const float* gamma, const float* beta, float eps, const int* lens,
                                 int mask_in, int mask_out, int T, int C, int rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int b = row / T, t = row % T;
    const bool pad = lens && t >= lens[b];
    float* xr = x + (long long)row * C;
    float* yr = y + (long long)row * C;
    if (pad && mask_out) {
        for (int c = lane; c < C; c += 64) yr[c] = 0.f;
        if (mask_in) for (int c = lane; c < C; c += 64) xr[c] = 0.f;
        return;
    }
    if (pad && mask_in) {  // LayerNorm of an all-zero row: (0 - 0) * rsqrt(0 + eps) * gamma + beta = beta
        for (int c = lane; c < C; c += 64) {
            xr[c] = 0.f;
            yr[c] = beta[c];
        }
        return;
    }
    float sum = 0.f;
    for (int c = lane; c < C; c += 64) sum += xr[c];
    const float mean = wave_sum(sum) / (float)C;
    float sq = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float d = xr[c] - mean;
        sq += d * d;
    }
    const float var = wave_sum(sq) / (float)C;
    const float inv = 1.0f / sqrtf(var + eps);
    for (int c = lane; c < C; c += 64) yr[c] = (xr[c] - mean) * inv * gamma[c] + beta[c];
}
hipError_t layernorm_launch(float* x, float* y, const float* gamma, const float* beta, float eps, const int* lens,
                            int mask_in, int mask_out, int B, int T, int C, hipStream_t s) {
    hipLaunchKernelGGL(layernorm_kernel, dim3((B * T + 3) / 4), dim3(256), 0, s, x, y, gamma, beta, eps, lens, mask_in,
                       mask_out, T, C, B * T);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Multi-head attention, online softmax.  Block = 4 waves x 4 queries; K/V tiles of 64 keys staged in LDS.
constexpr int MHA_QT = 16, MHA_KT = 64, MHA_DK_MAX = 96;
__global__ __launch_bounds__(256) void mha_kernel(const float* qkv, float* out, const int* lens, int T, int C, int dk) {
    __shared__ float qs[MHA_QT][MHA_DK_MAX];
    __shared__ float ks[MHA_KT][MHA_DK_MAX + 1];
    __shared__ float vs[MHA_KT][MHA_DK_MAX];
    __shared__ float ps[MHA_QT][MHA_KT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * MHA_QT;
    const int len = lens ? min(max(lens[b], 0), T) : T;
    const float* base = qkv + (long long)b * T * 3 * C;
    const float inv_sqrt = 1.0f / sqrtf((float)dk);
    if (q0 >= len) {   // a tile of padded queries: every caller masks these rows; zeros instead of attention over nothing
        for (int idx = tid; idx < MHA_QT * dk; idx += 256) {
            const int i = idx / dk, d = idx % dk;
            if (q0 + i < T) out[((long long)b * T + q0 + i) * C + h * dk + d] = 0.f;
        }
        return;
    }
    // keys past the utterance's end carry the -1e4 fill: their softmax weight exp(-1e4 - max) is exactly 0 in fp32 for
    // every valid query, so whole key tiles beyond `len` are skipped and the rows beyond it inside the last tile are
    // loaded as zeros (they may be uninitialised: their producers skip dead tiles too).  Bit-identical for valid rows.
    const int kend = len;
    for (int idx = tid; idx < MHA_QT * dk; idx += 256) {
        const int i = idx / dk, d = idx % dk;
        qs[i][d] = (q0 + i < len) ? base[(long long)(q0 + i) * 3 * C + h * dk + d] : 0.f;
    }
    float m[4], l[4], o0[4], o1[4];
#pragma unroll
    for (int qi = 0; qi < 4; ++qi) { m[qi] = -1e30f; l[qi] = 0.f; o0[qi] = 0.f; o1[qi] = 0.f; }
    for (int j0 = 0; j0 < kend; j0 += MHA_KT) {
        __syncthreads();
        for (int idx = tid; idx < MHA_KT * dk; idx += 256) {
            const int j = idx / dk, d = idx % dk;
            const bool ok = j0 + j < kend;
            const float* r = base + (long long)(j0 + j) * 3 * C + h * dk + d;
            ks[j][d] = ok ? r[C] : 0.f;
            vs[j][d] = ok ? r[2 * C] : 0.f;
        }
        __syncthreads();
        const int j = j0 + lane;
        float sc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int d = 0; d < dk; ++d) {
            const float kv = ks[lane][d];
#pragma unroll
            for (int qi = 0; qi < 4; ++qi) sc[qi] += qs[wave * 4 + qi][d] * kv;
        }
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) {
            const int i = q0 + wave * 4 + qi;
            float s = sc[qi] * inv_sqrt;
            if (!(i < len && j < len)) s = -1e4f;  // masked_fill(mask == 0, -1e4)
            const bool exists = j < T;
            const float tmax = wave_max(exists ? s : -1e30f);
            const float mn = fmaxf(m[qi], tmax);
            const float alpha = expf(m[qi] - mn);
            const float p = exists ? expf(s - mn) : 0.f;
            l[qi] = l[qi] * alpha + wave_sum(p);
            o0[qi] *= alpha;
            o1[qi] *= alpha;
            m[qi] = mn;
            ps[wave * 4 + qi][lane] = p;
        }
        // PV for this tile (ps rows are private to the wave: no block barrier needed, but LDS writes must land)
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
        const int jn = min(MHA_KT, T - j0);
        for (int jj = 0; jj < jn; ++jj) {
            const float v0 = vs[jj][lane];
            const float v1 = (lane + 64 < dk) ? vs[jj][lane + 64] : 0.f;
#pragma unroll
            for (int qi = 0; qi < 4; ++qi) {
                const float p = ps[wave * 4 + qi][jj];
                o0[qi] += p * v0;
                o1[qi] += p * v1;
            }
        }
    }
#pragma unroll
    for (int qi = 0; qi < 4; ++qi) {
        const int i = q0 + wave * 4 + qi;
        if (i >= T) continue;
        float* dst = out + ((long long)b * T + i) * C + h * dk;
        const float inv = 1.0f / l[qi];
        if (lane < dk) dst[lane] = o0[qi] * inv;
        if (lane + 64 < dk) dst[lane + 64] = o1[qi] * inv;
    }
}
// ---- the same attention on the matrix cores (head size 96): exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) for both
// S^T = K Q^T and O = P V, flash-attention style.  Workgroup = 4 waves x 32 queries; keys / values stream through LDS
// in tiles of 32.  Layout trick: the score tile is computed TRANSPOSED (rows = keys, columns = queries), so that a lane
// owns one query: its 16 accumulator registers are 16 keys of that query (the other 16 sit in the partner lane +32), the
// online-softmax statistics are per-lane scalars, and the probabilities are already in the A-operand layout of the
// P V product (lane = query row, lane half = which key of the pair) - no cross-lane movement for P at all.  Only the
// per-query rescale factors have to cross lanes (O's rows are queries): 32 floats through a wave-private LDS array.
typedef __attribute__((ext_vector_type(16))) float f32x16m;
constexpr int MHX_DK = 96, MHX_KT = 32, MHX_PITCH = MHX_DK * 4 + 16;   // +16 B: rows land on different banks
__global__ __launch_bounds__(256) void mha_mfma_kernel(const float* qkv, float* out, const int* lens, int T, int C) {
    __shared__ __attribute__((aligned(16))) char ks[MHX_KT * MHX_PITCH];
    __shared__ __attribute__((aligned(16))) char vs[MHX_KT * MHX_PITCH];
    __shared__ __attribute__((aligned(16))) float bc[4][32];   // per-wave broadcast of a per-query scalar
    constexpr int DK = MHX_DK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, x = lane & 31, hf = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 128;
    const int len = lens ? min(max(lens[b], 0), T) : T;
    const float* base = qkv + (long long)b * T * 3 * C + h * DK;
    float* ob = out + (long long)b * T * C + h * DK;
    if (q0 >= len) {   // a tile of padded queries: every caller masks these rows
        for (int idx = tid; idx < 128 * (DK / 4); idx += 256) {
            const int i = idx / (DK / 4), c4 = idx % (DK / 4);
            if (q0 + i < T) *(f32x4*)(ob + (long long)(q0 + i) * C + c4 * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        return;
    }
    const float inv_sqrt = 1.0f / sqrtf((float)DK);
    const int q = q0 + wave * 32 + x;                // this lane's query
    // Q fragment: channels 8g + 4 hf .. +3 for g = 0..11 (both operands of an MFMA use the same channel for a lane half,
    // which is all the contraction needs), zero for queries past the utterance
    f32x4 qf[DK / 8];
#pragma unroll
    for (int g = 0; g < DK / 8; ++g)
        qf[g] = q < len ? *(const f32x4*)(base + (long long)q * 3 * C + 8 * g + 4 * hf) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x16m o[DK / 32];
#pragma unroll
    for (int dt = 0; dt < DK / 32; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;
    for (int j0 = 0; j0 < len; j0 += MHX_KT) {
        __syncthreads();
        {   // K / V rows of this tile, zeros past the end: all six 16 B loads of a thread in flight, then the LDS writes
            constexpr int NP = MHX_KT * (DK / 4) / 256;
            static_assert(NP * 256 == MHX_KT * (DK / 4), "whole passes");
            f32x4 kv[NP], vv[NP];
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                const int idx = tid + u * 256, j = idx / (DK / 4), c4 = idx % (DK / 4);
                kv[u] = vv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (j0 + j < len) {
                    const float* r = base + (long long)(j0 + j) * 3 * C + c4 * 4;
                    kv[u] = *(const f32x4*)(r + C);
                    vv[u] = *(const f32x4*)(r + 2 * C);
                }
            }
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                const int idx = tid + u * 256, j = idx / (DK / 4), c4 = idx % (DK / 4);
                *(f32x4*)(ks + j * MHX_PITCH + c4 * 16) = kv[u];
                *(f32x4*)(vs + j * MHX_PITCH + c4 * 16) = vv[u];
            }
        }
        __syncthreads();
        // S^T[key][query] = sum_c K[key][c] Q[query][c]
        f32x16m st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int g = 0; g < DK / 8; ++g) {
            const f32x4 kf = *(const f32x4*)(ks + x * MHX_PITCH + (8 * g + 4 * hf) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[g][e], st, 0, 0, 0);
        }
        // scores / sqrt(dk), masked_fill(mask == 0, -1e4); register r of this lane is key j0 + (r&3) + 8 (r>>2) + 4 hf
        float tmax = -1e30f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * hf;
            float sv = st[r] * inv_sqrt;
            if (!(q < len && j < len)) sv = -1e4f;
            st[r] = sv;
            tmax = fmaxf(tmax, sv);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mn = fmaxf(m_run, tmax);
        const float alpha = expf(m_run - mn);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pv = expf(st[r] - mn);
            st[r] = pv;
            ps += pv;
        }
        ps += __shfl_xor(ps, 32, 64);
        l_run = l_run * alpha + ps;
        m_run = mn;
        // rescale O (its rows are queries): alpha crosses lanes through the wave's LDS array
        if (hf == 0) bc[wave][x] = alpha;
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const f32x4 al = *(const f32x4*)&bc[wave][8 * r4 + 4 * hf];
#pragma unroll
            for (int dt = 0; dt < DK / 32; ++dt)
#pragma unroll
                for (int e = 0; e < 4; ++e) o[dt][4 * r4 + e] *= al[e];
        }
        // O[query][d] += sum_key P[query][key] V[key][d]: A = P (lane = query, half = key of the pair), B = V
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int jl = (r & 3) + 8 * (r >> 2) + 4 * hf;   // key of this lane half within the tile
#pragma unroll
            for (int dt = 0; dt < DK / 32; ++dt) {
                const float vv = *(const float*)(vs + jl * MHX_PITCH + (dt * 32 + x) * 4);
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(st[r], vv, o[dt], 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();   // bc is rewritten next tile
    }
    // out = O / l
    if (hf == 0) bc[wave][x] = 1.0f / l_run;
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const f32x4 il = *(const f32x4*)&bc[wave][8 * r4 + 4 * hf];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int qi = q0 + wave * 32 + 8 * r4 + 4 * hf + e;   // query of accumulator register 4 r4 + e
            if (qi >= T) continue;
#pragma unroll
            for (int dt = 0; dt < DK / 32; ++dt) ob[(long long)qi * C + dt * 32 + x] = o[dt][4 * r4 + e] * il[e];
        }
    }
}

// ---- the same kernel for FEW, LONG sequences (one 1000-character text at B = 1: 8 x 2 workgroups of the form above, each walking
// all 32 key tiles one after the other): a workgroup takes ONE tile of 32 queries and its four waves split the keys — 128 keys are
// staged per round, wave w attends to keys 32 w .. 32 w + 31 of the round — and the four partial (max, sum, O) triples are merged
// once at the end (the flash-decoding reduction, fixed order w = 0..3).  4x the workgroups, 1/4 of the serial chain per wave.
__global__ __launch_bounds__(256) void mha_mfma_split_kernel(const float* qkv, float* out, const int* lens, int T, int C) {
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    constexpr int DK = MHX_DK, TILE = MHX_KT * MHX_PITCH;
    char* ks_all = dyn;                          // [4][32 rows]
    char* vs_all = dyn + 4 * TILE;               // [4][32 rows]
    float (*bc)[32] = (float (*)[32])(dyn + 8 * TILE);   // per-wave broadcast of a per-query scalar; later the merge's statistics
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, x = lane & 31, hf = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 32;
    const int len = lens ? min(max(lens[b], 0), T) : T;
    const float* base = qkv + (long long)b * T * 3 * C + h * DK;
    float* ob = out + (long long)b * T * C + h * DK;
    if (q0 >= len) {   // a tile of padded queries: every caller masks these rows
        for (int idx = tid; idx < 32 * (DK / 4); idx += 256) {
            const int i = idx / (DK / 4), c4 = idx % (DK / 4);
            if (q0 + i < T) *(f32x4*)(ob + (long long)(q0 + i) * C + c4 * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        return;
    }
    const float inv_sqrt = 1.0f / sqrtf((float)DK);
    const int q = q0 + x;                        // this lane's query (the same 32 queries in every wave)
    f32x4 qf[DK / 8];
#pragma unroll
    for (int g = 0; g < DK / 8; ++g)
        qf[g] = q < len ? *(const f32x4*)(base + (long long)q * 3 * C + 8 * g + 4 * hf) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x16m o[DK / 32];
#pragma unroll
    for (int dt = 0; dt < DK / 32; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;
    const char* ks = ks_all + wave * TILE;
    const char* vs = vs_all + wave * TILE;
    for (int j00 = 0; j00 < len; j00 += 4 * MHX_KT) {
        __syncthreads();
        // K / V rows of this round (128 keys), zeros past the end: passes of 4 x 2 loads in flight per thread
        for (int base_idx = tid; base_idx < 4 * MHX_KT * (DK / 4); base_idx += 4 * 256) {
            f32x4 kv[4], vv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base_idx + u * 256, j = idx / (DK / 4), c4 = idx % (DK / 4);
                kv[u] = vv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (j00 + j < len) {
                    const float* r = base + (long long)(j00 + j) * 3 * C + c4 * 4;
                    kv[u] = *(const f32x4*)(r + C);
                    vv[u] = *(const f32x4*)(r + 2 * C);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = base_idx + u * 256, j = idx / (DK / 4), c4 = idx % (DK / 4);
                *(f32x4*)(ks_all + j * MHX_PITCH + c4 * 16) = kv[u];
                *(f32x4*)(vs_all + j * MHX_PITCH + c4 * 16) = vv[u];
            }
        }
        __syncthreads();
        const int j0 = j00 + wave * MHX_KT;
        if (j0 >= len) continue;                 // (uniform per wave; the barriers above are outside)
        f32x16m st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int g = 0; g < DK / 8; ++g) {
            const f32x4 kf = *(const f32x4*)(ks + x * MHX_PITCH + (8 * g + 4 * hf) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[g][e], st, 0, 0, 0);
        }
        float tmax = -1e30f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = j0 + (r & 3) + 8 * (r >> 2) + 4 * hf;
            float sv = st[r] * inv_sqrt;
            if (!(q < len && j < len)) sv = -1e4f;
            st[r] = sv;
            tmax = fmaxf(tmax, sv);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mn = fmaxf(m_run, tmax);
        const float alpha = expf(m_run - mn);
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pv = expf(st[r] - mn);
            st[r] = pv;
            ps += pv;
        }
        ps += __shfl_xor(ps, 32, 64);
        l_run = l_run * alpha + ps;
        m_run = mn;
        if (hf == 0) bc[wave][x] = alpha;
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const f32x4 al = *(const f32x4*)&bc[wave][8 * r4 + 4 * hf];
#pragma unroll
            for (int dt = 0; dt < DK / 32; ++dt)
#pragma unroll
                for (int e = 0; e < 4; ++e) o[dt][4 * r4 + e] *= al[e];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int jl = (r & 3) + 8 * (r >> 2) + 4 * hf;
#pragma unroll
            for (int dt = 0; dt < DK / 32; ++dt) {
                const float vv = *(const float*)(vs + jl * MHX_PITCH + (dt * 32 + x) * 4);
                o[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(st[r], vv, o[dt], 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();   // bc is rewritten next round
    }
    // ---- merge: M = max_w m_w ; every wave scales its O and l by exp(m_w - M) ; waves 1..3 hand theirs to wave 0 through LDS
    __syncthreads();                             // the K / V tiles are dead: their LDS takes the partial sums
    float* mstat = (float*)(dyn + 8 * TILE);     // [4][32] maxima, then [4][32] scaled sums behind them
    if (hf == 0) mstat[wave * 32 + x] = m_run;
    __syncthreads();
    const float M = fmaxf(fmaxf(mstat[x], mstat[32 + x]), fmaxf(mstat[64 + x], mstat[96 + x]));
    const float sc = expf(m_run - M);            // (a wave that saw no key: m = -1e30 -> 0)
    __syncthreads();                             // every wave has read the maxima
    if (hf == 0) {
        mstat[wave * 32 + x] = sc;               // this wave's per-query scale (crosses lanes like alpha above)
        mstat[128 + wave * 32 + x] = l_run * sc;
    }
    __syncthreads();
    float* part = (float*)dyn;                   // [3 waves][DK / 32][16][64]
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const f32x4 al = *(const f32x4*)&mstat[wave * 32 + 8 * r4 + 4 * hf];
#pragma unroll
        for (int dt = 0; dt < DK / 32; ++dt)
#pragma unroll
            for (int e = 0; e < 4; ++e) o[dt][4 * r4 + e] *= al[e];
    }
    if (wave > 0) {
#pragma unroll
        for (int dt = 0; dt < DK / 32; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) part[(((wave - 1) * (DK / 32) + dt) * 16 + r) * 64 + lane] = o[dt][r];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 1; w < 4; ++w)
#pragma unroll
        for (int dt = 0; dt < DK / 32; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] += part[(((w - 1) * (DK / 32) + dt) * 16 + r) * 64 + lane];
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int qq = 8 * r4 + 4 * hf + e;  // query of accumulator register 4 r4 + e
            const int qi = q0 + qq;
            if (qi >= T) continue;
            const float lsum = ((mstat[128 + qq] + mstat[160 + qq]) + mstat[192 + qq]) + mstat[224 + qq];
            const float il = 1.0f / lsum;
#pragma unroll
            for (int dt = 0; dt < DK / 32; ++dt) ob[(long long)qi * C + dt * 32 + x] = o[dt][4 * r4 + e] * il;
        }
    }
}

hipError_t mha_launch(const float* qkv, float* out, const int* lens, int B, int T, int C, int heads, hipStream_t s) {
    const int dk = C / heads;
    if (dk == MHX_DK) {
        // long sequences (T > 128): keys split over the waves (see mha_mfma_split_kernel; built for few long sequences, where the
        // 128-query form fills < half the CUs).  The rule depends on T ONLY — not on B or the device — because the two kernels merge their
        // partial softmax sums in different orders: an utterance must get the same encoder output alone and inside any batch
        static const bool split_ok = [] { const char* e = ablate_env("DTTS_MHA_SPLIT"); return !e || atoi(e) != 0; }();
        if (split_ok && T > 128) {
            constexpr int LDS = 8 * MHX_KT * MHX_PITCH + 8 * 32 * (int)sizeof(float);
            static bool configured_dev[64] = {};
            int cur_dev = 0;
            (void)hipGetDevice(&cur_dev);
            if (!configured_dev[cur_dev & 63]) {
                hipError_t e = hipFuncSetAttribute((const void*)mha_mfma_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
                if (e != hipSuccess) return e;
                configured_dev[cur_dev & 63] = true;
            }
            hipLaunchKernelGGL(mha_mfma_split_kernel, dim3((T + 31) / 32, heads, B), dim3(256), LDS, s, qkv, out, lens, T, C);
            return hipGetLastError();
        }
        hipLaunchKernelGGL(mha_mfma_kernel, dim3((T + 127) / 128, heads, B), dim3(256), 0, s, qkv, out, lens, T, C);
        return hipGetLastError();
    }
    if (dk > MHA_DK_MAX) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mha_kernel, dim3((T + MHA_QT - 1) / MHA_QT, heads, B), dim3(256), 0, s, qkv, out, lens, T, C, dk);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// S2PA dictionary attention: one block per word, gloss rows streamed once with 16 B/lane loads.
constexpr int S2PA_LMAX = 1024, S2PA_DMAX4 = 3;  // D <= 768 (3 float4 per lane), L_k <= 1024
#ifndef S2PA_RU_P
#define S2PA_RU_P 4
#endif
#ifndef S2PA_RU_T
#define S2PA_RU_T 4
#endif
constexpr int S2PA_NW = 4, S2PA_RU = S2PA_RU_T;           // waves per workgroup; gloss rows a wave keeps in flight (3 x 16 B per lane each)
constexpr int S2PA_NTHR = S2PA_NW * 64;

// The workgroup's LDS, carved from DYNAMIC shared memory sized for the launch's L_k and row width D (round 5): the static form was sized for
// L_k = 1024 and D = 768 (22.5 KB), which capped the kernel at 6 workgroups per CU — the 1,620 words of a B = 60 batch took a second,
// almost empty round of workgroups.  At L_k ~ 150, D = 192 a workgroup needs 5.6 KB and all words are resident at once (8 per CU).
struct S2paShared {
    float* lg;             // [Lp] logits -> weights
    float* km;             // [Lp] key_map row
    float* part;           // [NW][D] the waves' partial context sums (16-byte aligned rows)
    float* red;            // [16 NW]
    float* wmax;           // [NW]
    float* sense;          // [16]
    float* pw;             // [64]
    int* pid;              // [64 + 4]
    int* n_live;           // [1]
    unsigned short* idx;   // [Lp] list of the live rows
    int D;
};
__host__ __device__ inline size_t s2pa_lds_bytes(int L, int D) {
    const size_t Lp = (size_t)((L + 63) & ~63);
    return Lp * 4 * 2 + (size_t)S2PA_NW * D * 4 + (16 * S2PA_NW + S2PA_NW + 16 + 64 + 68 + 4) * 4 + Lp * 2;
}
__device__ __forceinline__ S2paShared s2pa_carve(char* base, int L, int D) {
    const int Lp = (L + 63) & ~63;
    S2paShared sh;
    float* f = (float*)base;
    sh.part = f;                 // first: the dynamic-LDS base is 16-byte aligned, every row is D * 4 bytes (D % 4 == 0)
    f += S2PA_NW * D;
    sh.lg = f;
    f += Lp;
    sh.km = f;
    f += Lp;
    sh.red = f;                  // [16 NW]: the block reductions use 2 NW of it, the sense merge all of it
    f += 16 * S2PA_NW;
    sh.wmax = f;
    f += S2PA_NW;
    sh.sense = f;
    f += 16;
    sh.pw = f;
    f += 64;
    sh.pid = (int*)f;
    f += 68;
    sh.n_live = (int*)f;
    f += 4;
    sh.idx = (unsigned short*)f;
    sh.D = D;
    return sh;
}
// where one word's rows come from: the collated tensors, or the resident table
struct S2paRow {
    const float* kmr = nullptr;
    const f32x4* kb = nullptr;
    const f32x4* vb = nullptr;
    const int64_t* pin = nullptr;
    const int64_t* pmr = nullptr;
    int Lrow, Prow, special;  // special: -1 / -2 in table mode
};
__device__ __forceinline__ S2paRow s2pa_row(const S2paArgs& a, int row) {
    S2paRow r;
    r.Lrow = a.L_k;
    r.Prow = a.P;
    r.special = 0;
    if (a.entry) {
        // (readfirstlane: the row is the workgroup's, but hipcc fetches these with vector loads; as VGPR values they would put every
        // pointer derived below into VGPR pairs)
        const int e = __builtin_amdgcn_readfirstlane(a.entry[row]);
        if (e >= 0) {
            const int o = __builtin_amdgcn_readfirstlane(a.t_off[e]);
            r.Lrow = min(__builtin_amdgcn_readfirstlane(a.t_off[e + 1]) - o, a.L_k);
            r.kmr = a.t_key_map + o;
            r.kb = (const f32x4*)(a.t_keys + (long long)o * a.D);
            r.vb = (const f32x4*)(a.t_values + (long long)o * a.D);
            const int po = __builtin_amdgcn_readfirstlane(a.t_poff[e]);
            r.Prow = min(__builtin_amdgcn_readfirstlane(a.t_poff[e + 1]) - po, a.P);
            r.pin = a.t_pinyin + po;
            r.pmr = a.t_pinyin_map + po;
        } else {
            r.special = e;
            r.Lrow = 0;
            r.Prow = 0;
        }
    } else {
        r.kmr = a.key_map + (long long)row * a.L_k;
        r.kb = (const f32x4*)(a.keys + (long long)row * a.L_k * a.D);
        r.vb = (const f32x4*)(a.values + (long long)row * a.L_k * a.D);
        r.pin = a.pinyin + (long long)row * a.P;
        r.pmr = a.pinyin_map + (long long)row * a.P;
    }
    return r;
}
// key_map row into LDS; every logit starts at its masked / zero-vector value; ordered list of the rows that must be
// READ (ballot prefix by wave 0), so that the streaming loops run over dense work.  Returns the number of live rows;
// when there is none (all logits -1e9: uniform softmax, every value row contributes) the list holds 0..Lrow-1.
__device__ __forceinline__ int s2pa_list(S2paShared& sh, const S2paRow& r, int L, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    for (int l = tid; l < L; l += S2PA_NTHR) {
        const float k = l < r.Lrow ? r.kmr[l] : (r.special == -1 ? 1.f : 0.f);
        sh.km[l] = k;
        sh.lg[l] = k == 0.f ? -1e9f : 0.f;   // key_map == 0: masked regardless of content; table-mode BOS / last row: zero gloss vector
    }
    __syncthreads();
    if (wave == 0) {
        int cnt = 0;
        for (int base = 0; base < r.Lrow; base += 64) {
            const int l = base + lane;
            const bool live = l < r.Lrow && sh.km[l] != 0.f;
            const unsigned long long m = __ballot(live);
            if (live) sh.idx[cnt + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)l;
            cnt += __popcll(m);
        }
        if (cnt == 0)
            for (int l = lane; l < r.Lrow; l += 64) sh.idx[l] = (unsigned short)l;
        if (lane == 0) *sh.n_live = cnt;
    }
    __syncthreads();
    return *sh.n_live;
}
__device__ __forceinline__ float s2pa_block_max(S2paShared& sh, float v, int tid) {
    v = wave_max(v);
    if ((tid & 63) == 0) sh.red[tid >> 6] = v;
    __syncthreads();
    v = sh.red[0];
#pragma unroll
    for (int w = 1; w < S2PA_NW; ++w) v = fmaxf(v, sh.red[w]);
    return v;
}
__device__ __forceinline__ float s2pa_block_sum(S2paShared& sh, float v, int tid) {
    v = wave_sum(v);
    if ((tid & 63) == 0) sh.red[S2PA_NW + (tid >> 6)] = v;
    __syncthreads();
    v = 0.f;
#pragma unroll
    for (int w = 0; w < S2PA_NW; ++w) v += sh.red[S2PA_NW + w];   // fixed order: reproducible
    return v;
}
// the small per-word inputs of the tail, loaded at kernel start so that the tail has no dependent global loads
struct S2paPre {
    long long pm, id, mod;
    int pm_max;
};
__device__ __forceinline__ S2paPre s2pa_prefetch(const S2paArgs& a, const S2paRow& r, int row, int tid) {
    S2paPre p;
    p.pm = 0;
    p.id = 0;
    if (tid < a.P && tid < 64) {
        p.pm = tid < r.Prow ? r.pmr[tid] : (r.special == -1 ? 1 : 0);
        p.id = tid < r.Prow ? r.pin[tid] : 0;
    }
    p.mod = a.pron_modified ? a.pron_modified[row] : 0;
    p.pm_max = *a.pm_max;
    return p;
}
// everything that needs only the normalised weights sh.lg[0..L) and sh.km: dict_attn (transposed, as the reference
// returns it), the sense merge, the forced-pronunciation rule and the pinyin-embedding mix
__device__ __forceinline__ void s2pa_tail(S2paShared& sh, const S2paArgs& a, const S2paPre& pre, int row, int b, int t, int tid) {
    const int L = a.L_k;
    // the word's weights as ONE contiguous row of the internal [B][T_w][L_k] tensor (round 5; the reference's transposed [B,1,L_k,T_w] view
    // is produced by dtts_text2mel_fetch: a word's column there is L_k scattered 4-byte stores, 243 k of them per B = 60 batch)
    float* da = a.dict_attn + (long long)row * L;
    for (int l = tid; l < L; l += S2PA_NTHR) da[l] = sh.lg[l];
    // sense weights s_i = sum_l w[l] [key_map == i]: thread 16 j + i sums the rows l = j (mod 16) of sense i (16 independent LDS streams of
    // L / 16 steps instead of one L-long dependent chain), the partials of a wave are joined by two shuffles, the four waves' in a fixed order
    {
        const int i = tid & 15, j = tid >> 4;
        float s = 0.f;
        if (i >= 1) {
#pragma unroll 4
            for (int l = j; l < L; l += 16) s += (sh.km[l] == (float)i) ? sh.lg[l] : 0.f;
        }
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if ((tid & 63) < 16) sh.red[(tid >> 6) * 16 + i] = s;   // (red: 4 x 16 floats here; its reduction uses are over)
    }
    __syncthreads();
    if (tid < 16) sh.sense[tid] = ((sh.red[tid] + sh.red[16 + tid]) + sh.red[32 + tid]) + sh.red[48 + tid];
    __syncthreads();
    if (tid < a.P && tid < 64) {
        const long long pm = pre.pm;
        float w = (pm >= 1 && pm < 16) ? sh.sense[pm] : 0.f;
        if (a.language_zh && a.pron_modified) {
            const long long mod = pre.mod;
            if (mod >= 1 && mod <= (long long)pre.pm_max) {
                const float forced = (pm == mod) ? 1.f : 0.f;
                w = (forced - w) + w;  // weights_ - weights.detach() + weights (layers/utils.py:114)
            }
        } else if (a.language_zh) {
            w = (w - w) + w;
        }
        sh.pw[tid] = w;
        a.pron_attn[(long long)row * a.P + tid] = w;
        long long id = pre.id;
        if (id < 0 || id >= a.n_pinyin) id = 0;
        sh.pid[tid] = (int)id;
    }
    __syncthreads();
    for (int c = tid; c < a.H; c += S2PA_NTHR) {
        float s = 0.f;
        for (int p0 = 0; p0 < a.P; p0 += 4) {   // 4 embedding rows in flight, summed in p order
            float e[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = p0 + j < a.P ? a.pinyin_emb[(long long)sh.pid[p0 + j] * a.H + c] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (p0 + j < a.P) s += sh.pw[p0 + j] * e[j];
        }
        a.pron[(long long)row * a.H + c] = s;
    }
}

// One workgroup per word.  (Splitting words with many live rows across workgroups, flash-decoding style, was tried:
// the main kernel drops from 85 to 67 us but the merge pass costs 10 us plus a launch gap, and merging inside the kernel
// by the chunk that arrives last needs device-scope fences that write back / invalidate a whole L2 on this multi-XCD
// part - 235 us.  Net zero by the stream's clock, so the simple form stays.)
// DM4: float4 pieces of a row per lane (3: the 768-wide gloss embeddings; 1: rows <= 256 wide = the PRE-PROJECTED table, K = key Wk^T /
// V = value Wv^T, 192 wide); RU: 2 x the rows of a chunk (a wave keeps RU / 2 key rows + RU / 2 value rows, or RU aliased rows, in flight).
#ifndef S2PA_WPE
#define S2PA_WPE 6   // waves per SIMD = workgroups per CU of the table form: 80 VGPRs without a spill (7: 2 spilled, same time; 8: 18 spilled, 2x slower)
#endif
#ifdef S2PA_STAMP   // phase stamps of every workgroup (tools/s2pa_stamps.py; a variant build, never the release library)
__device__ unsigned long long s2pa_stamps[8 * 4096];
extern "C" __attribute__((visibility("default"))) int dtts_debug_s2pa_stamps(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(s2pa_stamps), sizeof(s2pa_stamps));
}
#define S2_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096) s2pa_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define S2_STAMP(i)
#endif
// (the projected-table form, DM4 = 1: 8 waves per SIMD = 8 workgroups per CU, so that every word of a B = 60 batch is resident at once)
// SPEC: the resident-table form (a.entry != null): the entry's contiguous rows streamed speculatively (below); its own instantiation, so that
// the list-driven loops of the tensor API do not count against its registers
template <int DM4, int RU, bool SPEC>
#ifndef S2PA_WPE3
#define S2PA_WPE3 4
#endif
__global__ __launch_bounds__(S2PA_NTHR, DM4 == 1 ? S2PA_WPE : S2PA_WPE3) void s2pa_kernel(const S2paArgs a) {
    extern __shared__ __attribute__((aligned(16))) char s2pa_smem[];
    S2paShared sh = s2pa_carve(s2pa_smem, a.L_k, a.D);
    S2_STAMP(0);
    constexpr int S2PA_DMAX4 = DM4;   // (shadows the namespace constant: every per-lane row array below has DM4 pieces)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // workgroup i -> word (b, t) = (i % B, i / B): position-major.  The kernel lasts as long as its longest words (tools/s2pa_stamps.py), and in
    // the collated tensors those are the BOS rows (t = 0: key_map all ones, L_k "live" zero vectors — 130 rows against ~36 for a dictionary
    // word): they are dispatched FIRST and stream beside everything else instead of starting in the second round of workgroups.
    // ... and so are the LAST padded rows (t = T_w - 1: key_map all ones for every sentence, dataset_utils.py:288-300), which follow them at once:
    // dispatched last, these 60 words started at 29 us and ended the launch at 70 us (round 5 stamps)
    const int b = blockIdx.x % a.B, tq = blockIdx.x / a.B;
    const int t = tq == 0 ? 0 : (tq == 1 ? a.T_w - 1 : tq - 1);
    const int row = b * a.T_w + t;
    const int L = a.L_k, D4 = a.D / 4;
    const S2paRow r = s2pa_row(a, row);
    const S2paPre pre = s2pa_prefetch(a, r, row, tid);
    // the query, 12 floats per lane
    f32x4 q[S2PA_DMAX4];
    const f32x4* qp = (const f32x4*)(a.qk + (long long)row * a.D);
    if constexpr (!SPEC) {
#pragma unroll
        for (int cc = 0; cc < S2PA_DMAX4; ++cc) q[cc] = (lane + 64 * cc < D4) ? qp[lane + 64 * cc] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // a word past the end of its utterance: its weights are still returned (dict_attn), its context is zeroed by the
    // caller whatever the values are, so nothing is read for it
    const bool dead = a.lens && t >= a.lens[b];
    // Resident table (round 5): the entry's rows are CONTIGUOUS and ~88 % of them are live (key_map != 0), so they are streamed
    // speculatively, all of them, without waiting for the key_map row -> ballot list -> row addresses chain: one dependent global round
    // trip and two barriers less per word for 12 % more bytes, in a kernel that waits for latency at 12-15 % of the HBM rate.
    // A dead row's logit stays at its masked value and it is left out of the running sum.
    constexpr bool spec = SPEC;
    if constexpr (SPEC) {
        // batch padding (entry -2: 52 % of the B x T_w slots of a B = 60 Biaobei batch): key_map all zero -> every logit masked -> the
        // softmax is uniform (expf(0) / (L * 1.0f) in the general path below), no sense carries weight, the context of a word past its
        // utterance is zero.  Written directly: the general path spends 8 us of barriers and reductions per such word to get the same values.
        if (r.special == -2 && dead) {   // (pinyin_map is all zero there: the forced-pronunciation rule cannot select anything either)
            const float u = 1.0f / (float)L;
            float* da = a.dict_attn + (long long)row * L;
            for (int l = tid; l < L; l += S2PA_NTHR) da[l] = u;
            for (int d = tid; d < a.D; d += S2PA_NTHR) a.wv[(long long)row * a.D + d] = 0.f;
            if (tid < a.P) a.pron_attn[(long long)row * a.P + tid] = 0.f;
            for (int c = tid; c < a.H; c += S2PA_NTHR) a.pron[(long long)row * a.H + c] = 0.f;
            return;
        }
    }
    int n = 0;
    if constexpr (!SPEC) {
        n = s2pa_list(sh, r, L, tid);
        // a word past its utterance whose key_map row turned out all zero (the collater's padding; CHECKED here, the tensors are the caller's)
        // and that is not forced to a sense: uniform weights, no sense carries weight, zero context — the same values the general path below
        // computes with eight more barriers (52 % of the B x T_w slots of a B = 60 batch)
        if (n == 0 && dead && pre.mod == 0) {
            const float u = 1.0f / (float)L;
            float* da = a.dict_attn + (long long)row * L;
            for (int l = tid; l < L; l += S2PA_NTHR) da[l] = u;
            for (int d = tid; d < a.D; d += S2PA_NTHR) a.wv[(long long)row * a.D + d] = 0.f;
            if (tid < a.P) a.pron_attn[(long long)row * a.P + tid] = 0.f;
            for (int c = tid; c < a.H; c += S2PA_NTHR) a.pron[(long long)row * a.H + c] = 0.f;
            return;
        }
    }
    // ---- ONE streaming pass over the live rows: a wave takes RU listed rows at a time (12 independent 16-byte loads per lane
    // and row before the first reduction), computes their logits, and folds exp(logit - m) * value row into a running
    // weighted sum with a running maximum m (the online-softmax recurrence), so the value rows are not re-read after the
    // softmax: when the table stores value == key (the reference does, binarizer_zh.py:232-233) a row is read once.
    // The four waves' partial sums are merged with the block-level maximum / normaliser of the full softmax below, which is
    // still computed from the stored logits exactly as before (dict_attn, the sense merge and the pinyin mix use it).
    const bool alias = r.kb == r.vb;
    float m_run = -3.0e38f;
    f32x4 acc[S2PA_DMAX4];
#pragma unroll
    for (int cc = 0; cc < S2PA_DMAX4; ++cc) acc[cc] = f32x4{0.f, 0.f, 0.f, 0.f};
    // (NR = rows of the chunk: the arithmetic is that of round 3's RU-row fold with its zero rows left out — exp(.) * 0 and + 0 are exact
    // no-ops — so both input forms keep giving bit-identical results.  Forcing a fifth wave per SIMD on the 768-wide form (104 -> 96 VGPRs,
    // 29 spilled) so that all live words of a B = 60 batch are resident at once: 96 -> 99 us, not kept)
    auto fold_n = [&](const auto& lg, const auto& v, int cnt, auto nr_tag) {
        constexpr int NR = decltype(nr_tag)::value;
        float mn = m_run;
#pragma unroll
        for (int j = 0; j < NR; ++j)
            if (j < cnt) mn = fmaxf(mn, lg[j]);
        const float sc = expf(m_run - mn);
        float e[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) e[j] = j < cnt ? expf(lg[j] - mn) : 0.f;
#pragma unroll
        for (int cc = 0; cc < S2PA_DMAX4; ++cc) {
            acc[cc] = acc[cc] * sc;
#pragma unroll
            for (int j = 0; j < NR; ++j) acc[cc] += v[j][cc] * e[j];
        }
        m_run = mn;
    };

    if constexpr (!SPEC) S2_STAMP(1);
    constexpr int RV = RU / 2;   // rows folded together.  Row -> (wave, fold) assignment is the SAME with and without aliasing (chunk c
                                 // of RV rows goes to wave c % NW), so the two input forms give bit-identical results
    if constexpr (SPEC) {
        // ---- resident table, projected rows (D <= 256).  tools/s2pa_stamps.py (round 5): every workgroup of a B = 60 batch starts within
        // 0.7 us and the kernel lasts as long as its slowest words; their time was the row stream, and the row stream was VALU ISSUE
        // (6 waves per SIMD each spending ~75 instructions per row: a 64-lane dot product reduced by six shuffles, three full-precision
        // exponentials per 2-row chunk), not HBM latency — requesting chunk c + 1 before reducing chunk c changed nothing.  Here a wave
        // streams FOUR rows per step, 16 lanes per row: a lane holds PER 16-byte pieces of its row (3 at D = 192), the logit is reduced over
        // 16 lanes by four row rotations, and every 16-lane group keeps its OWN running maximum / weighted sum over the rows it sees
        // (no cross-group traffic inside the loop); the four groups of a wave meet once per word.  ~17 instructions per row.
        static_assert(DM4 == 1 && (RU == 3 || RU == 4), "the projected table: D <= 256");
        constexpr int PER = RU;                              // 16-byte pieces per lane (the RU parameter of this form): 3 for D <= 192, 4 for D <= 256
        const int nrow = r.Lrow;
        const int g = lane >> 4, c16 = lane & 15;            // row slot of this lane within a step, its first piece
        f32x4 qg[PER];
#pragma unroll
        for (int pc = 0; pc < PER; ++pc) qg[pc] = (c16 + 16 * pc < D4) ? qp[c16 + 16 * pc] : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 ag[PER];
#pragma unroll
        for (int pc = 0; pc < PER; ++pc) ag[pc] = f32x4{0.f, 0.f, 0.f, 0.f};
        float mg = -3.0e38f;
        int n_w = 0;
        // the key_map row -> LDS; every logit starts at its masked / zero-vector value (the stream below overwrites the live ones)
        f32x4 kq[PER], vq[PER];
        float kmq = 0.f;
        auto load_rows = [&](int i) {                        // rows i .. i + 3 of this wave
            const int l = i + g;
            const bool in = l < nrow;
            kmq = in ? r.kmr[l] : 0.f;
            const f32x4* kr = r.kb + (long long)(in ? l : 0) * D4;
            const f32x4* vr = r.vb + (long long)(in ? l : 0) * D4;
#pragma unroll
            for (int pc = 0; pc < PER; ++pc) {
                const bool ok = in && c16 + 16 * pc < D4;
                kq[pc] = ok ? kr[c16 + 16 * pc] : f32x4{0.f, 0.f, 0.f, 0.f};
                vq[pc] = (ok && !dead) ? vr[c16 + 16 * pc] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };
        constexpr int STEP = S2PA_NW * 4;
        int i = wave * 4;
        if (i < nrow) load_rows(i);                          // in flight while the key_map row goes to LDS
        for (int l = tid; l < L; l += S2PA_NTHR) {
            const float kk = l < r.Lrow ? r.kmr[l] : (r.special == -1 ? 1.f : 0.f);
            sh.km[l] = kk;
            sh.lg[l] = kk == 0.f ? -1e9f : 0.f;
        }
        __syncthreads();
        S2_STAMP(1);
        while (i < nrow) {
            float d = 0.f;
#pragma unroll
            for (int pc = 0; pc < PER; ++pc) d += kq[pc][0] * qg[pc][0] + kq[pc][1] * qg[pc][1] + kq[pc][2] * qg[pc][2] + kq[pc][3] * qg[pc][3];
            // sum over the 16 lanes of the row (DPP row rotations: every lane ends with the full sum)
            d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0x128, 0xf, 0xf, false));   // row_ror:8
            d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0x124, 0xf, 0xf, false));   // row_ror:4
            d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0x122, 0xf, 0xf, false));   // row_ror:2
            d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0x121, 0xf, 0xf, false));   // row_ror:1
            const int l = i + g;
            const bool valid = l < nrow && kmq != 0.f;
            if (valid && c16 == 0) sh.lg[l] = d;
            n_w += __popcll(__ballot(valid && c16 == 0));
            if (valid && !dead) {                            // (uniform per 16-lane group)
                const float mn = fmaxf(mg, d);
                const float sc = __expf(mg - mn), e = __expf(d - mn);
#pragma unroll
                for (int pc = 0; pc < PER; ++pc) ag[pc] = ag[pc] * sc + vq[pc] * e;
                mg = mn;
            }
            i += STEP;
            if (i < nrow) load_rows(i);                      // (latency is covered by the CU's other waves: up to 8 per SIMD)
        }
        // the wave's four row groups meet: (m, a) <- (max, a exp(m - max) + a' exp(m' - max)), partner 16 then 32 lanes away
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            const float mo = __shfl_xor(mg, o, 64);
            const float mn = fmaxf(mg, mo);
            const float sa = __expf(mg - mn), sb = __expf(mo - mn);   // (both groups empty: -3e38 - -3e38 = 0 -> 1 * 0 + 1 * 0)
#pragma unroll
            for (int pc = 0; pc < PER; ++pc)
#pragma unroll
                for (int x4 = 0; x4 < 4; ++x4) ag[pc][x4] = ag[pc][x4] * sa + __shfl_xor(ag[pc][x4], o, 64) * sb;
            mg = mn;
        }
        m_run = mg;
        if (g == 0) {
#pragma unroll
            for (int pc = 0; pc < PER; ++pc)
                if (c16 + 16 * pc < D4) *(f32x4*)&sh.part[wave * sh.D + (c16 + 16 * pc) * 4] = ag[pc];
        }
        if (lane == 0) sh.pid[64 + wave] = n_w;              // live rows this wave saw (joined behind the barrier below)
    } else {
    if (alias) {
        // two chunks (c, c + NW) of this wave in flight at once: 2 * RV rows = 12 independent 16-byte loads per lane
        for (int i = wave * RV; i < n; i += 2 * S2PA_NW * RV) {
            f32x4 k[2][RV][S2PA_DMAX4];
            int lr[2][RV];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < RV; ++j) {
                    lr[h][j] = sh.idx[min(i + h * S2PA_NW * RV + j, n - 1)];
                    const f32x4* kr = r.kb + (long long)lr[h][j] * D4;
#pragma unroll
                    for (int cc = 0; cc < S2PA_DMAX4; ++cc)
                        k[h][j][cc] = (lane + 64 * cc < D4) ? kr[lane + 64 * cc] : f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i0 = i + h * S2PA_NW * RV;
                if (i0 >= n) break;
                float lg[RV];
#pragma unroll
                for (int j = 0; j < RV; ++j) {
                    float d = 0.f;
#pragma unroll
                    for (int cc = 0; cc < S2PA_DMAX4; ++cc)
                        if (lane + 64 * cc < D4)
                            d += k[h][j][cc][0] * q[cc][0] + k[h][j][cc][1] * q[cc][1] + k[h][j][cc][2] * q[cc][2] + k[h][j][cc][3] * q[cc][3];
                    lg[j] = wave_sum(d);
                    if (lane == 0 && i0 + j < n) sh.lg[lr[h][j]] = lg[j];
                }
                if (!dead) fold_n(lg, k[h], min(n - i0, RV), std::integral_constant<int, RV>{});
            }
        }
    } else {   // keys and values in flight together: one chunk per step
        for (int i = wave * RV; i < n; i += S2PA_NW * RV) {
            f32x4 k[RV][S2PA_DMAX4], v[RV][S2PA_DMAX4];
            int lr[RV];
#pragma unroll
            for (int j = 0; j < RV; ++j) {
                lr[j] = sh.idx[min(i + j, n - 1)];
                const f32x4* kr = r.kb + (long long)lr[j] * D4;
                const f32x4* vr = r.vb + (long long)lr[j] * D4;
#pragma unroll
                for (int cc = 0; cc < S2PA_DMAX4; ++cc) {
                    k[j][cc] = (lane + 64 * cc < D4) ? kr[lane + 64 * cc] : f32x4{0.f, 0.f, 0.f, 0.f};
                    v[j][cc] = (lane + 64 * cc < D4 && !dead) ? vr[lane + 64 * cc] : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            float lg[RV];
#pragma unroll
            for (int j = 0; j < RV; ++j) {
                float d = 0.f;
#pragma unroll
                for (int cc = 0; cc < S2PA_DMAX4; ++cc)
                    if (lane + 64 * cc < D4) d += k[j][cc][0] * q[cc][0] + k[j][cc][1] * q[cc][1] + k[j][cc][2] * q[cc][2] + k[j][cc][3] * q[cc][3];
                lg[j] = wave_sum(d);
                if (lane == 0 && i + j < n) sh.lg[lr[j]] = lg[j];
            }
            if (!dead) fold_n(lg, v, min(n - i, RV), std::integral_constant<int, RV>{});
        }
    }
    }   // (!SPEC)
    if constexpr (!SPEC) {
#pragma unroll
        for (int cc = 0; cc < S2PA_DMAX4; ++cc)
            if (lane + 64 * cc < D4) *(f32x4*)&sh.part[wave * sh.D + (lane + 64 * cc) * 4] = acc[cc];
    }
    if (lane == 0) sh.wmax[wave] = m_run;
    S2_STAMP(2);
    __syncthreads();
    S2_STAMP(3);
    if (spec) n = (sh.pid[64] + sh.pid[65]) + (sh.pid[66] + sh.pid[67]);
    // softmax over l (all L logits: live, masked -1e9, zero-vector 0)
    float mx = -3.0e38f;
    for (int l = tid; l < L; l += S2PA_NTHR) mx = fmaxf(mx, sh.lg[l]);
    mx = s2pa_block_max(sh, mx, tid);
    float sm = 0.f;
    for (int l = tid; l < L; l += S2PA_NTHR) {
        const float e = expf(sh.lg[l] - mx);
        sh.lg[l] = e;
        sm += e;
    }
    sm = s2pa_block_sum(sh, sm, tid);
    for (int l = tid; l < L; l += S2PA_NTHR) sh.lg[l] = sh.lg[l] / sm;
    __syncthreads();
    S2_STAMP(4);
    if (n > 0 || dead) {
        // context = sum_w exp(m_w - mx) * partial_w / sm  (a wave that saw no row has m_w = -3e38: factor 0)
        float f[S2PA_NW];
#pragma unroll
        for (int w = 0; w < S2PA_NW; ++w) f[w] = expf(sh.wmax[w] - mx) / sm;
        for (int d = tid; d < a.D; d += S2PA_NTHR) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < S2PA_NW; ++w) sum += sh.part[w * sh.D + d] * f[w];   // fixed order: reproducible
            a.wv[(long long)row * a.D + d] = dead ? 0.f : sum;
        }
    } else {
        // no live row: every logit is masked, the softmax is uniform and every value row of the word contributes (as in the
        // reference); rare (a fully padded word row inside an utterance), kept as a separate weighted pass
        const int n_val = r.Lrow;
        f32x4 va[S2PA_DMAX4];
#pragma unroll
        for (int cc = 0; cc < S2PA_DMAX4; ++cc) va[cc] = f32x4{0.f, 0.f, 0.f, 0.f};
        // four rows of a wave in flight at once (round 5: one row at a time was a chain of L_k / 4 dependent HBM round trips, 58 us for the
        // true-EOS row of every shorter sentence in the collated tensors — the slowest workgroups of the tensor-API launch); summed in l order
        for (int l0 = wave; l0 < n_val; l0 += 4 * S2PA_NW) {
            f32x4 vv[4][S2PA_DMAX4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int l = l0 + j * S2PA_NW;
                const f32x4* vr = r.vb + (long long)(l < n_val ? l : 0) * D4;
#pragma unroll
                for (int cc = 0; cc < S2PA_DMAX4; ++cc)
                    vv[j][cc] = (l < n_val && lane + 64 * cc < D4) ? vr[lane + 64 * cc] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int l = l0 + j * S2PA_NW;
                if (l >= n_val) break;
                const float w = sh.lg[l];
#pragma unroll
                for (int cc = 0; cc < S2PA_DMAX4; ++cc) va[cc] += vv[j][cc] * w;
            }
        }
        __syncthreads();
#pragma unroll
        for (int cc = 0; cc < S2PA_DMAX4; ++cc)
            if (lane + 64 * cc < D4) *(f32x4*)&sh.part[wave * sh.D + (lane + 64 * cc) * 4] = va[cc];
        __syncthreads();
        for (int d = tid; d < a.D; d += S2PA_NTHR) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < S2PA_NW; ++w) sum += sh.part[w * sh.D + d];
            a.wv[(long long)row * a.D + d] = sum;
        }
    }
    S2_STAMP(5);
    s2pa_tail(sh, a, pre, row, b, t, tid);
    S2_STAMP(6);
}

hipError_t s2pa_launch(const S2paArgs& a, hipStream_t s) {
    if (a.L_k > S2PA_LMAX || a.D > 768 || (a.D & 3) || a.P > 64) return hipErrorInvalidValue;
    const size_t lds = s2pa_lds_bytes(a.L_k, a.D);   // <= 22.6 KB (L_k = 1024, D = 768): below the 64 KB that need no attribute
    const dim3 g(a.B * a.T_w), t(S2PA_NTHR);
    if (a.entry && a.D <= 192) {   // resident table of projected rows: the speculative 16-lanes-per-row stream, 3 pieces per lane
        hipLaunchKernelGGL((s2pa_kernel<1, 3, true>), g, t, lds, s, a);
    } else if (a.entry && a.D <= 256) {
        hipLaunchKernelGGL((s2pa_kernel<1, 4, true>), g, t, lds, s, a);
    } else {   // collated tensors — or, ablation builds (tune bit 6), a resident table of raw 768-wide gloss rows
        if (a.D <= 256) hipLaunchKernelGGL((s2pa_kernel<1, S2PA_RU_P, false>), g, t, lds, s, a);
        else hipLaunchKernelGGL((s2pa_kernel<S2PA_DMAX4, S2PA_RU, false>), g, t, lds, s, a);
    }
    return hipGetLastError();
}

// ---- FFTBlocks input stage: positions by a block scan over one utterance's first channel, then a row-parallel
// embedding add + mask
__global__ __launch_bounds__(256) void fft_positions_kernel(const float* x, int* pos, int n_pos, int T, int C) {
    __shared__ int wsum[4];
    __shared__ int carry;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xb = x + (long long)b * T * C;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int t0 = 0; t0 < T; t0 += 256) {
        const int t = t0 + tid;
        const bool nz = t < T && xb[(long long)t * C] != 0.f;
        const unsigned long long m = __ballot(nz);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(m);
        __syncthreads();
        int base = carry;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        int p = nz ? base + before + 1 : 0;
        if (p >= n_pos) p = 0;   // cannot happen when n_pos > T; keeps the gather in range
        if (t < T) pos[(long long)b * T + t] = p;
        __syncthreads();
        if (tid == 0) carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void fft_input_kernel(const float* x, const float* table, const int* pos, const float* alpha,
                                                        const int* lens, float* y, long long rows, int T, int C) {
    const long long row = blockIdx.x * 4ll + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int b = (int)(row / T), t = (int)(row % T);
    const bool keep = !lens || t < lens[b];
    const float al = alpha ? *alpha : 1.f;
    const float* tr = table ? table + (long long)pos[row] * C : nullptr;
    for (int c = lane; c < C; c += 64) {
        float v = x[row * C + c];
        if (tr) v = __fadd_rn(v, __fmul_rn(al, tr[c]));   // two roundings, as x + alpha * positions
        y[row * C + c] = keep ? v : 0.f;
    }
}
hipError_t fft_input_launch(const float* x, const float* table, int n_pos, const float* alpha, const int* lens, int* pos_scratch,
                            float* y, int B, int T, int C, hipStream_t s) {
    const long long rows = (long long)B * T;
    if (table) hipLaunchKernelGGL(fft_positions_kernel, dim3(B), dim3(256), 0, s, x, pos_scratch, n_pos, T, C);
    hipLaunchKernelGGL(fft_input_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, table, pos_scratch, alpha, lens, y, rows, T, C);
    return hipGetLastError();
}

// ---- waveform -> int16 (utils/audio.py:11-16)
__global__ void wav_absmax_kernel(const float* wav, const int* lens, int hop, long long N, unsigned* amax_bits) {
    const int b = blockIdx.y;
    const long long n = lens ? min((long long)max(lens[b], 0) * hop, N) : N;
    const float* w = wav + (long long)b * N;
    float m = 0.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(w[i]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(amax_bits + b, __float_as_uint(m));  // non-negative floats order as their bits
}
__global__ void wav_to_int16_kernel(const float* wav, const int* lens, int hop, long long N, int norm, const unsigned* amax_bits,
                                    int16_t* out) {
    const int b = blockIdx.y;
    const long long n = lens ? min((long long)max(lens[b], 0) * hop, N) : N;
    const float* w = wav + (long long)b * N;
    int16_t* o = out + (long long)b * N;
    const float m = norm ? __uint_as_float(amax_bits[b]) : 1.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) {
        float v = i < n ? w[i] : 0.f;
        if (norm) v = v / m;               // wav / np.abs(wav).max(): IEEE fp32 division, as numpy
        v = v * 32767.f;                    // wav *= 32767 (fp32)
        o[i] = (int16_t)(int)v;             // astype(np.int16): truncation toward zero (|v| <= 32767 on this path)
    }
}
hipError_t wav_to_int16_launch(const float* wav, const int* lens, int hop, int B, long long N, int norm, unsigned* amax_bits,
                               int16_t* out, hipStream_t s) {
    const int bx = (int)std::min<long long>((N + 256 * 8 - 1) / (256 * 8), 1024);
    if (norm) {
        hipError_t e = hipMemsetAsync(amax_bits, 0, sizeof(unsigned) * B, s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(wav_absmax_kernel, dim3(bx, B), dim3(256), 0, s, wav, lens, hop, N, amax_bits);
    }
    hipLaunchKernelGGL(wav_to_int16_kernel, dim3(bx, B), dim3(256), 0, s, wav, lens, hop, N, norm, amax_bits, out);
    return hipGetLastError();
}

__global__ void max_i64_kernel(const int64_t* x, long long n, int* out) {
    int m = 0;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        m = max(m, (int)x[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}
__global__ void max_entry_pm_kernel(const int* entry, const int* t_pmmax, long long n, int* out) {
    int m = 0;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int e = entry[i];
        m = max(m, e >= 0 ? t_pmmax[e] : (e == -1 ? 1 : 0));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}
hipError_t max_entry_pm_launch(const int* entry, const int* t_pmmax, long long n, int* out, hipStream_t s) {
    hipError_t e = hipMemsetAsync(out, 0, sizeof(int), s);
    if (e != hipSuccess) return e;
    const int blocks = (int)((n + 255) / 256 > 256 ? 256 : (n + 255) / 256);
    hipLaunchKernelGGL(max_entry_pm_kernel, dim3(blocks), dim3(256), 0, s, entry, t_pmmax, n, out);
    return hipGetLastError();
}
hipError_t max_i64_launch(const int64_t* x, long long n, int* out, hipStream_t s) {
    hipError_t e = hipMemsetAsync(out, 0, sizeof(int), s);
    if (e != hipSuccess) return e;
    const int blocks = (int)((n + 255) / 256 > 256 ? 256 : (n + 255) / 256);
    hipLaunchKernelGGL(max_i64_kernel, dim3(blocks), dim3(256), 0, s, x, n, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
__global__ void add_kernel(const f32x4* a, const f32x4* b, f32x4* y, long long n4) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x)
        y[i] = a[i] + b[i];
}
hipError_t add_launch(const float* a, const float* b, float* y, long long n, hipStream_t s) {
    const long long n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 > 2048 ? 2048 : (n4 + 255) / 256);
    hipLaunchKernelGGL(add_kernel, dim3(blocks), dim3(256), 0, s, (const f32x4*)a, (const f32x4*)b, (f32x4*)y, n4);
    return hipGetLastError();
}

__global__ void mask_rows_kernel(const float* x, float* y, const int* lens, int T, int C, int rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const bool keep = (row % T) < lens[row / T];
    for (int c = lane; c < C; c += 64) y[(long long)row * C + c] = keep ? x[(long long)row * C + c] : 0.f;
}
hipError_t mask_rows_launch(const float* x, float* y, const int* lens, int B, int T, int C, hipStream_t s) {
    hipLaunchKernelGGL(mask_rows_kernel, dim3((B * T + 3) / 4), dim3(256), 0, s, x, y, lens, T, C, B * T);
    return hipGetLastError();
}

// 64 rows per workgroup, one atomicAdd per workgroup into ilens[b] (zeroed by the launcher)
__global__ void rowcount_nonzero_kernel(const float* x, int* ilens, int T, int C) {
    __shared__ int cnt;
    const int b = blockIdx.y, t0 = blockIdx.x * 64, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    const int t1 = min(t0 + 64, T);
    for (int t = t0 + wave; t < t1; t += 4) {
        const float* r = x + ((long long)b * T + t) * C;
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += fabsf(r[c]);
        s = wave_sum(s);
        if (lane == 0 && s != 0.f) atomicAdd(&cnt, 1);
    }
    __syncthreads();
    if (threadIdx.x == 0 && cnt) atomicAdd(ilens + b, cnt);
}
hipError_t rowcount_nonzero_launch(const float* x, int* ilens, int B, int T, int C, hipStream_t s) {
    hipError_t e = hipMemsetAsync(ilens, 0, sizeof(int) * B, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(rowcount_nonzero_kernel, dim3((T + 63) / 64, B), dim3(256), 0, s, x, ilens, T, C);
    return hipGetLastError();
}

__global__ void dur_head_kernel(const float* h, const float* w, const float* bias, const int* ilens, float* dur, int T,
                                int C, int rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* r = h + (long long)row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += r[c] * w[c];
    s = wave_sum(s) + bias[0];
    const float sp = s > 20.f ? s : log1pf(expf(s));  // F.softplus, beta = 1, threshold = 20
    if (lane == 0) dur[row] = ((row % T) < ilens[row / T]) ? sp : 0.f;
}
hipError_t dur_head_launch(const float* h, const float* w, const float* bias, const int* ilens, float* dur, int B, int T,
                           int C, hipStream_t s) {
    hipLaunchKernelGGL(dur_head_kernel, dim3((B * T + 3) / 4), dim3(256), 0, s, h, w, bias, ilens, dur, T, C, B * T);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
__global__ void durations_kernel(const float* dur, const int* ilens, int* starts, int* total, int T) {
    const int b = blockIdx.x;
    const int n = ilens[b];
    int* st = starts + (long long)b * (T + 1);
    // integer durations, then an exclusive scan (one wave; T_w is a few dozen words, 1k for long-form)
    int carry = 0, all = 0;
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + threadIdx.x;
        int d = 0;
        if (t < n) {
            float v = rintf(expf(dur[(long long)b * T + t]) - 1.0f);  // torch.round: half to even
            v = fminf(fmaxf(v, 0.f), 1.0e6f);
            d = (int)v;
        }
        all += d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) all += __shfl_xor(all, o, 64);
    const bool fill = (all == 0);  // "all of the predicted durations are 0. fill 0 with 1." (tts_modules.py:248-250)
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + threadIdx.x;
        int d = 0;
        if (t < n) {
            float v = rintf(expf(dur[(long long)b * T + t]) - 1.0f);
            v = fminf(fmaxf(v, 0.f), 1.0e6f);
            d = fill ? 1 : (int)v;
        }
        int inc = d;  // inclusive scan across the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(inc, o, 64);
            if ((int)threadIdx.x >= o) inc += u;
        }
        if (t < T) st[t] = carry + inc - d;
        carry += __shfl(inc, 63, 64);
    }
    if (threadIdx.x == 0) {
        st[T] = carry;
        total[b] = carry;
    }
}
hipError_t durations_launch(const float* dur, const int* ilens, int* starts, int* total, int B, int T, hipStream_t s) {
    hipLaunchKernelGGL(durations_kernel, dim3(B), dim3(64), 0, s, dur, ilens, starts, total, T);
    return hipGetLastError();
}

// total[b]: in = sum of the integer durations; out = number of frames with mel2word > 0 AFTER the padding to frames_multiple
// (the padded columns repeat the last column, modules/dict_tts/model.py:98-100: an utterance that reaches T_raw keeps
// them as valid frames, exactly as the reference's B = 1 inference vocodes all T_mel frames, tasks/tts/dict_tts.py:255)
__global__ void mel2word_fill_kernel(const int* starts, int* total, int64_t* m2w, int T_w, int T_raw, int T_mel) {
    const int b = blockIdx.x;
    const int* st = starts + (long long)b * (T_w + 1);
    int64_t* row = m2w + (long long)b * T_mel;
    const int tot = total[b];
    for (int f = tot + threadIdx.x; f < T_raw; f += blockDim.x) row[f] = 0;
    for (int t = threadIdx.x; t < T_w; t += blockDim.x) {
        const int s0 = st[t], s1 = st[t + 1];
        for (int f = s0; f < s1 && f < T_raw; ++f) row[f] = t + 1;
    }
    __syncthreads();
    const int64_t last = T_raw > 0 ? row[T_raw - 1] : 0;
    for (int f = T_raw + threadIdx.x; f < T_mel; f += blockDim.x) row[f] = last;
    if (threadIdx.x == 0 && last > 0) total[b] = tot + (T_mel - T_raw);
}
hipError_t mel2word_fill_launch(const int* starts, int* total, const int* ilens, int64_t* m2w, int B, int T_w,
                                int T_raw, int T_mel, hipStream_t s) {
    (void)ilens;
    hipLaunchKernelGGL(mel2word_fill_kernel, dim3(B), dim3(256), 0, s, starts, total, m2w, T_w, T_raw, T_mel);
    return hipGetLastError();
}

__global__ void mel2word_copy_kernel(const int64_t* src, int64_t* dst, int* total, int T_in, int T_mel) {
    __shared__ int cnt;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    int n = 0;
    for (int f = threadIdx.x; f < T_mel; f += blockDim.x) {
        const int64_t v = src[(long long)b * T_in + (f < T_in ? f : T_in - 1)];
        dst[(long long)b * T_mel + f] = v;
        n += v > 0 ? 1 : 0;
    }
    atomicAdd(&cnt, n);
    __syncthreads();
    if (threadIdx.x == 0) total[b] = cnt;
}
hipError_t mel2word_copy_launch(const int64_t* src, int64_t* dst, int* total, int B, int T_in, int T_mel, hipStream_t s) {
    hipLaunchKernelGGL(mel2word_copy_kernel, dim3(B), dim3(256), 0, s, src, dst, total, T_in, T_mel);
    return hipGetLastError();
}

__global__ void expand_kernel(const float* weo, const int64_t* m2w, float* x, float* x_mask, const float* zero_row, int T_w,
                              int T_mel, int C, int rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int b = row / T_mel;
    const long long w = m2w[row];
    f32x4* dst = (f32x4*)(x + (long long)row * C);
    if (w > 0 && w <= T_w) {
        const f32x4* src = (const f32x4*)(weo + ((long long)b * T_w + (w - 1)) * C);
        for (int c = lane; c < C / 4; c += 64) dst[c] = src[c];
    } else if (zero_row) {
        for (int c = lane; c < C / 4; c += 64) dst[c] = ((const f32x4*)zero_row)[c];
    } else {
        for (int c = lane; c < C / 4; c += 64) dst[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (x_mask && lane == 0) x_mask[row] = w > 0 ? 1.f : 0.f;
}
hipError_t expand_launch(const float* weo, const int64_t* m2w, float* x, float* x_mask, int B, int T_w, int T_mel, int C,
                         hipStream_t s, const float* zero_row) {
    hipLaunchKernelGGL(expand_kernel, dim3((B * T_mel + 3) / 4), dim3(256), 0, s, weo, m2w, x, x_mask, zero_row, T_w, T_mel, C,
                       B * T_mel);
    return hipGetLastError();
}

__global__ void transpose_cf_to_cl_kernel(const float* x, float* y, int C, int T, int ldT, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long long bt = i / C;
        const long long b = bt / T, t = bt % T;
        y[i] = x[(b * C + c) * ldT + t];
    }
}
hipError_t transpose_cf_to_cl_launch(const float* x, float* y, int B, int C, int T, hipStream_t s, int ldT) {
    const long long n = (long long)B * C * T;
    const int blocks = (int)std::min<long long>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(transpose_cf_to_cl_kernel, dim3(blocks), dim3(256), 0, s, x, y, C, T, ldT > 0 ? ldT : T, n);
    return hipGetLastError();
}

// standard-normal fill (counter-based: splitmix64 of (seed, index) -> two uniforms -> Box-Muller).  Used when the
// caller passes no prior sample: the reference draws N(0,1) from torch's CPU generator (fvae_semantics.py:110-111), which
// no device code can reproduce bit for bit; callers that need the reference's noise pass z_p.
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__global__ void normal_fill_kernel(float* y, long long n, unsigned long long seed) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const unsigned long long r = splitmix64(seed * 0xD1342543DE82EF95ull + (unsigned long long)i);
        const float u1 = ((float)(unsigned)(r >> 40) + 0.5f) * (1.0f / 16777216.0f);   // (0,1)
        const float u2 = ((float)(unsigned)((r >> 8) & 0xFFFFFF) + 0.5f) * (1.0f / 16777216.0f);
        y[i] = sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
    }
}
hipError_t normal_fill_launch(float* y, long long n, unsigned long long seed, hipStream_t s) {
    const int blocks = (int)std::min<long long>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(normal_fill_kernel, dim3(blocks), dim3(256), 0, s, y, n, seed);
    return hipGetLastError();
}

} // namespace dtts
