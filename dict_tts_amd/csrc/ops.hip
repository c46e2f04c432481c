// Non-GEMM kernels of the Dict-TTS path.  See ops.h for the contracts and the reference lines they follow.
#include "ops.h"

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
    const int len = lens ? lens[b] : T;
    const float* base = qkv + (long long)b * T * 3 * C;
    const float inv_sqrt = 1.0f / sqrtf((float)dk);
    for (int idx = tid; idx < MHA_QT * dk; idx += 256) {
        const int i = idx / dk, d = idx % dk;
        qs[i][d] = (q0 + i < T) ? base[(long long)(q0 + i) * 3 * C + h * dk + d] : 0.f;
    }
    float m[4], l[4], o0[4], o1[4];
#pragma unroll
    for (int qi = 0; qi < 4; ++qi) { m[qi] = -1e30f; l[qi] = 0.f; o0[qi] = 0.f; o1[qi] = 0.f; }
    for (int j0 = 0; j0 < T; j0 += MHA_KT) {
        __syncthreads();
        for (int idx = tid; idx < MHA_KT * dk; idx += 256) {
            const int j = idx / dk, d = idx % dk;
            const bool ok = j0 + j < T;
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
hipError_t mha_launch(const float* qkv, float* out, const int* lens, int B, int T, int C, int heads, hipStream_t s) {
    const int dk = C / heads;
    if (dk > MHA_DK_MAX) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mha_kernel, dim3((T + MHA_QT - 1) / MHA_QT, heads, B), dim3(256), 0, s, qkv, out, lens, T, C, dk);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// S2PA dictionary attention: one block per word, gloss rows streamed once with 16 B/lane loads.
constexpr int S2PA_LMAX = 1024, S2PA_DMAX4 = 3;  // D <= 768 (3 float4 per lane), L_k <= 1024
// S2PA_NW waves per word, each keeping S2PA_RU gloss rows (3 x 16 B per lane each) in flight
template <int S2PA_NW, int S2PA_RU>
__global__ __launch_bounds__(S2PA_NW * 64) void s2pa_kernel(const S2paArgs a) {
    __shared__ float lg[S2PA_LMAX];
    __shared__ float km[S2PA_LMAX];
    __shared__ __attribute__((aligned(16))) float part[S2PA_NW][768];
    __shared__ float red[2 * S2PA_NW];
    __shared__ float sense[16];
    __shared__ float pw[64];
    __shared__ int pid[64 + 4];
    constexpr int NTHR = S2PA_NW * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = blockIdx.x;  // b * T_w + t
    const int b = row / a.T_w, t = row % a.T_w;
    const int L = a.L_k, D4 = a.D / 4;
    // row sources: the collated tensors, or the resident table
    const float* kmr = nullptr;
    const f32x4* kb = nullptr;
    const f32x4* vb = nullptr;
    const int64_t* pin = nullptr;
    const int64_t* pmr = nullptr;
    int Lrow = L, Prow = a.P;
    int special = 0;  // -1 / -2 in table mode
    if (a.entry) {
        const int e = a.entry[row];
        if (e >= 0) {
            const int o = a.t_off[e];
            Lrow = min(a.t_off[e + 1] - o, L);
            kmr = a.t_key_map + o;
            kb = (const f32x4*)(a.t_keys + (long long)o * a.D);
            vb = (const f32x4*)(a.t_values + (long long)o * a.D);
            const int po = a.t_poff[e];
            Prow = min(a.t_poff[e + 1] - po, a.P);
            pin = a.t_pinyin + po;
            pmr = a.t_pinyin_map + po;
        } else {
            special = e;
            Lrow = 0;
            Prow = 0;
        }
    } else {
        kmr = a.key_map + (long long)row * L;
        kb = (const f32x4*)(a.keys + (long long)row * L * a.D);
        vb = (const f32x4*)(a.values + (long long)row * L * a.D);
        pin = a.pinyin + (long long)row * a.P;
        pmr = a.pinyin_map + (long long)row * a.P;
    }
    // key_map row; every logit starts at its masked / zero-vector value, only rows that must be READ are listed
    __shared__ unsigned short idx[S2PA_LMAX];
    __shared__ int n_live;
    for (int l = tid; l < L; l += NTHR) {
        const float k = l < Lrow ? kmr[l] : (special == -1 ? 1.f : 0.f);
        km[l] = k;
        lg[l] = k == 0.f ? -1e9f : 0.f;   // key_map == 0: masked regardless of content; table-mode BOS / last row: zero gloss vector
    }
    // the query, 12 floats per lane
    f32x4 q[S2PA_DMAX4];
    const f32x4* qp = (const f32x4*)(a.qk + (long long)row * a.D);
#pragma unroll
    for (int c = 0; c < S2PA_DMAX4; ++c) q[c] = (lane + 64 * c < D4) ? qp[lane + 64 * c] : f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    if (wave == 0) {   // ordered compaction of the unmasked rows (ballot prefix), so that the streaming loops below
        int cnt = 0;   // run over dense work and can keep several rows in flight
        for (int base = 0; base < Lrow; base += 64) {
            const int l = base + lane;
            const bool live = l < Lrow && km[l] != 0.f;
            const unsigned long long m = __ballot(live);
            if (live) idx[cnt + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)l;
            cnt += __popcll(m);
        }
        if (cnt == 0)   // every row masked: all logits are -1e9, the softmax is uniform and every value row contributes
            for (int l = lane; l < Lrow; l += 64) idx[l] = (unsigned short)l;
        if (lane == 0) n_live = cnt;
    }
    __syncthreads();
    const int n = n_live;
    // a word past the end of its utterance: its weights are still returned (dict_attn), its context is zeroed by the
    // caller whatever the values are, so nothing is read for it
    const bool dead = a.lens && t >= a.lens[b];
    const int n_val = dead ? 0 : (n ? n : Lrow);
    // logits: a wave takes 4 listed rows at a time - 12 independent 16-byte loads per lane before the first reduction
    constexpr int RU = S2PA_RU;
    for (int i = wave * RU; i < n; i += S2PA_NW * RU) {
        f32x4 k[RU][S2PA_DMAX4];
        int lr[RU];
#pragma unroll
        for (int j = 0; j < RU; ++j) {
            lr[j] = idx[min(i + j, n - 1)];
            const f32x4* kr = kb + (long long)lr[j] * D4;
#pragma unroll
            for (int c = 0; c < S2PA_DMAX4; ++c)
                k[j][c] = (lane + 64 * c < D4) ? kr[lane + 64 * c] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < RU; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < S2PA_DMAX4; ++c)
                if (lane + 64 * c < D4) acc += k[j][c][0] * q[c][0] + k[j][c][1] * q[c][1] + k[j][c][2] * q[c][2] + k[j][c][3] * q[c][3];
            acc = wave_sum(acc);
            if (lane == 0 && i + j < n) lg[lr[j]] = acc;
        }
    }
    __syncthreads();
    // softmax over l
    float mx = -3.0e38f;
    for (int l = tid; l < L; l += NTHR) mx = fmaxf(mx, lg[l]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int w = 1; w < S2PA_NW; ++w) mx = fmaxf(mx, red[w]);
    float sm = 0.f;
    for (int l = tid; l < L; l += NTHR) {
        const float e = expf(lg[l] - mx);
        lg[l] = e;
        sm += e;
    }
    sm = wave_sum(sm);
    if (lane == 0) red[S2PA_NW + wave] = sm;
    __syncthreads();
    sm = 0.f;
#pragma unroll
    for (int w = 0; w < S2PA_NW; ++w) sm += red[S2PA_NW + w];   // fixed order: reproducible
    float* da = a.dict_attn + ((long long)b * L) * a.T_w + t;
    for (int l = tid; l < L; l += NTHR) {
        const float w = lg[l] / sm;
        lg[l] = w;
        da[(long long)l * a.T_w] = w;
    }
    __syncthreads();
    // weighted sum of the value rows (rows with zero weight contribute exactly zero: skipped)
    f32x4 acc[S2PA_DMAX4];
#pragma unroll
    for (int c = 0; c < S2PA_DMAX4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    // (masked rows have weight exactly 0 and rows >= Lrow are zero vectors: only the listed rows contribute)
    for (int i = wave * RU; i < n_val; i += S2PA_NW * RU) {
        f32x4 v[RU][S2PA_DMAX4];
        float w[RU];
#pragma unroll
        for (int j = 0; j < RU; ++j) {
            const int l = idx[min(i + j, n_val - 1)];
            w[j] = i + j < n_val ? lg[l] : 0.f;
            const f32x4* vr = vb + (long long)l * D4;
#pragma unroll
            for (int c = 0; c < S2PA_DMAX4; ++c)
                v[j][c] = (lane + 64 * c < D4) ? vr[lane + 64 * c] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < RU; ++j)
#pragma unroll
            for (int c = 0; c < S2PA_DMAX4; ++c) acc[c] += v[j][c] * w[j];
    }
#pragma unroll
    for (int c = 0; c < S2PA_DMAX4; ++c)
        if (lane + 64 * c < D4) *(f32x4*)&part[wave][(lane + 64 * c) * 4] = acc[c];
    // sense weights s_i = sum_l w[l] [key_map == i], deterministic order
    if (tid < 16) {
        float s = 0.f;
        if (tid >= 1)
            for (int l = 0; l < L; ++l) s += (km[l] == (float)tid) ? lg[l] : 0.f;
        sense[tid] = s;
    }
    __syncthreads();
    for (int c = tid; c < a.D; c += NTHR) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < S2PA_NW; ++w) sum += part[w][c];
        a.wv[(long long)row * a.D + c] = sum;
    }
    // pronunciation weights
    if (tid < a.P && tid < 64) {
        const long long pm = tid < Prow ? pmr[tid] : (special == -1 ? 1 : 0);
        float w = (pm >= 1 && pm < 16) ? sense[pm] : 0.f;
        if (a.language_zh && a.pron_modified) {
            const long long mod = a.pron_modified[row];
            if (mod >= 1 && mod <= (long long)(*a.pm_max)) {
                const float forced = (pm == mod) ? 1.f : 0.f;
                w = (forced - w) + w;  // weights_ - weights.detach() + weights (layers/utils.py:114)
            }
        } else if (a.language_zh) {
            w = (w - w) + w;
        }
        pw[tid] = w;
        a.pron_attn[(long long)row * a.P + tid] = w;
        long long id = tid < Prow ? pin[tid] : 0;
        if (id < 0 || id >= a.n_pinyin) id = 0;
        pid[tid] = (int)id;
    }
    __syncthreads();
    for (int c = tid; c < a.H; c += NTHR) {
        float s = 0.f;
        for (int p0 = 0; p0 < a.P; p0 += 4) {   // 4 embedding rows in flight, summed in p order
            float e[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = p0 + j < a.P ? a.pinyin_emb[(long long)pid[p0 + j] * a.H + c] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (p0 + j < a.P) s += pw[p0 + j] * e[j];
        }
        a.pron[(long long)row * a.H + c] = s;
    }
}
hipError_t s2pa_launch(const S2paArgs& a, hipStream_t s) {
    if (a.L_k > S2PA_LMAX || a.D > 768 || (a.D & 3) || a.P > 64) return hipErrorInvalidValue;
    static const int cfg = getenv("DTTS_S2PA_CFG") ? atoi(getenv("DTTS_S2PA_CFG")) : 0;   // tuning switch
    const dim3 grid(a.B * a.T_w);
    switch (cfg) {
    case 2: hipLaunchKernelGGL((s2pa_kernel<4, 8>), grid, dim3(256), 0, s, a); break;
    case 3: hipLaunchKernelGGL((s2pa_kernel<8, 4>), grid, dim3(512), 0, s, a); break;
    case 4: hipLaunchKernelGGL((s2pa_kernel<16, 2>), grid, dim3(1024), 0, s, a); break;
    case 5: hipLaunchKernelGGL((s2pa_kernel<16, 4>), grid, dim3(1024), 0, s, a); break;
    case 6: hipLaunchKernelGGL((s2pa_kernel<8, 8>), grid, dim3(512), 0, s, a); break;
    default: hipLaunchKernelGGL((s2pa_kernel<4, 4>), grid, dim3(256), 0, s, a); break;
    }
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

__global__ void rowcount_nonzero_kernel(const float* x, int* ilens, int T, int C) {
    __shared__ int cnt;
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    for (int t = wave; t < T; t += 4) {
        const float* r = x + ((long long)b * T + t) * C;
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += fabsf(r[c]);
        s = wave_sum(s);
        if (lane == 0 && s != 0.f) atomicAdd(&cnt, 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) ilens[b] = cnt;
}
hipError_t rowcount_nonzero_launch(const float* x, int* ilens, int B, int T, int C, hipStream_t s) {
    hipLaunchKernelGGL(rowcount_nonzero_kernel, dim3(B), dim3(256), 0, s, x, ilens, T, C);
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

__global__ void mel2word_fill_kernel(const int* starts, const int* total, int64_t* m2w, int T_w, int T_raw, int T_mel) {
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
}
hipError_t mel2word_fill_launch(const int* starts, const int* total, const int* ilens, int64_t* m2w, int B, int T_w,
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

__global__ void expand_kernel(const float* weo, const int64_t* m2w, float* x, float* x_mask, int T_w, int T_mel, int C,
                              int rows) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int b = row / T_mel;
    const long long w = m2w[row];
    f32x4* dst = (f32x4*)(x + (long long)row * C);
    if (w > 0 && w <= T_w) {
        const f32x4* src = (const f32x4*)(weo + ((long long)b * T_w + (w - 1)) * C);
        for (int c = lane; c < C / 4; c += 64) dst[c] = src[c];
    } else {
        for (int c = lane; c < C / 4; c += 64) dst[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (lane == 0) x_mask[row] = w > 0 ? 1.f : 0.f;
}
hipError_t expand_launch(const float* weo, const int64_t* m2w, float* x, float* x_mask, int B, int T_w, int T_mel, int C,
                         hipStream_t s) {
    hipLaunchKernelGGL(expand_kernel, dim3((B * T_mel + 3) / 4), dim3(256), 0, s, weo, m2w, x, x_mask, T_w, T_mel, C,
                       B * T_mel);
    return hipGetLastError();
}

__global__ void transpose_cf_to_cl_kernel(const float* x, float* y, int C, int T, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long long bt = i / C;
        const long long b = bt / T, t = bt % T;
        y[i] = x[(b * C + c) * T + t];
    }
}
hipError_t transpose_cf_to_cl_launch(const float* x, float* y, int B, int C, int T, hipStream_t s) {
    const long long n = (long long)B * C * T;
    const int blocks = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    hipLaunchKernelGGL(transpose_cf_to_cl_kernel, dim3(blocks), dim3(256), 0, s, x, y, C, T, n);
    return hipGetLastError();
}

} // namespace dtts
