// libdicttts_hip.so — context, weight folding/packing, and the orchestration of the two entry points
// (acoustic model, vocoder).  C ABI declared in include/dicttts_hip.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <ctime>
#include <functional>
#include <unistd.h>
#include <map>
#include <string>
#include <vector>

#include "../../include/dicttts_hip.h"
#include "tune_env.h"
#include "conv1d.h"
#include "ops.h"
#include "vconv.h"
#include "rblock.h"
#include "vpair.h"
#include "flowstack.h"

using namespace dtts;

namespace {

struct HostTensor {
    std::vector<float> f;
    std::vector<int64_t> shape;
    int64_t numel() const {
        int64_t n = 1;
        for (auto s : shape) n *= s;
        return n;
    }
};

// Memory-safety mode (dtts_config.debug_redzone, tests only): every workspace buffer sits between two RED ZONES of RZ bytes, the whole
// arena is filled with 0xFF (= NaN as fp32 / fp16 / bf16, -1 as an integer) before each forward, so that
//   * an out-of-range WRITE of a kernel damages a red zone (dtts_debug_check counts the bytes that are no longer 0xFF),
//   * an out-of-range or stale READ that is actually consumed shows up as NaN in the outputs (buffers are never zero by luck).
// Weight packs / tables (dev_alloc below) get the same red zones.  Off (the default): no red zones, no fills, no cost.
constexpr size_t RZ = 4096;

struct Arena {
    char* base = nullptr;
    size_t cap = 0, off = 0;
    bool debug = false;
    struct Buf { size_t start, bytes; };
    std::vector<Buf> bufs;   // debug: the buffers handed out since the last reserve / rewind
    static constexpr int DBG_BUFS = 1024;   // debug mode budgets red zones for this many buffers per forward (the largest forward hands out < 100)
    hipError_t reserve(size_t n, hipStream_t s) {
        off = 0;
        bufs.clear();
        if (debug) n += (size_t)DBG_BUFS * (RZ + 256);   // red zones + alignment of up to DBG_BUFS buffers (alloc fails beyond: see below)
        if (n > cap) {
            if (base) {
                hipError_t e = hipDeviceSynchronize();
                if (e != hipSuccess) return e;
                (void)hipFree(base);
                base = nullptr;
                cap = 0;
            }
            n = n + n / 8 + (1 << 20);
            hipError_t e = hipMalloc((void**)&base, n);
            if (e != hipSuccess) return e;
            cap = n;
        }
        if (debug) return hipMemsetAsync(base, 0xFF, cap, s);
        return hipSuccess;
    }
    void rewind() {   // walk the same layout again (decode re-derives the buffers encode laid out)
        off = 0;
        bufs.clear();
    }
    template <class T>
    T* alloc(size_t count) {
        if (debug) off += RZ;
        size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
        if (off + bytes + (debug ? RZ : 0) > cap) return nullptr;
        T* p = (T*)(base + off);
        if (debug) bufs.push_back({off, count * sizeof(T)});
        off += bytes;
        return p;
    }
    void release() {
        if (base) (void)hipFree(base);
        base = nullptr;
        cap = off = 0;
        bufs.clear();
    }
};

struct EncLayer {
    PackedConv qkv, o, ffn1, ffn2;
    float *g1 = nullptr, *b1 = nullptr, *g2 = nullptr, *b2 = nullptr;
};
struct Encoder {
    std::vector<EncLayer> l;
    float *lg = nullptr, *lb = nullptr;
};
struct WNet {
    PackedConv cond;
    std::vector<PackedConv> in, rs;
    int hidden = 0, layers = 0;
};
struct Flow {
    PackedConv pre, post;
    WNet wn;
    int in_coff = 0, out_coff = 0;  // physical channel offsets of the logical x0 / x1 halves (flip parity)
};

struct TimerSlot {
    bool enabled = false;
    std::vector<hipEvent_t> pool;
    size_t used = 0;
    double ms_done = 0;
    int64_t launches = 0;
};

} // namespace

struct dtts_ctx {
    dtts_config cfg;
    std::string err;
    std::map<std::string, HostTensor> w;
    std::vector<void*> allocs;
    bool debug_rz = false;                                   // dtts_config.debug_redzone
    int device = 0;                                          // the HIP device that was current at dtts_create: weights and workspaces live there
    int n_cu = 256;                                          // its compute units
    struct StaticBuf { char* p; size_t bytes; };
    std::vector<StaticBuf> rz_static;                         // debug: weight packs / tables (user pointer, payload bytes) between red zones
    bool acoustic_ready = false, vocoder_ready = false, fft_ready = false;
    // ---- FFT block stack
    struct FftLayer {
        PackedConv qkv, o, ffn1, ffn2;
        float *g1 = nullptr, *b1 = nullptr, *g2 = nullptr, *b2 = nullptr;
    };
    std::vector<FftLayer> fft;
    float *fft_g = nullptr, *fft_b = nullptr, *fft_alpha = nullptr;
    Arena a_fft;
    // ---- acoustic model
    float *word_emb = nullptr, *pinyin_emb = nullptr;
    Encoder sem, lin;
    PackedConv s2_q, s2_kT, s2_k, s2_v, s2_o;   // s2_kT: k_transform applied transposed to the query (tensor API); s2_k: to the table rows at upload
    std::vector<PackedConv> dur_conv;
    std::vector<float*> dur_g, dur_b;
    float *dur_w = nullptr, *dur_bias = nullptr;
    PackedConv g_pre, g_pre_poly, dec_pre, dec_out;   // g_pre_poly: the strided g_pre_net as a 3-tap convolution over 4-frame groups (vconv), or empty
    std::vector<Flow> flows;  // in execution (reversed) order
    float* fs_w = nullptr;    // packed weights of the fused prior-flow kernel (flowstack.hip), or null = launch by launch
    PackedConv fs_cond;       // cond_layer of ALL blocks as one 1x1 convolution (execution order)
    WNet dec_wn;
    // ---- vocoder
    PackedConv conv_pre, conv_post;
    std::vector<PackedConv> ups;
    std::vector<std::vector<PackedConv>> rb1, rb2;  // [resblock][3]
    std::vector<std::vector<PackedConv>> rbf1, rbf2;  // fused-ResBlock copies (taps zero padded), empty where unsupported
    int hop = 1;
    int tune = 0;
    float *post_w = nullptr, *post_b = nullptr;   // conv_post as [taps][C] fp32 for the fused epilogue of the last ResBlock (rblock.hip), or null
    // ---- workspaces and per-call state
    Arena a_enc, a_dec, a_voc;
    unsigned* amax_bits = nullptr;  // dtts_wav_to_int16 scratch
    unsigned long long noise_counter = 0x5EEDull;   // device prior samples (z_p == NULL): one stream per call, offset by noise_seed
    unsigned long long noise_seed = 0;              // per context (dtts_create: time, pid, device, instance; dtts_set_noise_seed overrides)
    unsigned long long* ovf_dev = nullptr;          // fp16 range guard counter (DTTS_VOC_F16), device
    bool guard_on = false;
    // always-on overflow detector of the 16-bit vocoder modes: non-finite pre-tanh values counted by the conv_post epilogue (device), and the
    // pinned host word every dtts_hifigan_forward copies it to behind its last kernel (dtts_vocoder_nonfinite reads it without a sync)
    unsigned* bad_dev = nullptr;
    volatile unsigned* bad_host = nullptr;
    // static fp16 analysis of the ResBlock operands (build_vocoder): bound(M) <= wc_lin * M + wc_const for |mel| <= M (worst case, L1),
    // est_lin * M + est_const = the propagated RMS (an ESTIMATE under independence); 0 / 0 when the mode has no fp16 operands
    double wc_lin = 0, wc_const = 0, est_lin = 0, est_const = 0;
    bool voc_span = false;                          // DTTS_TIMER_VOC_CONV: one event pair spans the whole kernel family of a forward (below)
    int amax_cap = 0;
    int B = 0, T_w = 0, L_k = 0, P = 0, T_mel = 0;
    bool encoded = false;
    float *weo = nullptr, *dur = nullptr, *pron_attn = nullptr, *dict_attn = nullptr, *context = nullptr, *x_mask = nullptr;
    int64_t* m2w = nullptr;
    int *mel_lens = nullptr, *lens = nullptr;
    TimerSlot timers[DTTS_TIMER_COUNT];
    // ---- resident dictionary table (dtts_dict_table_upload)
    int t_entries = 0;
    int *t_off = nullptr, *t_poff = nullptr, *t_pmmax = nullptr;
    float *t_keys = nullptr, *t_values = nullptr, *t_key_map = nullptr;
    bool t_projected = false;   // t_keys / t_values hold K = key Wk^T and V = value Wv^T (hidden_size wide) instead of the raw gloss rows
    int64_t *t_pinyin = nullptr, *t_pinyin_map = nullptr;
};

static std::string g_create_err;

namespace {

int fail(dtts_ctx* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    else g_create_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                                       \
    do {                                                                                                   \
        hipError_t _e = (expr);                                                                            \
        if (_e != hipSuccess) return fail(h, DTTS_E_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));   \
    } while (0)

// device allocation owned by the context (freed by dtts_destroy / dev_free); debug_redzone: between two 0xFF red zones
void* dev_alloc(dtts_ctx* h, size_t bytes) {
    bytes = std::max<size_t>(bytes, 16);
    char* d = nullptr;
    if (!h->debug_rz) {
        if (hipMalloc((void**)&d, bytes) != hipSuccess) return nullptr;
        h->allocs.push_back(d);
        return d;
    }
    const size_t padded = (bytes + 255) & ~(size_t)255;
    if (hipMalloc((void**)&d, padded + 2 * RZ) != hipSuccess) return nullptr;
    h->allocs.push_back(d);
    if (hipMemset(d, 0xFF, padded + 2 * RZ) != hipSuccess) return nullptr;
    h->rz_static.push_back({d + RZ, bytes});
    return d + RZ;
}
void dev_free(dtts_ctx* h, void* user) {
    if (!user) return;
    char* basep = (char*)user - (h->debug_rz ? RZ : 0);
    auto it = std::find(h->allocs.begin(), h->allocs.end(), (void*)basep);
    if (it == h->allocs.end()) return;
    (void)hipFree(basep);
    h->allocs.erase(it);
    for (size_t i = 0; i < h->rz_static.size(); ++i)
        if (h->rz_static[i].p == (char*)user) {
            h->rz_static.erase(h->rz_static.begin() + i);
            break;
        }
}

template <class T>
T* upload(dtts_ctx* h, const std::vector<T>& v) {
    T* d = (T*)dev_alloc(h, v.size() * sizeof(T));
    if (!d) return nullptr;
    if (!v.empty() && hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}

uint16_t f2bf_host(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
float bf2f_host(uint16_t hbits) {
    uint32_t u = (uint32_t)hbits << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// fp32 -> IEEE half bits, round-to-nearest-even, saturating at the largest finite half (weights never get there)
uint16_t f2h_host(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    const uint32_t a = u & 0x7fffffffu;
    if (a >= 0x7f800000u) return (uint16_t)(sign | (a > 0x7f800000u ? 0x7e00u : 0x7c00u));
    if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7bffu);            // >= 65520 rounds past the largest finite half: saturate
    if (a < 0x33000001u) return (uint16_t)sign;                          // <= 2^-25: rounds to zero
    const int e = (int)(a >> 23) - 127;
    uint32_t m = (a & 0x7fffffu) | 0x800000u;                            // 24-bit significand
    int shift = e >= -14 ? 13 : 13 + (-14 - e);                          // bits dropped (subnormal halves drop more)
    const uint32_t half_ulp = 1u << (shift - 1), rem = m & ((1u << shift) - 1);
    uint32_t q = m >> shift;
    if (rem > half_ulp || (rem == half_ulp && (q & 1u))) ++q;
    uint32_t out = e >= -14 ? (((uint32_t)(e + 15) << 10) + (q - 0x400u)) : q;   // a carry out of the significand bumps the exponent
    return (uint16_t)(sign | out);
}

// Pack one convolution into MFMA fragment order and upload it.  getw(co, ci, tap) addresses the LOGICAL
// weight; bias is in logical channel order.  gate_H > 0: logical C_out = 2*gate_H, packed co-tiles
// alternate (tanh[32j..32j+32), sigmoid[H+32j..H+32j+32)).
bool pack_conv(dtts_ctx* h, PackedConv& L, int engine, int C_out, int C_in, int K,
               const std::function<float(int, int, int)>& getw, const std::vector<float>& bias, int dil, int stride,
               int pad, int gate_H = 0, double flops_per_row = -1) {
    L.engine = engine;
    L.C_in = C_in;
    L.C_out = C_out;
    L.K = K;
    L.dil = dil;
    L.stride = stride;
    L.pad = pad;
    L.gate_H = gate_H;
    L.CK = (C_in <= 32) ? 32 : 64;
    L.C_in_pad = (C_in + L.CK - 1) / L.CK * L.CK;
    L.C_out_pad = (C_out + 31) / 32 * 32;
    L.flops_per_row = flops_per_row >= 0 ? flops_per_row : 2.0 * C_out * C_in * K;
    const int KG = engine == ENG_F32 ? 8 : 16, E = KG / 2;
    const int NG = L.C_in_pad / KG, NCT = L.C_out_pad / 32;
    // + zero k-steps of slack behind the last tap: the kernels prefetch weight fragments past the end instead of clamping.  vconv walks
    // one C_in CHUNK at a time and, at the end of a chunk's last tap, its running pointer wraps to "next tap, same chunk" = up to a whole
    // tap's k-groups (NG) beyond the end for the last chunk (found in round 3: the 192 -> 2048 conditioning convolution read one
    // 64 KB step past the old 8-step slack — a GPU page fault whenever the allocation ended on a mapped-region boundary)
    const size_t n = (size_t)K * L.C_in_pad * L.C_out_pad + (size_t)(L.C_in_pad / 16 + 8) * 16 * L.C_out_pad;
    std::vector<float> wf;
    std::vector<uint16_t> whi, wlo;
    if (engine == ENG_F32) wf.assign(n, 0.f);
    else {
        whi.assign(n, 0);
        if (engine == ENG_BF16X3) wlo.assign(n, 0);
    }
    for (int pco = 0; pco < L.C_out_pad; ++pco) {
        int co = pco;
        if (gate_H) {
            const int tile = pco / 32, j = tile / 2, within = pco % 32;
            co = (tile & 1) ? gate_H + j * 32 + within : j * 32 + within;
            if (j * 32 + within >= gate_H) co = -1;
        }
        if (co < 0 || co >= C_out) continue;
        const int ct = pco / 32, col = pco % 32;
        for (int tap = 0; tap < K; ++tap)
            for (int ci = 0; ci < C_in; ++ci) {
                const int g = ci / KG, within = ci % KG, half = within / E, e = within % E;
                const size_t idx = ((((size_t)tap * NG + g) * NCT + ct) * 64 + half * 32 + col) * E + e;
                const float v = getw(co, ci, tap);
                if (engine == ENG_F32) wf[idx] = v;
                else if (engine == ENG_F16) whi[idx] = f2h_host(v);
                else {
                    const uint16_t hi = f2bf_host(v);
                    whi[idx] = hi;
                    if (engine == ENG_BF16X3) wlo[idx] = f2bf_host(v - bf2f_host(hi));
                }
            }
    }
    if (engine == ENG_F32) {
        L.w_hi = upload(h, wf);
        // + the same weights as three bf16 pieces in k-groups of 16 (conv1d.h: ENG_BF16X6) for the short-sequence kernel
        if (!DTTS_TUNE(h, 131072) && L.C_in_pad % 16 == 0) {   // DTTS_TUNE bit 17: fp32 MFMA only
            const int KG6 = 16, E6 = 8, NG6 = L.C_in_pad / KG6;
            std::vector<uint16_t> pc[3];
            for (auto& v : pc) v.assign(n, 0);
            for (int pco = 0; pco < L.C_out_pad; ++pco) {
                int co = pco;
                if (gate_H) {
                    const int tile = pco / 32, j = tile / 2, within = pco % 32;
                    co = (tile & 1) ? gate_H + j * 32 + within : j * 32 + within;
                    if (j * 32 + within >= gate_H) co = -1;
                }
                if (co < 0 || co >= C_out) continue;
                const int ct = pco / 32, col = pco % 32;
                for (int tap = 0; tap < K; ++tap)
                    for (int ci = 0; ci < C_in; ++ci) {
                        const int g = ci / KG6, within = ci % KG6, half = within / E6, e = within % E6;
                        const size_t idx = ((((size_t)tap * NG6 + g) * NCT + ct) * 64 + half * 32 + col) * E6 + e;
                        float r = getw(co, ci, tap);
                        for (int pl = 0; pl < 3; ++pl) {
                            const uint16_t b16 = f2bf_host(r);
                            pc[pl][idx] = b16;
                            r -= bf2f_host(b16);   // exact: the remainder of a round-to-nearest bf16 fits fp32
                        }
                    }
            }
            for (int pl = 0; pl < 3; ++pl) L.x6[pl] = upload(h, pc[pl]);
            if (!L.x6[0] || !L.x6[1] || !L.x6[2]) return false;
        }
    } else {
        L.w_hi = upload(h, whi);
        if (engine == ENG_BF16X3) L.w_lo = upload(h, wlo);
    }
    if (!bias.empty()) {
        std::vector<float> bp(bias);
        bp.resize(std::max<size_t>(bias.size(), (size_t)L.C_out_pad), 0.f);  // zero padded: 16 B loads in vconv's epilogue
        L.bias = upload(h, bp);
    }
    return L.w_hi != nullptr && (bias.empty() || L.bias != nullptr);
}

// ---------------------------------------------------------------------------------------------------------
// weight access
struct Need {
    dtts_ctx* h;
    std::string missing;
    const HostTensor* get(const std::string& name) {
        auto it = h->w.find(name);
        if (it == h->w.end()) {
            if (missing.empty()) missing = name;
            return nullptr;
        }
        return &it->second;
    }
};

// fold weight norm if <base>.weight is absent: w = v * (g / ||v||), norm over all dims but 0
// (torch.nn.utils.weight_norm dim=0; remove_weight_norm at tasks/tts/ps_flow.py:262-268, hifigan.py:144-151)
const HostTensor* folded_weight(dtts_ctx* h, Need& need, const std::string& base) {
    auto it = h->w.find(base + ".weight");
    if (it != h->w.end()) return &it->second;
    const HostTensor* g = need.get(base + ".weight_g");
    const HostTensor* v = need.get(base + ".weight_v");
    if (!g || !v) return nullptr;
    HostTensor out;
    out.shape = v->shape;
    out.f.resize(v->f.size());
    const int64_t d0 = v->shape[0], inner = v->numel() / d0;
    for (int64_t i = 0; i < d0; ++i) {
        double ss = 0;
        for (int64_t j = 0; j < inner; ++j) ss += (double)v->f[i * inner + j] * v->f[i * inner + j];
        const float nrm = (float)std::sqrt(ss);
        const float sc = g->f[i] / nrm;
        for (int64_t j = 0; j < inner; ++j) out.f[i * inner + j] = v->f[i * inner + j] * sc;
    }
    auto& slot = h->w[base + ".weight"];
    slot = std::move(out);
    return &slot;
}

std::vector<float> bias_of(Need& need, const std::string& base) {
    const HostTensor* b = need.get(base + ".bias");
    return b ? b->f : std::vector<float>();
}

// ordinary Conv1d weight [C_out][C_in][K]
bool pack_plain(dtts_ctx* h, Need& need, PackedConv& L, int engine, const std::string& base, int dil, int stride, int pad,
                bool with_bias = true, int gate_H = 0) {
    const HostTensor* w = folded_weight(h, need, base);
    if (!w) return false;
    const int C_out = (int)w->shape[0], C_in = (int)w->shape[1], K = w->shape.size() > 2 ? (int)w->shape[2] : 1;
    std::vector<float> bias = with_bias ? bias_of(need, base) : std::vector<float>();
    if (with_bias && bias.empty()) return false;
    const float* p = w->f.data();
    return pack_conv(h, L, engine, C_out, C_in, K,
                     [=](int co, int ci, int tap) { return p[((size_t)co * C_in + ci) * K + tap]; }, bias, dil, stride,
                     pad, gate_H);
}

// ConvTranspose1d weight [C_in][C_out][k], stride u, padding p -> polyphase Conv1d with u*C_out channels
// (phase-major), taps over input offsets {-1,0,+1} (or a single tap when k == u, p == 0):
// out[u*q + r] = sum_delta sum_ci x[q + delta][ci] * w[ci][co][r + p - u*delta]
bool pack_transposed(dtts_ctx* h, Need& need, PackedConv& L, int engine, const std::string& base, int u, int p) {
    const HostTensor* w = folded_weight(h, need, base);
    if (!w) return false;
    const int C_in = (int)w->shape[0], C_out = (int)w->shape[1], k = (int)w->shape[2];
    std::vector<float> b0 = bias_of(need, base);
    if (b0.empty()) return false;
    if (k > 2 * u || p >= u) {
        fail(h, DTTS_E_INVAL, "%s: unsupported transposed conv k=%d stride=%d pad=%d", base.c_str(), k, u, p);
        return false;
    }
    const bool single = (k == u && p == 0);
    const int K = single ? 1 : 3, pad = single ? 0 : 1;
    std::vector<float> bias((size_t)u * C_out);
    for (int r = 0; r < u; ++r)
        for (int co = 0; co < C_out; ++co) bias[(size_t)r * C_out + co] = b0[co];
    const float* pw = w->f.data();
    // k = 2u, pad = u/2 (HifiGAN's upsamplers): phase r < u/2 reads input offsets {-1, 0}, r >= u/2 reads {0, +1} — a third
    // of the 3-tap polyphase weights are structural zeros and the kernel skips them per wave (vconv.hip: poly_half)
    const int cop = u * C_out, wave_ch = (cop % 256 == 0) ? 64 : 32;   // channels per wave of the vconv configuration this layer gets
    const bool half = !single && k == 2 * u && 2 * p == u && (cop / 2) % wave_ch == 0;
    const bool ok = pack_conv(
        h, L, engine, u * C_out, C_in, K,
        [=](int pco, int ci, int tap) {
            const int r = pco / C_out, co = pco % C_out, delta = tap - pad;
            const int j = r + p - u * delta;
            return (j >= 0 && j < k) ? pw[((size_t)ci * C_out + co) * k + j] : 0.f;
        },
        bias, 1, 1, pad, 0, 2.0 * C_in * C_out * k /* per INPUT row: u outputs x k/u taps */);
    L.poly_half = (half && !DTTS_TUNE(h, 2)) ? 1 : 0;
    return ok;
}

float* upload_named(dtts_ctx* h, Need& need, const std::string& name) {
    const HostTensor* t = need.get(name);
    return t ? upload(h, t->f) : nullptr;
}

bool build_encoder(dtts_ctx* h, Need& need, Encoder& E, const std::string& p) {
    const int n = h->cfg.enc_layers, C = h->cfg.hidden_size, K = h->cfg.enc_ffn_kernel_size;
    E.l.resize(n);
    for (int i = 0; i < n; ++i) {
        EncLayer& l = E.l[i];
        const std::string a = p + ".attn_layers." + std::to_string(i);
        const HostTensor *wq = need.get(a + ".conv_q.weight"), *wk = need.get(a + ".conv_k.weight"),
                         *wv = need.get(a + ".conv_v.weight");
        const HostTensor *bq = need.get(a + ".conv_q.bias"), *bk = need.get(a + ".conv_k.bias"),
                         *bv = need.get(a + ".conv_v.bias");
        if (!wq || !wk || !wv || !bq || !bk || !bv) return false;
        std::vector<float> bias(3 * C);
        for (int c = 0; c < C; ++c) {
            bias[c] = bq->f[c];
            bias[C + c] = bk->f[c];
            bias[2 * C + c] = bv->f[c];
        }
        const float *pq = wq->f.data(), *pk = wk->f.data(), *pv = wv->f.data();
        if (!pack_conv(h, l.qkv, ENG_F32, 3 * C, C, 1,
                       [=](int co, int ci, int) {
                           const float* src = co < C ? pq : (co < 2 * C ? pk : pv);
                           return src[(size_t)(co % C) * C + ci];
                       },
                       bias, 1, 1, 0))
            return false;
        if (!pack_plain(h, need, l.o, ENG_F32, a + ".conv_o", 1, 1, 0)) return false;
        const std::string f = p + ".ffn_layers." + std::to_string(i);
        if (!pack_plain(h, need, l.ffn1, ENG_F32, f + ".conv_1", 1, 1, K / 2)) return false;
        if (!pack_plain(h, need, l.ffn2, ENG_F32, f + ".conv_2", 1, 1, 0)) return false;
        l.g1 = upload_named(h, need, p + ".norm_layers_1." + std::to_string(i) + ".gamma");
        l.b1 = upload_named(h, need, p + ".norm_layers_1." + std::to_string(i) + ".beta");
        l.g2 = upload_named(h, need, p + ".norm_layers_2." + std::to_string(i) + ".gamma");
        l.b2 = upload_named(h, need, p + ".norm_layers_2." + std::to_string(i) + ".beta");
        if (!l.g1 || !l.b1 || !l.g2 || !l.b2) return false;
    }
    E.lg = upload_named(h, need, p + ".last_ln.gamma");
    E.lb = upload_named(h, need, p + ".last_ln.beta");
    return E.lg && E.lb;
}

// eng: ENG_F32 (exact fp32 MFMA, generic kernel) or ENG_BF16X3 (split operands: the vconv kernel's WaveNet form)
bool build_wn(dtts_ctx* h, Need& need, WNet& W, const std::string& p, int hidden, int k, int layers, int eng = ENG_F32) {
    W.hidden = hidden;
    W.layers = layers;
    W.in.resize(layers);
    W.rs.resize(layers);
    for (int i = 0; i < layers; ++i) {
        if (!pack_plain(h, need, W.in[i], eng, p + ".in_layers." + std::to_string(i), 1, 1, (k - 1) / 2, true, hidden))
            return false;
        if (!pack_plain(h, need, W.rs[i], eng, p + ".res_skip_layers." + std::to_string(i), 1, 1, 0)) return false;
    }
    return pack_plain(h, need, W.cond, ENG_F32, p + ".cond_layer", 1, 1, 0);
}

int build_acoustic(dtts_ctx* h) {
    Need need{h, ""};
    const dtts_config& c = h->cfg;
    const std::string m = "model.";
    const std::string enc = m + "dict_encoder.S2PA_module";
    bool ok = true;
    h->word_emb = upload_named(h, need, enc + ".word_emb.weight");
    const std::string att = enc + ".s2pa_attention";
    h->pinyin_emb = upload_named(h, need, att + ".pinyin_embedding.weight");
    ok = ok && h->word_emb && h->pinyin_emb;
    ok = ok && build_encoder(h, need, h->sem, enc + ".semantic_encoder");
    ok = ok && build_encoder(h, need, h->lin, enc + ".linguistic_encoder");
    // S2PA projections (no bias).  k_transform is applied TRANSPOSED to the query (see ops.h)
    const HostTensor *wq = need.get(att + ".q_transform.weight"), *wk = need.get(att + ".k_transform.weight"),
                     *wv = need.get(att + ".v_transform.weight"), *wo = need.get(att + ".output_transform.weight");
    if (ok && wq && wk && wv && wo) {
        const int H = c.hidden_size, D = c.gloss_dim;
        const float *pq = wq->f.data(), *pk = wk->f.data(), *pv = wv->f.data(), *po = wo->f.data();
        ok = ok && pack_conv(h, h->s2_q, ENG_F32, H, H, 1, [=](int co, int ci, int) { return pq[(size_t)co * H + ci]; }, {}, 1, 1, 0);
        ok = ok && pack_conv(h, h->s2_kT, ENG_F32, D, H, 1, [=](int co, int ci, int) { return pk[(size_t)ci * D + co]; }, {}, 1, 1, 0);
        ok = ok && pack_conv(h, h->s2_k, ENG_F32, H, D, 1, [=](int co, int ci, int) { return pk[(size_t)co * D + ci]; }, {}, 1, 1, 0);
        ok = ok && pack_conv(h, h->s2_v, ENG_F32, H, D, 1, [=](int co, int ci, int) { return pv[(size_t)co * D + ci]; }, {}, 1, 1, 0);
        ok = ok && pack_conv(h, h->s2_o, ENG_F32, H, H, 1, [=](int co, int ci, int) { return po[(size_t)co * H + ci]; }, {}, 1, 1, 0);
    } else
        ok = false;
    // duration predictor
    h->dur_conv.resize(c.dur_predictor_layers);
    h->dur_g.resize(c.dur_predictor_layers);
    h->dur_b.resize(c.dur_predictor_layers);
    for (int i = 0; ok && i < c.dur_predictor_layers; ++i) {
        const std::string p = m + "dur_predictor.conv." + std::to_string(i);
        ok = ok && pack_plain(h, need, h->dur_conv[i], ENG_F32, p + ".1", 1, 1, (c.dur_predictor_kernel - 1) / 2);
        h->dur_g[i] = upload_named(h, need, p + ".3.weight");
        h->dur_b[i] = upload_named(h, need, p + ".3.bias");
        ok = ok && h->dur_g[i] && h->dur_b[i];
    }
    h->dur_w = upload_named(h, need, m + "dur_predictor.linear.0.weight");
    h->dur_bias = upload_named(h, need, m + "dur_predictor.linear.0.bias");
    ok = ok && h->dur_w && h->dur_bias;
    // FVAE
    ok = ok && pack_plain(h, need, h->g_pre, (c.decoder_fp32 || DTTS_TUNE(h, 1024)) ? ENG_F32 : ENG_BF16X3, m + "fvae.g_pre_net.0", 1, 4, 2);   // DTTS_TUNE bit 10: fp32 (round 2)
    // g_pre_net = Conv1d(k = 8, stride 4, pad 2) as a STRIDE-1, 3-tap convolution over 4-frame groups: [B][T][C] is also [B][T/4][4C]
    // (T is a multiple of frames_multiple = 4), out[q] = sum_j W_j x[4q + j - 2] reads group q - 1 (frames 2, 3), q (all four) and q + 1
    // (frames 0, 1) — on the split-operand vconv kernel, which skips the two all-zero half taps per input chunk (vconv.hip: in_half).
    // DTTS_TUNE bit 16: the strided form on the generic kernel.
    h->g_pre_poly = PackedConv();
    if (ok && !c.decoder_fp32 && !DTTS_TUNE(h, 1024) && !DTTS_TUNE(h, 65536) && c.frames_multiple == 4 && c.hidden_size % 64 == 0) {
        const HostTensor* wg = folded_weight(h, need, m + "fvae.g_pre_net.0");
        std::vector<float> bg = bias_of(need, m + "fvae.g_pre_net.0");
        if (wg && wg->shape.size() == 3 && wg->shape[2] == 8 && !bg.empty()) {
            const int Co = (int)wg->shape[0], Ci = (int)wg->shape[1];
            const float* pw = wg->f.data();
            ok = pack_conv(h, h->g_pre_poly, ENG_BF16X3, Co, 4 * Ci, 3,
                           [=](int co, int cip, int tap) {
                               const int ph = cip / Ci, ci = cip % Ci, j = 4 * (tap - 1) + ph + 2;
                               return (j >= 0 && j < 8) ? pw[((size_t)co * Ci + ci) * 8 + j] : 0.f;
                           },
                           bg, 1, 1, 1, 0, 2.0 * Co * Ci * 8);
        }
    }
    const int half = c.latent_size / 2;
    h->flows.clear();
    int parity = 0;
    // one fused kernel for the whole prior flow where the configuration allows (DTTS_TUNE bit 8: launch by launch again)
    bool fuse_flows = flowstack_supported(c.prior_glow_hidden, c.glow_kernel_size, c.prior_glow_n_layers, c.prior_glow_n_blocks, c.latent_size) &&
                      !DTTS_TUNE(h, 256);
    std::vector<float> fs_host, fs_cond_w, fs_cond_b;
    for (int f = c.prior_glow_n_blocks - 1; ok && f >= 0; --f) {
        // reversed(flows): Flip, then the coupling layer (glow_modules.py:157-163).  The flip is not executed:
        // it is tracked as a parity and folded into the channel order of pre / post.
        parity ^= 1;
        Flow fl;
        const std::string p = m + "fvae.prior_flow.flows." + std::to_string(2 * f);
        const HostTensor *wpre = need.get(p + ".pre.weight"), *wpost = need.get(p + ".post.weight");
        std::vector<float> bpre = bias_of(need, p + ".pre"), bpost = bias_of(need, p + ".post");
        if (!wpre || !wpost || bpre.empty() || bpost.empty()) { ok = false; break; }
        const int Hf = c.prior_glow_hidden;
        const float *ppre = wpre->f.data(), *ppost = wpost->f.data();
        const bool rev = parity == 1;
        // logical x0[c] = phys[rev ? 15 - c : c], c < half ; logical x1[c] = phys[rev ? 7 - c : 8 + c]
        fl.in_coff = rev ? half : 0;
        fl.out_coff = rev ? 0 : half;
        ok = ok && pack_conv(h, fl.pre, ENG_F32, Hf, half, 1,
                             [=](int co, int ci, int) { return ppre[(size_t)co * half + (rev ? half - 1 - ci : ci)]; }, bpre, 1, 1, 0);
        std::vector<float> nb(half);
        for (int q = 0; q < half; ++q) nb[q] = -bpost[rev ? half - 1 - q : q];
        // x1 = x1 - m  ->  epilogue residual add with negated weights
        ok = ok && pack_conv(h, fl.post, ENG_F32, half, Hf, 1,
                             [=](int co, int ci, int) { return -ppost[(size_t)(rev ? half - 1 - co : co) * Hf + ci]; }, nb, 1, 1, 0);
        ok = ok && build_wn(h, need, fl.wn, p + ".enc", Hf, c.glow_kernel_size, c.prior_glow_n_layers);
        h->flows.push_back(fl);
        if (ok && fuse_flows) {   // the same block for the fused kernel (flowstack.hip): logical weights with the flip folded in as above
            FlowStackHostWeights fw;
            fw.pre.resize((size_t)Hf * half);
            for (int co = 0; co < Hf; ++co)
                for (int ci = 0; ci < half; ++ci) fw.pre[(size_t)co * half + ci] = ppre[(size_t)co * half + (rev ? half - 1 - ci : ci)];
            fw.bpre = bpre;
            fw.post.resize((size_t)half * Hf);
            for (int q = 0; q < half; ++q)
                for (int ci = 0; ci < Hf; ++ci) fw.post[(size_t)q * Hf + ci] = -ppost[(size_t)(rev ? half - 1 - q : q) * Hf + ci];
            fw.bpost = nb;
            for (int l = 0; ok && l < c.prior_glow_n_layers; ++l) {
                const std::string bi = p + ".enc.in_layers." + std::to_string(l), br = p + ".enc.res_skip_layers." + std::to_string(l);
                const HostTensor *wi = folded_weight(h, need, bi), *wr = folded_weight(h, need, br);
                std::vector<float> b1 = bias_of(need, bi), b2 = bias_of(need, br);
                const size_t n_rs = (size_t)(l == c.prior_glow_n_layers - 1 ? Hf : 2 * Hf);
                if (!wi || !wr || wi->f.size() != (size_t)2 * Hf * Hf * c.glow_kernel_size || wr->f.size() != n_rs * Hf || b1.size() != (size_t)2 * Hf ||
                    b2.size() != n_rs) {
                    fuse_flows = false;   // an unexpected shape: the launch-by-launch path reports it
                    break;
                }
                fw.in.push_back(wi->f);
                fw.bin.push_back(b1);
                fw.rs.push_back(wr->f);
                fw.brs.push_back(b2);
            }
            const HostTensor* wc = fuse_flows ? folded_weight(h, need, p + ".enc.cond_layer") : nullptr;
            std::vector<float> bc = fuse_flows ? bias_of(need, p + ".enc.cond_layer") : std::vector<float>();
            const size_t n_c = (size_t)2 * Hf * c.prior_glow_n_layers;
            if (fuse_flows && (!wc || wc->f.size() != n_c * c.hidden_size || bc.size() != n_c)) fuse_flows = false;
            if (fuse_flows) {
                flowstack_pack(fw, c.prior_glow_n_layers, !c.decoder_fp32, fs_host);
                fs_cond_w.insert(fs_cond_w.end(), wc->f.begin(), wc->f.end());
                fs_cond_b.insert(fs_cond_b.end(), bc.begin(), bc.end());
            }
        }
    }
    if (ok && parity != 0) return fail(h, DTTS_E_INVAL, "odd number of flow blocks is not supported");
    h->fs_w = nullptr;
    if (ok && fuse_flows && !h->flows.empty()) {
        h->fs_w = upload(h, fs_host);
        const int n_c = (int)fs_cond_b.size(), Cg = c.hidden_size;
        const float* pc = fs_cond_w.data();
        // (split-bf16 operands on the vconv kernel like the WaveNet layers it conditions, unless the exact-fp32 decoder was asked for)
        ok = ok && h->fs_w && pack_conv(h, h->fs_cond, (c.decoder_fp32 || Cg % 64 || n_c % 256 || DTTS_TUNE(h, 2048)) ? ENG_F32 : ENG_BF16X3, n_c, Cg, 1,
                                         [=](int co, int ci, int) { return pc[(size_t)co * Cg + ci]; }, fs_cond_b, 1, 1, 0);
    }
    ok = ok && pack_transposed(h, need, h->dec_pre, ENG_F32, m + "fvae.decoder.pre_net.0", 4, 0);
    // the decoder WaveNet carries 4.09 of the acoustic model's 4.69 MFLOP per frame: split-bf16 operands (three bf16 MFMAs
    // per product = 5.3x the fp32-MFMA rate, mel error ~3e-5 against the 1e-3 gate) unless the hidden width does not tile
    const int dec_eng = (c.fvae_enc_dec_hidden % 64 == 0 && !c.decoder_fp32) ? ENG_BF16X3 : ENG_F32;
    ok = ok && build_wn(h, need, h->dec_wn, m + "fvae.decoder.wn", c.fvae_enc_dec_hidden, c.fvae_kernel_size, c.fvae_dec_n_layers, dec_eng);
    ok = ok && pack_plain(h, need, h->dec_out, ENG_F32, m + "fvae.decoder.out_proj", 1, 1, 0);
    if (!ok) {
        if (!need.missing.empty()) return fail(h, DTTS_E_NOENT, "missing weight tensor '%s'", need.missing.c_str());
        if (h->err.empty()) return fail(h, DTTS_E_NOMEM, "packing / uploading acoustic weights failed");
        return DTTS_E_INVAL;
    }
    h->acoustic_ready = true;
    return DTTS_OK;
}

// Static fp16 analysis of the ResBlock operands (VERDICT r4 #3a).  The fused kernels round leaky_relu(x) and leaky_relu(xt) to fp16 in all 72
// ResBlock convolutions, where the reference computes in fp32 (modules/hifigan/hifigan.py:51-58).  Propagated from |mel| <= M through the
// folded weights, per channel:
//   worst case (a PROOF when it stays below 65504):  u_out[co] = |b[co]| + sum_ci u_in[ci] * sum_k |w[co][ci][k]|   (transposed convolutions:
//     the largest output phase), leaky_relu does not grow a bound, the residual adds, the stage output is the mean of its ResBlocks;
//   RMS estimate (NOT a proof: independent, zero-mean terms):  m_out[co] = b^2 + sum_ci m_in[ci] * sum_k w^2, leaky_relu halves it.
// Every (operand, channel) bound is affine in M (M^2 for the second moments): a_q + b_q * M with a_q = the value at M = 0 and b_q = value(1) -
// value(0), both >= 0.  The PEAK over channels is a maximum of affine functions — convex, so a secant through the peaks at M = 0 and 1
// would UNDERestimate it beyond M = 1 (a bias-dominated channel sets both peaks while another channel's gain term overtakes it at
// M = 6: ADVICE r5).  Reported instead: max_q a_q + M * max_q b_q >= max_q (a_q + b_q M) for every M >= 0 — looser, but a bound.
// For real checkpoints the worst case is astronomically loose (it compounds sum|w| ~ 10-40 per convolution over 6 convolutions per
// ResBlock): it proves small-gain generators only.  Everything else runs fp16 under the always-on detector (conv_post epilogue).
bool vocoder_fp16_analysis(dtts_ctx* h, Need& need) {
    const dtts_config& c = h->cfg;
    const std::string v = "vocoder.";
    const int nk = c.n_resblock_kernels;
    std::vector<double> pt_wc[2], pt_m2[2];   // [pass][operand point x channel, in traversal order]: pass 0 = M = 0 (the bias part), pass 1 = M = 1
    for (int pass = 0; pass < 2; ++pass) {
        const double M = pass;
        auto conv = [&](const std::string& base, const std::vector<double>& uin, const std::vector<double>& min, std::vector<double>& uout,
                        std::vector<double>& mout) -> bool {   // Conv1d weight [co][ci][k]
            const HostTensor* w = folded_weight(h, need, base);
            const std::vector<float> b = bias_of(need, base);
            if (!w || b.empty() || w->shape.size() != 3) return false;
            const int co_n = (int)w->shape[0], ci_n = (int)w->shape[1], k_n = (int)w->shape[2];
            if ((int)uin.size() < ci_n) return false;
            uout.assign(co_n, 0.0);
            mout.assign(co_n, 0.0);
            for (int co = 0; co < co_n; ++co) {
                double su = std::fabs((double)b[co]), sm = (double)b[co] * b[co];
                for (int ci = 0; ci < ci_n; ++ci) {
                    double a1 = 0, a2 = 0;
                    const float* pw = &w->f[((size_t)co * ci_n + ci) * k_n];
                    for (int k = 0; k < k_n; ++k) {
                        a1 += std::fabs((double)pw[k]);
                        a2 += (double)pw[k] * pw[k];
                    }
                    su += a1 * uin[ci];
                    sm += a2 * min[ci];
                }
                uout[co] = su;
                mout[co] = sm;
            }
            return true;
        };
        std::vector<double> u(c.audio_num_mel_bins, M), m(c.audio_num_mel_bins, M * M), u2, m2;
        if (!conv(v + "conv_pre", u, m, u2, m2)) return false;
        u.swap(u2);
        m.swap(m2);
        for (int i = 0; i < c.n_upsamples; ++i) {
            const int r = c.upsample_rates[i], k_n = c.upsample_kernel_sizes[i], pad = (k_n - r) / 2;
            const HostTensor* w = folded_weight(h, need, v + "ups." + std::to_string(i));   // ConvTranspose1d [ci][co][k]
            const std::vector<float> b = bias_of(need, v + "ups." + std::to_string(i));
            if (!w || b.empty() || w->shape.size() != 3 || (int)w->shape[2] != k_n) return false;
            const int ci_n = (int)w->shape[0], co_n = (int)w->shape[1];
            std::vector<double> ux(co_n, 0.0), mx(co_n, 0.0);
            for (int co = 0; co < co_n; ++co)
                for (int ph = 0; ph < r; ++ph) {   // output phase ph collects the taps k == ph + pad (mod r)
                    double su = std::fabs((double)b[co]), sm = (double)b[co] * b[co];
                    for (int ci = 0; ci < ci_n; ++ci) {
                        const float* pw = &w->f[((size_t)ci * co_n + co) * k_n];
                        double a1 = 0, a2 = 0;
                        for (int k = (ph + pad) % r; k < k_n; k += r) {
                            a1 += std::fabs((double)pw[k]);
                            a2 += (double)pw[k] * pw[k];
                        }
                        su += a1 * u[ci];                    // leaky_relu(0.1) in front does not grow the bound
                        sm += a2 * 0.505 * m[ci];
                    }
                    ux[co] = std::max(ux[co], su);
                    mx[co] = std::max(mx[co], sm);
                }
            std::vector<double> us(co_n, 0.0), ms(co_n, 0.0);
            for (int j = 0; j < nk; ++j) {
                std::vector<double> x = ux, xm = mx, xt, xtm, y, ym, xa(co_n), xam(co_n);
                const std::string rb = v + "resblocks." + std::to_string(i * nk + j);
                for (int mth = 0; mth < 3; ++mth) {
                    for (int q = 0; q < co_n; ++q) {
                        pt_wc[pass].push_back(x[q]);                             // fp16 operand: leaky_relu(x) (and, iterations 1 / 2, the stored fp16 stream)
                        pt_m2[pass].push_back(xm[q]);
                        xa[q] = x[q];
                        xam[q] = 0.505 * xm[q];
                    }
                    if (!conv(rb + ".convs1." + std::to_string(mth), xa, xam, xt, xtm)) return false;
                    for (int q = 0; q < co_n; ++q) {
                        pt_wc[pass].push_back(xt[q]);                            // fp16 operand: leaky_relu(xt)
                        pt_m2[pass].push_back(xtm[q]);
                        xtm[q] *= 0.505;
                    }
                    if (!conv(rb + ".convs2." + std::to_string(mth), xt, xtm, y, ym)) return false;
                    for (int q = 0; q < co_n; ++q) {
                        x[q] += y[q];
                        xm[q] += ym[q];
                    }
                }
                for (int q = 0; q < co_n; ++q) {
                    us[q] += x[q] / nk;
                    ms[q] += xm[q] / nk;      // (second moment of a mean of nk terms: at most their mean)
                }
            }
            u.swap(us);
            m.swap(ms);
        }
    }
    if (pt_wc[0].size() != pt_wc[1].size()) return false;
    double a_wc = 0, b_wc = 0, a_m2 = 0, b_m2 = 0;   // max_q a_q, max_q b_q (possibly different channels: that is the point)
    for (size_t q = 0; q < pt_wc[0].size(); ++q) {
        a_wc = std::max(a_wc, pt_wc[0][q]);
        b_wc = std::max(b_wc, pt_wc[1][q] - pt_wc[0][q]);
        a_m2 = std::max(a_m2, pt_m2[0][q]);
        b_m2 = std::max(b_m2, pt_m2[1][q] - pt_m2[0][q]);
    }
    h->wc_const = a_wc;
    h->wc_lin = b_wc;
    h->est_const = std::sqrt(a_m2);
    h->est_lin = std::sqrt(b_m2);
    return true;
}

int build_vocoder(dtts_ctx* h) {
    Need need{h, ""};
    const dtts_config& c = h->cfg;
    // DTTS_VOC_F16 (default): the six serial convolutions on bf16 hi/lo split operands, the ResBlocks on fp16 operands
    if (c.vocoder_precision != DTTS_VOC_BF16 && c.vocoder_precision != DTTS_VOC_BF16X3 && c.vocoder_precision != DTTS_VOC_F16)
        return fail(h, DTTS_E_INVAL, "vocoder_precision %d is not one of DTTS_VOC_BF16 / _BF16X3 / _F16", c.vocoder_precision);
    // the split-operand staging of conv_pre reads the caller's mel rows with 16-byte loads (vconv.hip): the row width must keep them aligned
    if (c.vocoder_precision == DTTS_VOC_F16 && c.audio_num_mel_bins % 4)
        return fail(h, DTTS_E_INVAL, "DTTS_VOC_F16 needs audio_num_mel_bins %% 4 == 0 (got %d); use DTTS_VOC_BF16X3", c.audio_num_mel_bins);
    if (!h->ovf_dev) {
        if (!(h->ovf_dev = (unsigned long long*)dev_alloc(h, sizeof(unsigned long long))) || hipMemset(h->ovf_dev, 0, sizeof(unsigned long long)) != hipSuccess)
            return fail(h, DTTS_E_NOMEM, "range-guard counter");
    }
    if (!h->bad_dev) {
        if (!(h->bad_dev = (unsigned*)dev_alloc(h, sizeof(unsigned))) || hipMemset(h->bad_dev, 0, sizeof(unsigned)) != hipSuccess)
            return fail(h, DTTS_E_NOMEM, "overflow-detector counter");
        void* hp = nullptr;
        if (hipHostMalloc(&hp, 64, hipHostMallocDefault) != hipSuccess) return fail(h, DTTS_E_NOMEM, "overflow-detector host word");
        h->bad_host = (volatile unsigned*)hp;
        *h->bad_host = 0;
    }
    const int eng = c.vocoder_precision == DTTS_VOC_BF16 ? ENG_BF16 : ENG_BF16X3;                                   // serial convolutions
    const int eng_rb = c.vocoder_precision == DTTS_VOC_F16 ? ENG_F16 : eng;                                         // ResBlock convolutions
    const std::string v = "vocoder.";
    bool ok = pack_plain(h, need, h->conv_pre, eng, v + "conv_pre", 1, 1, 3);
    h->ups.resize(c.n_upsamples);
    h->hop = 1;
    for (int i = 0; ok && i < c.n_upsamples; ++i) {
        const int u = c.upsample_rates[i], k = c.upsample_kernel_sizes[i];
        // OPTION (DTTS_TUNE bit 13, off by default): ups.1 — the most expensive serial convolution, 8.4 of their 18.9 MFLOP per frame — in the
        // fp16 two-product form (vconv.hip H2; weights packed as a single fp16 tensor).  Measured: vocoder -0.7 %, pipelined step -0.4 %,
        // waveform RMS error 4.6e-5 -> 7.2e-5 (gate 1e-4): the gain does not pay for a third of the gate's margin, so three products stay.
        const bool h2 = c.vocoder_precision == DTTS_VOC_F16 && i == 1 && DTTS_TUNE(h, 8192) && (u * (c.upsample_initial_channel >> (i + 1))) % 256 == 0 &&
                        (c.upsample_initial_channel >> i) % 128 == 0;
        ok = ok && pack_transposed(h, need, h->ups[i], h2 ? ENG_F16 : eng, v + "ups." + std::to_string(i), u, (k - u) / 2);
        h->hop *= u;
    }
    const int nk = c.n_resblock_kernels;
    h->rb1.assign((size_t)c.n_upsamples * nk, {});
    h->rb2.assign((size_t)c.n_upsamples * nk, {});
    for (int i = 0; ok && i < c.n_upsamples * nk; ++i) {
        const int j = i % nk, k = c.resblock_kernel_sizes[j];
        h->rb1[i].resize(3);
        h->rb2[i].resize(3);
        for (int mth = 0; ok && mth < 3; ++mth) {
            const int d = c.resblock_dilation_sizes[j][mth];
            const std::string r = v + "resblocks." + std::to_string(i);
            ok = ok && pack_plain(h, need, h->rb1[i][mth], eng_rb, r + ".convs1." + std::to_string(mth), d, 1, (k * d - d) / 2);
            ok = ok && pack_plain(h, need, h->rb2[i][mth], eng_rb, r + ".convs2." + std::to_string(mth), 1, 1, (k - 1) / 2);
        }
    }
    // fused ResBlock kernel (bf16 mode, narrow stages): the same weights with the tap axis zero padded so that the
    // number of k-steps is a multiple of the register ring depth
    h->rbf1.assign((size_t)c.n_upsamples * nk, {});
    h->rbf2.assign((size_t)c.n_upsamples * nk, {});
    for (int i = 0; ok && eng_rb != ENG_BF16X3 && i < c.n_upsamples * nk; ++i) {
        const int j = i % nk, k = c.resblock_kernel_sizes[j];
        const int ch = c.upsample_initial_channel >> (i / nk + 1);
        if (!rblock_supported(ch, k) || (DTTS_TUNE(h, 8) && ch >= 128)) {   // DTTS_TUNE bit 3: the wide stages' k = 3 ResBlocks per iteration (vpair) again
            bool vp = h->rb1[i][0].C_in_pad == ch;
            for (int mth = 0; mth < 3; ++mth) vp = vp && vpair_supported(ch, k, c.resblock_dilation_sizes[j][mth]);
            if (eng_rb == ENG_F16 && !vp)
                return fail(h, DTTS_E_INVAL, "DTTS_VOC_F16 needs ResBlock widths 32/64/128/256 and odd kernels 3..11 (resblock %d: %d channels, k=%d); use DTTS_VOC_BF16X3", i, ch, k);
            continue;
        }
        const int kp = rblock_padded_taps(ch, k);
        h->rbf1[i].resize(3);
        h->rbf2[i].resize(3);
        for (int mth = 0; ok && mth < 3; ++mth) {
            const std::string r = v + "resblocks." + std::to_string(i);
            for (int which = 0; which < 2 && ok; ++which) {
                const std::string base = r + (which ? ".convs2." : ".convs1.") + std::to_string(mth);
                const HostTensor* w = folded_weight(h, need, base);
                std::vector<float> bias = bias_of(need, base);
                if (!w || bias.empty()) { ok = false; break; }
                const float* pw = w->f.data();
                PackedConv& L = which ? h->rbf2[i][mth] : h->rbf1[i][mth];
                const int slack = ch >= 64 ? 1 : 2;   // >= 4 zero k-steps behind the last tap: the weight prefetch never clamps
                ok = pack_conv(h, L, eng_rb, ch, ch, kp + slack,
                               [=](int co, int ci, int tap) { return tap < k ? pw[((size_t)co * ch + ci) * k + tap] : 0.f; }, bias,
                               which ? 1 : c.resblock_dilation_sizes[j][mth], 1, 0);
                L.K = k;
            }
        }
    }
    ok = ok && pack_plain(h, need, h->conv_post, eng, v + "conv_post", 1, 1, 3);
    {   // conv_post (C -> 1, k = 7) + tanh fused into the last stage's last ResBlock kernel when that stage runs on rblock at C = 32
        const int last_ch = c.upsample_initial_channel >> c.n_upsamples;
        const HostTensor* w = ok ? folded_weight(h, need, v + "conv_post") : nullptr;
        const std::vector<float> b = ok ? bias_of(need, v + "conv_post") : std::vector<float>();
        bool fusable = ok && eng_rb != ENG_BF16X3 && last_ch == 32 && w && w->shape.size() == 3 && w->shape[0] == 1 && w->shape[1] == 32 &&
                       w->shape[2] == 7 && b.size() == 1 && nk >= 2;
        for (int j = 0; fusable && j < nk; ++j) fusable = !h->rbf1[(size_t)(c.n_upsamples - 1) * nk + j].empty();
        if (fusable && !DTTS_TUNE(h, 1)) {
            std::vector<float> wt((size_t)7 * 32);
            for (int ci = 0; ci < 32; ++ci)
                for (int k = 0; k < 7; ++k) wt[(size_t)k * 32 + ci] = w->f[(size_t)ci * 7 + k];
            h->post_w = upload(h, wt);
            h->post_b = upload(h, b);
            ok = h->post_w && h->post_b;
        }
    }
    if (!ok) {
        if (!need.missing.empty()) return fail(h, DTTS_E_NOENT, "missing weight tensor '%s'", need.missing.c_str());
        if (h->err.empty()) return fail(h, DTTS_E_NOMEM, "packing / uploading vocoder weights failed");
        return DTTS_E_INVAL;
    }
    if (c.vocoder_precision == DTTS_VOC_F16 && !vocoder_fp16_analysis(h, need)) return fail(h, DTTS_E_NOENT, "fp16 analysis: missing weight tensor '%s'", need.missing.c_str());
    h->vocoder_ready = true;
    return DTTS_OK;
}

// ---------------------------------------------------------------------------------------------------------
// vconv parameter blocks (vocoder convolutions and the decoder's split-operand WaveNet layers)
// -DDTTS_ABLATE builds: phase-ablation bits for the vocoder kernels, read ONCE when the library is loaded (never per launch)
#ifdef DTTS_ABLATE
static const int g_ablate = ablate_env("DTTS_VCONV_DBG") ? atoi(ablate_env("DTTS_VCONV_DBG")) : 0;
#else
constexpr int g_ablate = 0;
#endif

VConvParams vparams(const PackedConv& L, const unsigned short* x, const int* lens, int B, int T) {
    VConvParams p;
    memset(&p, 0, sizeof p);
    p.x = x;
    p.ldx = L.C_in_pad;
    p.w = (const uint4*)L.w_hi;
    p.bias = L.bias;
    p.lens = lens;
    p.B = B;
    p.T = T;
    p.C_in_pad = L.C_in_pad;
    p.C_out = L.C_out;
    p.C_out_pad = L.C_out_pad;
    p.K = L.K;
    p.dil = L.dil;
    p.pad = L.pad;
    p.slope = 1.f;
    p.div = 1.f;
    p.in_slope = 1.f;
    p.C_in = L.C_in;
    p.poly_half = L.poly_half;
    p.dbg = g_ablate & 15;
    return p;
}
// waveform-exact form: fp32 input [B][T][ld] (leaky_relu(in_slope) applied while staging), hi/lo split operands
VConvParams vparams_x3(const PackedConv& L, const float* xf, int ld, float in_slope, const int* lens, int B, int T) {
    VConvParams p = vparams(L, nullptr, lens, B, T);
    p.xf = xf;
    p.ldx = ld;
    p.in_slope = in_slope;
    p.wlo = (const uint4*)L.w_lo;
    p.h2 = L.engine == ENG_F16 ? 1 : 0;
    return p;
}

// ---------------------------------------------------------------------------------------------------------
// launch helpers
struct Timed {
    dtts_ctx* h;
    int which;
    hipStream_t s;
    hipEvent_t e1 = nullptr;
    Timed(dtts_ctx* h_, int which_, hipStream_t s_) : h(h_), which(which_), s(s_) {
        TimerSlot& t = h->timers[which];
        if (!t.enabled) return;
        if (which == DTTS_TIMER_VOC_CONV && h->voc_span) {   // inside a family span: count the launch, record nothing
            t.launches += 1;
            return;
        }
        if (t.used + 2 > t.pool.size()) {
            for (int i = 0; i < 256; ++i) {
                hipEvent_t e;
                if (hipEventCreate(&e) != hipSuccess) return;
                t.pool.push_back(e);
            }
        }
        hipEvent_t e0 = t.pool[t.used];
        e1 = t.pool[t.used + 1];
        t.used += 2;
        t.launches += 1;
        (void)hipEventRecord(e0, s);
    }
    void stop() {   // close the span now (the destructor closes it at scope exit otherwise)
        if (e1) (void)hipEventRecord(e1, s);
        e1 = nullptr;
    }
    ~Timed() { stop(); }
};

ConvParams base_params(const float* x, int ldx, int B, int T_in, int T_out, float* y, int ldy) {
    ConvParams p;
    memset(&p, 0, sizeof p);
    p.x = x;
    p.ldx = ldx;
    p.x_bstride = (long long)T_in * ldx;
    p.B = B;
    p.T_in = T_in;
    p.T_out = T_out;
    p.out_div = 1.f;
    p.out_mul = 1.f;
    p.y_bstride_rows = T_out;
    p.split = INT_MAX;
    p.seg[0].y = y;
    p.seg[0].ld = ldy;
    return p;
}
void set_res(ConvParams& p, int s, const float* res, int ld) {
    p.seg[s].res = res;
    p.seg[s].ld_res = ld;
}

#define LAUNCH(expr)                                                                                          \
    do {                                                                                                      \
        hipError_t _e = (expr);                                                                               \
        if (_e != hipSuccess) return fail(h, DTTS_E_HIP, "%s: %s", #expr, hipGetErrorString(_e));             \
    } while (0)

int run_encoder(dtts_ctx* h, const Encoder& E, float* x, float* hbuf, float* qkv, float* att, float* ff, float* out,
                const int* lens, int B, int T, hipStream_t s) {
    const int C = h->cfg.hidden_size, F = 4 * C;
    for (size_t i = 0; i < E.l.size(); ++i) {
        const EncLayer& l = E.l[i];
        LAUNCH(layernorm_launch(x, hbuf, l.g1, l.b1, 1e-4f, lens, 1, 0, B, T, C, s));
        ConvParams p = base_params(hbuf, C, B, T, T, qkv, 3 * C);
        LAUNCH(conv1d_launch(l.qkv, p, s));
        LAUNCH(mha_launch(qkv, att, lens, B, T, C, h->cfg.num_heads, s));
        p = base_params(att, C, B, T, T, x, C);
        set_res(p, 0, x, C);
        LAUNCH(conv1d_launch(l.o, p, s));
        LAUNCH(layernorm_launch(x, hbuf, l.g2, l.b2, 1e-4f, lens, 0, 0, B, T, C, s));
        p = base_params(hbuf, C, B, T, T, ff, F);
        p.in_lens = lens;
        p.post_act = 1;
        LAUNCH(conv1d_launch(l.ffn1, p, s));
        p = base_params(ff, F, B, T, T, x, C);
        p.in_lens = lens;
        p.out_lens = lens;
        p.zero_masked = 1;
        set_res(p, 0, x, C);
        LAUNCH(conv1d_launch(l.ffn2, p, s));
    }
    LAUNCH(layernorm_launch(x, out, E.lg, E.lb, 1e-4f, lens, 0, 1, B, T, C, s));
    return DTTS_OK;
}

// WN.forward with x_mask = 1 (modules/commons/wavenet.py:54-78): x is updated in place, `out` receives the skip sum
// g == null: `cond` already holds the conditioning (the caller computed it)
int run_wn(dtts_ctx* h, const WNet& W, float* x, const float* g, int g_ld, float* cond, float* acts, float* out, int B,
           int T, hipStream_t s, const int64_t* cond_m2w = nullptr, int cond_Tw = 0) {
    const int H = W.hidden;
    ConvParams p;
    if (g) {
        p = base_params(g, g_ld, B, T, T, cond, 2 * H * W.layers);
        LAUNCH(conv1d_launch(W.cond, p, s));
    }
    for (int i = 0; i < W.layers; ++i) {
        if (W.in[i].engine == ENG_BF16X3) {   // split-operand WaveNet layer on the vconv kernel (vconv.hip: WaveNet epilogue)
            {   // acts = tanh(in(x) + cond_t) * sigmoid(in(x) + cond_s)
                VConvParams v = vparams_x3(W.in[i], x, H, 1.f, nullptr, B, T);
                v.bias = nullptr;
                v.gbias = W.in[i].bias;
                v.gate_H = H;
                v.cond = cond;
                v.ld_cond = 2 * H * W.layers;
                v.cond_coff = i * 2 * H;
                v.cond_m2w = (const long long*)cond_m2w;   // word-level conditioning gathered in the epilogue (decoder)
                v.cond_Tw = cond_Tw;
                v.yf = acts;
                v.ldyf = H;
                LAUNCH(vconv_launch(v, s));
            }
            {   // res / skip: x += rs[:H], out (+)= rs[H:]  (the last layer has the skip half only)
                VConvParams v = vparams_x3(W.rs[i], acts, H, 1.f, nullptr, B, T);
                v.bias = nullptr;
                v.gbias = W.rs[i].bias;
                if (i < W.layers - 1) {
                    v.split = H;
                    v.yf = x;
                    v.ldyf = H;
                    v.res = x;
                    v.ldres = H;
                    v.yf2 = out;
                    v.ldyf2 = H;
                    if (i > 0) {
                        v.res_b = out;
                        v.ldres_b = H;
                    }
                } else {
                    v.split = 1 << 30;   // single segment through the same epilogue
                    v.yf = out;
                    v.ldyf = H;
                    if (i > 0) {
                        v.res = out;
                        v.ldres = H;
                    }
                }
                LAUNCH(vconv_launch(v, s));
            }
            continue;
        }
        p = base_params(x, H, B, T, T, acts, H);
        p.cond = cond;
        p.ld_cond = 2 * H * W.layers;
        p.cond_coff = i * 2 * H;
        LAUNCH(conv1d_launch(W.in[i], p, s));
        p = base_params(acts, H, B, T, T, x, H);
        if (i < W.layers - 1) {
            p.split = H;
            set_res(p, 0, x, H);
            p.seg[1].y = out;
            p.seg[1].ld = H;
            if (i > 0) set_res(p, 1, out, H);
        } else {
            p.seg[0].y = out;
            if (i > 0) set_res(p, 0, out, H);
        }
        LAUNCH(conv1d_launch(W.rs[i], p, s));
    }
    return DTTS_OK;
}

} // namespace


// ---------------------------------------------------------------------------------------------------------
// HifiGAN, bf16 mode: vconv kernels.  Every activation exists twice: the fp32 residual stream and the bf16
// leaky_relu copy the next convolution consumes (written by the producer's epilogue).
namespace {

struct StageMult { int m[9]; };  // cumulative upsampling factor per stage, passed by value (no H2D copy on the stream)
__global__ void scale_lens_kernel2(const int32_t* lens, int32_t* out, int B, int T, int n_stage, StageMult mult) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * n_stage) return;
    const int sidx = i / B, b = i % B;
    int l = lens ? lens[b] : T;
    l = l < 0 ? 0 : (l > T ? T : l);
    out[i] = l * mult.m[sidx];
}

// wav[b][i] = 0 for i >= lens[b] * hop: the samples past an utterance's end (the generator's last kernel writes the valid ones only).  Round 5
// zero-filled the WHOLE [B][T * hop] buffer with hipMemsetAsync in front of every forward — 45 MB at B = 60, half of it about to be overwritten,
// through rocclr's generic fill kernel (51 us on average beside the other stream's kernels, up to 395 us: profiles/r05_h_kernel_trace.md).
__global__ void zero_wav_tails_kernel(float* wav, const int32_t* lens, int T, int hop) {
    const int b = blockIdx.y;
    const long long n = (long long)T * hop;
    int l = lens[b];
    l = l < 0 ? 0 : (l > T ? T : l);
    const long long first = (long long)l * hop;                       // (hop is a multiple of 4: 16-byte stores stay aligned)
    const long long i = first + ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) *(float4*)(wav + (long long)b * n + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    else for (long long k = i; k < n; ++k) wav[(long long)b * n + k] = 0.f;
}

// exact = DTTS_VOC_F16: fp32 tensors between kernels (no 16-bit copies), serial convolutions on split operands, ResBlock
// kernels on fp16 operands
int hifigan_forward_fused(dtts_ctx* h, const float* mel, const int32_t* lens, int B, int T, float* wav, hipStream_t s) {
    const dtts_config& c = h->cfg;
    const bool exact = c.vocoder_precision == DTTS_VOC_F16;
    const int el = exact ? EL_F16 : EL_BF16;
    const int nup = c.n_upsamples, nk = c.n_resblock_kernels;
    typedef unsigned short bf;
    size_t max_elems = (size_t)B * T * c.upsample_initial_channel;
    {
        size_t rows = T;
        int ch = c.upsample_initial_channel;
        for (int i = 0; i < nup; ++i) {
            rows *= c.upsample_rates[i];
            ch /= 2;
            max_elems = std::max<size_t>(max_elems, (size_t)B * rows * ch);
        }
    }
    const int melC = h->conv_pre.C_in_pad;
    const bool fuse = exact || !c.vocoder_unfused;   // vocoder_unfused: per-convolution kernels (a testing aid of the bf16 mode)
    // which stages run ALL their ResBlocks in one launch (rblock.hip; OPT-IN, tune bit 9 — measured: HBM traffic -0.36 MB / mel frame, vocoder
    // alone +1.1 %, pipelined step +1.9 %: LABNOTES round 4): C <= 64, every ResBlock has a whole-ResBlock kernel, and the batch has at least
    // two tiles per CU (small grids keep one launch per ResBlock: half-size tiles fill the chip there)
    int fuse_n[8] = {1, 1, 1, 1, 1, 1, 1, 1};   // ResBlocks in the stage's FIRST launch (1 = one launch per ResBlock)
    size_t s_elems = max_elems;   // capacity of the stage-sum buffer
    {
        long long rows = T;
        int ch = c.upsample_initial_channel;
        for (int i = 0; i < nup; ++i) {
            rows *= c.upsample_rates[i];
            ch /= 2;
            // (experiment, tune bit 12) C = 32 only: the first TWO ResBlocks in one launch — the k = 3 launch alone is HBM-bound (x in, stage sum out:
            // 4.6 TB/s), together with k = 7 its bytes ride on that launch's compute; the last ResBlock (fused conv_post) stays on its own
            if (fuse && DTTS_TUNE(h, 4096) && !DTTS_TUNE(h, 512) && nk == 3 && ch == 32 && !h->rbf1[(size_t)i * nk].empty() && !h->rbf1[(size_t)i * nk + 1].empty()) {
                const int k2 = std::max(h->rbf1[(size_t)i * nk][0].K, h->rbf1[(size_t)i * nk + 1][0].K), TT2 = 1024 - 12 * (k2 - 1);
                if (TT2 >= 64 && (long long)B * ((rows + TT2 - 1) / TT2) >= 2LL * h->n_cu) fuse_n[i] = 2;
                continue;
            }
            bool all = fuse && DTTS_TUNE(h, 512) && nk >= 2 && nk <= 3 && (ch == 32 || ch == 64);
            int kmax = 0;
            for (int j = 0; j < nk && all; ++j) {
                all = !h->rbf1[(size_t)i * nk + j].empty();
                if (all) kmax = std::max(kmax, h->rbf1[(size_t)i * nk + j][0].K);
            }
            if (!all) continue;
            const int W = ch == 32 ? 1024 : 512, TT = W - 12 * (kmax - 1);
            if (TT < 64 || (long long)B * ((rows + TT - 1) / TT) < 2LL * h->n_cu) continue;
            if (i == nup - 1 && h->post_w) {   // with the fused conv_post the tiles overlap: one private strip of the stage sum per tile
                const long long prow = rblock_private_rows(ch, kmax, B, (int)rows);
                if (prow <= 0 || (size_t)prow * ch * sizeof(float) >= (size_t)INT_MAX) continue;
                s_elems = std::max(s_elems, (size_t)prow * ch);
            }
            fuse_n[i] = nk;
        }
    }
    const size_t s_cap_bytes = s_elems * sizeof(float);
    HIPCHK(h->a_voc.reserve((s_elems - max_elems) * sizeof(float) + max_elems * (4 * sizeof(float) + 4 * sizeof(bf)) + (size_t)B * T * melC * sizeof(bf) +
                            (size_t)(nup + 2) * B * sizeof(int) + (64 << 10), s));
    Arena& A = h->a_voc;
    float* Xf = A.alloc<float>(max_elems);
    float* Rf = A.alloc<float>(max_elems);
    float* Sf = A.alloc<float>(s_elems);
    float* Rg = A.alloc<float>(max_elems);   // second ping-pong buffer of the fused-iteration path (vpair.hip)
    bf* Xa = A.alloc<bf>(max_elems);
    bf* Ra = A.alloc<bf>(max_elems);
    bf* Ta = A.alloc<bf>(max_elems);
    bf* Sa = A.alloc<bf>(max_elems);
    bf* melb = A.alloc<bf>((size_t)B * T * melC);
    int* lensS = A.alloc<int>((size_t)(nup + 1) * B);
    // one tile counter per launch of a persistent kernel (vpair / rblock: dynamic tile claiming), zeroed once per forward
    constexpr int N_CTR = 64;
    unsigned* ctrs = A.alloc<unsigned>(N_CTR);
    int n_ctr = 0;
    if (!Xf || !Rf || !Sf || !Rg || !Xa || !Ra || !Ta || !Sa || !melb || !lensS || !ctrs) return fail(h, DTTS_E_NOMEM, "vocoder workspace");
    const bool dyn_tiles = !DTTS_TUNE(h, 4);   // DTTS_TUNE bit 2: static tile assignment (round 2)
    if (dyn_tiles) HIPCHK(hipMemsetAsync(ctrs, 0, N_CTR * sizeof(unsigned), s));
    {
        StageMult mult;
        mult.m[0] = 1;
        for (int i = 0; i < nup; ++i) mult.m[i + 1] = mult.m[i] * c.upsample_rates[i];
        hipLaunchKernelGGL(scale_lens_kernel2, dim3((B * (nup + 1) + 255) / 256), dim3(256), 0, s, lens, lensS, B, T, nup + 1, mult);
    }
    // samples past an utterance's end are zero (no lens: every sample is a valid one and is written below)
    if (lens && (h->hop & 3) == 0 && ((uintptr_t)wav & 15) == 0) {
        const long long n = (long long)T * h->hop;
        hipLaunchKernelGGL(zero_wav_tails_kernel, dim3((unsigned)((n / 4 + 255) / 256), B), dim3(256), 0, s, wav, lens, T, h->hop);
        HIPCHK(hipGetLastError());
    } else if (lens) HIPCHK(hipMemsetAsync(wav, 0, (size_t)B * T * h->hop * sizeof(float), s));
    const int TV = DTTS_TIMER_VOC_CONV;
    // The family's launches are consecutive on the stream (nothing else runs between conv_pre and the last ResBlock / conv_post): ONE
    // hipEvent pair per forward spans them all — the per-launch pairs of round 2 put 50 event packets between the kernels of every forward
    // (DTTS_TUNE bit 4 brings them back).  The span includes the kernel boundaries; launches are still counted one by one.
    struct SpanGuard {
        dtts_ctx* h;
        Timed* t;
        ~SpanGuard() {
            h->voc_span = false;
            delete t;   // closes the span (records the end event)
        }
    } span{h, nullptr};
    if (h->timers[TV].enabled && !DTTS_TUNE(h, 16)) {
        span.t = new Timed(h, TV, s);
        if (span.t->e1) h->timers[TV].launches -= 1;   // (the span itself is not a launch; e1 is null when no event could be created)
        h->voc_span = true;
    }
    bool post_done = false;
    int Tcur = T, ch = c.upsample_initial_channel;
    if (exact) {   // conv_pre: mel fp32 in, fp32 out
        VConvParams p = vparams_x3(h->conv_pre, mel, c.audio_num_mel_bins, 1.f, lensS, B, T);
        p.yf = Sf;
        p.ldyf = ch;
        Timed tm(h, TV, s);
        LAUNCH(vconv_launch(p, s));
    } else {   // conv_pre: only its leaky_relu(0.1) bf16 copy is consumed (by ups[0])
        LAUNCH(f32_to_bf16_pad_launch(mel, melb, (long long)B * T, c.audio_num_mel_bins, melC, s));
        VConvParams p = vparams(h->conv_pre, melb, lensS, B, T);
        p.ya = Sa;
        p.ldya = ch;
        p.slope = 0.1f;
        Timed tm(h, TV, s);
        LAUNCH(vconv_launch(p, s));
    }
    for (int i = 0; i < nup; ++i) {
        const int u = c.upsample_rates[i];
        const int* lin = lensS + (size_t)i * B;
        const int* lout = lensS + (size_t)(i + 1) * B;
        ch /= 2;
        // the bf16 leaky_relu copy of the stage input is consumed only by the per-convolution fallback: the fused kernels
        // (vpair, rblock) read the fp32 stream and round it themselves
        bool need_xa = false;
        for (int j = 0; j < nk; ++j) {
            const auto& c1 = h->rb1[(size_t)i * nk + j];
            const bool fused_rb = fuse && !h->rbf1[(size_t)i * nk + j].empty();
            const bool fused_vp = fuse && vpair_supported(ch, c1[0].K, c1[0].dil) && vpair_supported(ch, c1[2].K, c1[2].dil) && c1[0].C_in_pad == ch;
            need_xa = need_xa || !(fused_rb || fused_vp);
        }
        {   // ups[i] (polyphase): Sa [B,Tcur,2ch] -> Xf / Xa [B,Tcur,u*ch] == [B,Tcur*u,ch]
            VConvParams p = exact ? vparams_x3(h->ups[i], Sf, 2 * ch, 0.1f, lin, B, Tcur) : vparams(h->ups[i], Sa, lin, B, Tcur);
            p.small_tiles = exact && !DTTS_TUNE(h, 32);   // narrow split-operand upsamplers: 64-row tiles, 4 workgroups / CU (-0.15 ms same-box)
            p.yf = Xf;
            p.ldyf = u * ch;
            p.ya = need_xa ? Xa : nullptr;
            p.ldya = u * ch;
            p.slope = 0.1f;
            Timed tm(h, TV, s);
            LAUNCH(vconv_launch(p, s));
        }
        Tcur *= u;
        const bool last_stage = i == nup - 1;
        for (int j = 0; j < nk; ++j) {
            const auto& c1 = h->rb1[(size_t)i * nk + j];
            const auto& c2 = h->rb2[(size_t)i * nk + j];
            if (fuse && !h->rbf1[(size_t)i * nk + j].empty()) {   // whole ResBlock in one kernel (rblock.hip)
                const auto& f1 = h->rbf1[(size_t)i * nk + j];
                const auto& f2 = h->rbf2[(size_t)i * nk + j];
                RBlockParams rp;
                memset(&rp, 0, sizeof rp);
                rp.x = Xf;
                rp.S = Sf;
                rp.lens = lout;
                rp.B = B;
                rp.T = Tcur;
                auto fill_set = [&](RBlockParams::Set& st, int jj) {
                    const auto& g1 = h->rbf1[(size_t)i * nk + jj];
                    const auto& g2 = h->rbf2[(size_t)i * nk + jj];
                    st.K = g1[0].K;
                    st.Kp = rblock_padded_taps(ch, g1[0].K);
                    for (int mth = 0; mth < 3; ++mth) {
                        st.w1[mth] = (const uint4*)g1[mth].w_hi;
                        st.w2[mth] = (const uint4*)g2[mth].w_hi;
                        st.b1[mth] = g1[mth].bias;
                        st.b2[mth] = g2[mth].bias;
                        st.dil[mth] = g1[mth].dil;
                    }
                };
                // ALL ResBlocks of a C <= 64 stage in ONE launch (rblock.hip: work items (tile, ResBlock)): x crosses HBM once per tile and the
                // stage sum is accumulated through L2 / Infinity Cache.  Opt-in: tune bit 9
                if (j > 0 && j < fuse_n[i]) continue;      // (launched with j = 0)
                const int j_last = j == 0 ? fuse_n[i] - 1 : j;
                const bool stage_fused = fuse_n[i] == nk;
                rp.nrb = j_last - j + 1;
                rp.last_mode = j_last == nk - 1 ? 2 : 1;
                rp.K = 0;
                for (int jj = j; jj <= j_last; ++jj) {
                    fill_set(rp.rb[jj - j], jj);
                    rp.K = std::max(rp.K, rp.rb[jj - j].K);
                }
                (void)f1;
                (void)f2;
                rp.mode = j == 0 ? 0 : (j == nk - 1 ? 2 : 1);
                if (nk == 1) rp.mode = 2;
                rp.div = (float)nk;
                rp.slope = last_stage ? 0.01f : 0.1f;
                rp.Sa = exact ? nullptr : Sa;
                rp.drop_S = exact ? 0 : 1;   // bf16 mode: after a stage only its bf16 leaky_relu copy is consumed (by ups[i+1] / conv_post)
                if (last_stage && j_last == nk - 1 && h->post_w) {   // conv_post + tanh in this kernel's epilogue: the stage output stays on chip
                    rp.wav = wav;
                    rp.post_w = h->post_w;
                    rp.post_b = h->post_b;
                    rp.Sa = nullptr;
                    post_done = true;
                    if (stage_fused) rp.s_private = (int)std::min<size_t>(s_cap_bytes, (size_t)INT_MAX);   // one private strip of S per tile (the tiles overlap)
                }
                rp.el = el;
                rp.tile_ctr = (dyn_tiles && n_ctr < N_CTR) ? ctrs + n_ctr++ : nullptr;
                rp.ovf = (exact && h->guard_on) ? h->ovf_dev : nullptr;
                rp.bad = h->bad_dev;
                rp.small_tile = DTTS_TUNE(h, 16384) ? 1 : 0;
                rp.pingpong = DTTS_TUNE(h, 128) ? 1 : 0;   // tune bit 7 (-DDTTS_ABLATE builds only): the two-group form of rblock2.hip (experiment)
                rp.dbg = (g_ablate >> 4) & 15;
                if (nk == 1) return fail(h, DTTS_E_INVAL, "fused ResBlock path needs >= 2 resblock kernels");
#ifdef DTTS_ABLATE
                static unsigned long long* rstats = nullptr;
                if (ablate_env("DTTS_RB_STATS")) {   // per-phase cycle sums of rblock2's two groups, printed per launch
                    if (!rstats) hipMalloc((void**)&rstats, 18 * sizeof(unsigned long long));
                    hipMemsetAsync(rstats, 0, 18 * sizeof(unsigned long long), s);
                    rp.stats = rstats;
                }
#endif
                {
                    Timed tm(h, TV, s);
                    LAUNCH(rblock_launch(rp, ch, s));
                }
#ifdef DTTS_ABLATE
                if (rp.stats) {
                    unsigned long long hs[18];
                    hipStreamSynchronize(s);
                    hipMemcpy(hs, rp.stats, sizeof hs, hipMemcpyDeviceToHost);
                    for (int g = 0; g < 2 && hs[8]; ++g) {
                        const double n = hs[g * 9 + 8] ? (double)hs[g * 9 + 8] : 1.0;
                        const unsigned long long* a = hs + g * 9;
                        fprintf(stderr, "rblock2 C=%d K=%d grp %d tiles=%llu cycles/tile: write_x %.0f  bar_after_N %.0f  conv1 %.0f  bar_after_M %.0f  rewrite_xt %.0f  conv2 %.0f  rewrite_x %.0f  epilogue %.0f\n",
                                ch, rp.K, g, a[8], a[0] / n, a[1] / n, a[2] / n, a[3] / n, a[4] / n, a[5] / n, a[6] / n, a[7] / n);
                    }
                }
#endif
                continue;
            }
            if (fuse && vpair_supported(ch, c1[0].K, c1[0].dil) && vpair_supported(ch, c1[2].K, c1[2].dil) && c1[0].C_in_pad == ch) {
                // one kernel per ResBlock iteration (vpair.hip): fp32 stream in, fp32 stream out
                const float* xin = Xf;
                for (int mth = 0; mth < 3; ++mth) {
                    VPairParams vp;
                    memset(&vp, 0, sizeof vp);
                    vp.x = xin;
                    vp.w1 = (const uint4*)c1[mth].w_hi;
                    vp.w2 = (const uint4*)c2[mth].w_hi;
                    vp.b1 = c1[mth].bias;
                    vp.b2 = c2[mth].bias;
                    vp.lens = lout;
                    vp.B = B;
                    vp.T = Tcur;
                    vp.K = c1[mth].K;
                    vp.dil = c1[mth].dil;
                    vp.div = (float)nk;
                    vp.slope = last_stage ? 0.01f : 0.1f;
                    vp.el = el;
                    vp.tile_ctr = (dyn_tiles && n_ctr < N_CTR) ? ctrs + n_ctr++ : nullptr;
                    vp.ovf = (exact && h->guard_on) ? h->ovf_dev : nullptr;
                    vp.dbg = g_ablate >> 8;
#ifdef DTTS_ABLATE
                    if (ablate_env("DTTS_VP_STATS")) {   // per-phase cycles of wave 0 (staging, c1, rewrite, c2, epilogue incl. store acks), printed per launch
                        static unsigned long long* dstats = nullptr;
                        if (!dstats) hipMalloc((void**)&dstats, 64);
                        hipMemsetAsync(dstats, 0, 64, s);
                        vp.stats = dstats;
                    }
#endif
                    // round 6: the stream BETWEEN the three iterations is fp16 (DTTS_VOC_F16 only; tune bit 15: fp32 as in round 5).  fp16(x) is what the
                    // next iteration's convolution operand was anyway; the residual add sees the rounded value (tools/precision_sim.py --stream:
                    // waveform error 5.3e-5 -> 6.7e-5, gate 1e-4).  The ResBlock's RESULT (iteration 2) stays fp32.
                    const bool s16 = exact && !DTTS_TUNE(h, 32768);
                    // (ablation builds: DTTS_S16 = mask of the hops that are 16-bit — bit 2 i: iteration 0 -> 1 of stage i, bit 2 i + 1: iteration 1 -> 2)
                    static const int s16m = ablate_env("DTTS_S16") ? atoi(ablate_env("DTTS_S16")) : ~0;
                    const bool hop_a = s16 && ((s16m >> (2 * i)) & 1), hop_b = s16 && ((s16m >> (2 * i + 1)) & 1);
                    vp.x16 = (mth == 1 ? hop_a : (mth == 2 ? hop_b : false)) ? 1 : 0;
                    if (mth < 2) {
                        vp.y = mth == 0 ? Rf : Rg;
                        vp.mode = 1;
                        vp.y16 = (mth == 0 ? hop_a : hop_b) ? 1 : 0;
                    } else {
                        vp.y = Sf;
                        vp.mode = j == 0 ? 1 : (j == nk - 1 ? 3 : 2);
                        vp.ya = exact ? nullptr : Sa;
                        vp.drop_y = exact ? 0 : 1;   // bf16 mode: after a stage only its bf16 leaky_relu copy is consumed (by ups[i+1])
                    }
                    xin = vp.y;
                    Timed tm(h, TV, s);
                    LAUNCH(vpair_launch(vp, ch, s));
#ifdef DTTS_ABLATE
                    if (vp.stats) {
                        unsigned long long hs[8];
                        hipStreamSynchronize(s);
                        hipMemcpy(hs, vp.stats, 64, hipMemcpyDeviceToHost);
                        const double n = hs[5] ? (double)hs[5] : 1.0;
                        fprintf(stderr, "vpair C=%d K=%d d=%d mode=%d tiles=%llu  cycles/tile: stage %.0f c1 %.0f rewrite %.0f c2 %.0f epilogue %.0f\n", ch, vp.K, vp.dil,
                                vp.mode, hs[5], hs[0] / n, hs[1] / n, hs[2] / n, hs[3] / n, hs[4] / n);
                    }
#endif
                }
                continue;
            }
            if (exact) return fail(h, DTTS_E_STATE, "DTTS_VOC_F16: resblock %d has no fused kernel", i * nk + j);   // build_vocoder rejects such configs
            for (int mth = 0; mth < 3; ++mth) {
                {   // xt = c1(leaky_relu(x)); only leaky_relu(xt) in bf16 is ever consumed
                    VConvParams p = vparams(c1[mth], mth == 0 ? Xa : Ra, lout, B, Tcur);
                    p.ya = Ta;
                    p.ldya = ch;
                    p.slope = 0.1f;
                    Timed tm(h, TV, s);
                    LAUNCH(vconv_launch(p, s));
                }
                {   // x = c2(leaky_relu(xt)) + x
                    VConvParams p = vparams(c2[mth], Ta, lout, B, Tcur);
                    p.res = mth == 0 ? Xf : Rf;
                    p.ldres = ch;
                    if (mth < 2) {
                        p.yf = Rf;
                        p.ldyf = ch;
                        p.ya = Ra;
                        p.ldya = ch;
                        p.slope = 0.1f;
                    } else {   // xs (+)= x ; the last resblock also applies / num_kernels and emits the next stage's input
                        p.yf = Sf;
                        p.ldyf = ch;
                        if (j > 0) {
                            p.res2 = Sf;
                            p.ldres2 = ch;
                        }
                        if (j == nk - 1) {
                            p.div = (float)nk;
                            p.ya = Sa;
                            p.ldya = ch;
                            p.slope = last_stage ? 0.01f : 0.1f;  // F.leaky_relu default before conv_post (hifigan.py:138)
                        }
                    }
                    Timed tm(h, TV, s);
                    LAUNCH(vconv_launch(p, s));
                }
            }
        }
    }
    if (!post_done) {   // wav = tanh(conv_post(leaky_relu(x, 0.01)))
        VConvParams p = exact ? vparams_x3(h->conv_post, Sf, ch, 0.01f, lensS + (size_t)nup * B, B, Tcur)
                              : vparams(h->conv_post, Sa, lensS + (size_t)nup * B, B, Tcur);
        p.yf = wav;
        p.ldyf = 1;
        p.post_tanh = 1;
        p.bad = h->bad_dev;
        Timed tm(h, TV, s);
        LAUNCH(vconv_launch(p, s));
    }
    return DTTS_OK;
}

} // namespace

// =========================================================================================================
// C ABI
// =========================================================================================================
extern "C" {

int dtts_config_sizeof(void) { return (int)sizeof(dtts_config); }

void dtts_default_config(dtts_config* c) {
    memset(c, 0, sizeof *c);
    c->hidden_size = 192;
    c->num_heads = 2;
    c->enc_ffn_kernel_size = 5;
    c->enc_layers = 4;
    c->gloss_dim = 768;
    c->word_size = 8000;
    c->value_embedding_size = 185;
    c->n_phone = 6;
    c->audio_num_mel_bins = 80;
    c->latent_size = 16;
    c->fvae_enc_dec_hidden = 192;
    c->fvae_kernel_size = 5;
    c->fvae_dec_n_layers = 4;
    c->fvae_enc_n_layers = 8;
    c->prior_glow_hidden = 64;
    c->glow_kernel_size = 3;
    c->prior_glow_n_blocks = 4;
    c->prior_glow_n_layers = 4;
    c->dur_predictor_layers = 3;
    c->dur_predictor_kernel = 5;
    c->dur_chans = 128;
    c->frames_multiple = 4;
    c->language_zh = 1;
    c->upsample_initial_channel = 512;
    c->n_upsamples = 4;
    const int ur[4] = {8, 8, 2, 2}, uk[4] = {16, 16, 4, 4}, rk[3] = {3, 7, 11};
    for (int i = 0; i < 4; ++i) {
        c->upsample_rates[i] = ur[i];
        c->upsample_kernel_sizes[i] = uk[i];
    }
    c->n_resblock_kernels = 3;
    for (int i = 0; i < 3; ++i) {
        c->resblock_kernel_sizes[i] = rk[i];
        c->resblock_dilation_sizes[i][0] = 1;
        c->resblock_dilation_sizes[i][1] = 3;
        c->resblock_dilation_sizes[i][2] = 5;
    }
    c->vocoder_precision = DTTS_VOC_F16;
    c->fft_layers = 4;
    c->fft_kernel_size = 9;
    c->fft_use_pos_embed = 1;
    c->fft_use_last_norm = 1;
}

int dtts_create(const dtts_config* cfg, dtts_handle* out) {
    if (!cfg || !out) return fail(nullptr, DTTS_E_INVAL, "dtts_create: null argument");
    if (cfg->tune_flags & ~TUNE_MASK) {   // (release library: an untested experiment's bit is refused, never silently ignored)
        char msg[160];
        snprintf(msg, sizeof msg, "dtts_create: tune_flags 0x%x carries bits this build does not honour (supported mask 0x%x; the others exist only in "
                 "-DDTTS_ABLATE builds)", (unsigned)cfg->tune_flags, (unsigned)TUNE_MASK);
        return fail(nullptr, DTTS_E_INVAL, "%s", msg);
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(nullptr, DTTS_E_HIP, "dtts_create: no HIP device visible (the HIP path has no CPU fallback)");
    if (cfg->hidden_size % 64 || cfg->hidden_size / cfg->num_heads > 96 || cfg->gloss_dim > 768 || cfg->gloss_dim % 4 ||
        cfg->n_upsamples > 8 || cfg->n_resblock_kernels > 4 || cfg->latent_size % 8 || cfg->prior_glow_hidden % 32 ||
        cfg->fvae_enc_dec_hidden % 32)
        return fail(nullptr, DTTS_E_INVAL, "dtts_create: unsupported configuration");
    dtts_ctx* h = new dtts_ctx();
    h->cfg = *cfg;
    // A/B switches of tuning experiments: dtts_config.tune_flags (0 = the measured defaults; bits documented in include/dicttts_hip.h).
    // The release library does NOT read the environment; -DDTTS_ABLATE builds (tools/ab_*.sh) OR the DTTS_TUNE variable in.
    h->tune = cfg->tune_flags;
    if (const char* e = ablate_env("DTTS_TUNE")) h->tune |= atoi(e);
    {   // prior-sample seed: different per context, process, device and start time (data-parallel ranks and restarts must not draw the
        // same z_p sequence); dtts_set_noise_seed makes it reproducible
        static unsigned long long instance = 0;
        int dev = 0;
        (void)hipGetDevice(&dev);
        unsigned long long z = (unsigned long long)time(nullptr) * 0x9E3779B97F4A7C15ull ^ ((unsigned long long)getpid() << 32) ^
                               ((unsigned long long)dev << 20) ^ ++instance;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        h->noise_seed = z ^ (z >> 31);
    }
    h->guard_on = cfg->vocoder_range_guard != 0;
    (void)hipGetDevice(&h->device);
    if (hipDeviceGetAttribute(&h->n_cu, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess || h->n_cu <= 0) h->n_cu = 256;
    h->debug_rz = cfg->debug_redzone != 0;
    h->a_fft.debug = h->a_enc.debug = h->a_dec.debug = h->a_voc.debug = h->debug_rz;
    *out = h;
    return DTTS_OK;
}

void dtts_destroy(dtts_handle h) {
    if (h && h->bad_host) {
        (void)hipDeviceSynchronize();
        (void)hipHostFree((void*)h->bad_host);
        h->bad_host = nullptr;
    }
    if (!h) return;
    (void)hipDeviceSynchronize();
    for (void* p : h->allocs) (void)hipFree(p);
    if (h->amax_bits) (void)hipFree(h->amax_bits);
    h->a_fft.release();
    h->a_enc.release();
    h->a_dec.release();
    h->a_voc.release();
    for (auto& t : h->timers)
        for (auto e : t.pool) (void)hipEventDestroy(e);
    delete h;
}

const char* dtts_last_error(dtts_handle h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int dtts_load_weight(dtts_handle h, const char* name, const void* host_ptr, const int64_t* shape, int ndim, int dtype) {
    if (!h || !name || !host_ptr || ndim < 0 || ndim > 8) return fail(h, DTTS_E_INVAL, "dtts_load_weight: bad argument");
    if (dtype != DTTS_F32) return fail(h, DTTS_E_INVAL, "dtts_load_weight(%s): only fp32 tensors are accepted", name);
    HostTensor t;
    t.shape.assign(shape, shape + ndim);
    const int64_t n = t.numel();
    t.f.assign((const float*)host_ptr, (const float*)host_ptr + n);
    h->w[name] = std::move(t);
    return DTTS_OK;
}

// ---------------------------------------------------------------------------------------------------------
// FFT block stack (FFTBlocks / EncSALayer): the same fp32-MFMA convolution, attention and LayerNorm
// kernels as the S2PA encoders, with torch-LayerNorm eps, bias-free attention projections and the k**-0.5 GELU FFN
namespace {
int build_fft(dtts_ctx* h) {
    Need need{h, ""};
    const dtts_config& c = h->cfg;
    const int C = c.hidden_size, K = c.fft_kernel_size;
    if (c.fft_layers <= 0 || K <= 0 || !(K & 1) || C % c.num_heads || C / c.num_heads > 96)
        return fail(h, DTTS_E_INVAL, "FFT blocks: unsupported configuration (layers=%d kernel=%d hidden=%d heads=%d)", c.fft_layers, K, C,
                    c.num_heads);
    h->fft.resize(c.fft_layers);
    bool ok = true;
    for (int i = 0; i < c.fft_layers && ok; ++i) {
        dtts_ctx::FftLayer& l = h->fft[i];
        const std::string p = "fft.layers." + std::to_string(i) + ".op.";
        const HostTensor* win = need.get(p + "self_attn.in_proj_weight");   // [3C][C], no bias (EncSALayer: bias=False)
        const HostTensor* wout = need.get(p + "self_attn.out_proj.weight");
        if (!win || !wout) {
            ok = false;
            break;
        }
        if (win->numel() != (int64_t)3 * C * C || wout->numel() != (int64_t)C * C)
            return fail(h, DTTS_E_INVAL, "%sself_attn: projection shapes do not match hidden_size %d", p.c_str(), C);
        const float *pi = win->f.data(), *po = wout->f.data();
        ok = ok && pack_conv(h, l.qkv, ENG_F32, 3 * C, C, 1, [=](int co, int ci, int) { return pi[(size_t)co * C + ci]; },
                             std::vector<float>(), 1, 1, 0);
        ok = ok && pack_conv(h, l.o, ENG_F32, C, C, 1, [=](int co, int ci, int) { return po[(size_t)co * C + ci]; },
                             std::vector<float>(), 1, 1, 0);
        ok = ok && pack_plain(h, need, l.ffn1, ENG_F32, p + "ffn.ffn_1", 1, 1, K / 2);
        ok = ok && pack_plain(h, need, l.ffn2, ENG_F32, p + "ffn.ffn_2", 1, 1, 0);
        if (ok && (l.ffn1.K != K || l.ffn1.C_out != 4 * C))
            return fail(h, DTTS_E_INVAL, "%sffn.ffn_1: kernel %d / width %d differ from the configuration (%d / %d)", p.c_str(), l.ffn1.K,
                        l.ffn1.C_out, K, 4 * C);
        l.g1 = upload_named(h, need, p + "layer_norm1.weight");
        l.b1 = upload_named(h, need, p + "layer_norm1.bias");
        l.g2 = upload_named(h, need, p + "layer_norm2.weight");
        l.b2 = upload_named(h, need, p + "layer_norm2.bias");
        ok = ok && l.g1 && l.b1 && l.g2 && l.b2;
    }
    if (ok && c.fft_use_last_norm) {
        h->fft_g = upload_named(h, need, "fft.layer_norm.weight");
        h->fft_b = upload_named(h, need, "fft.layer_norm.bias");
        ok = h->fft_g && h->fft_b;
    }
    if (ok && c.fft_use_pos_embed && h->w.count("fft.pos_embed_alpha"))   // absent with use_pos_embed_alpha=False: alpha = 1
        ok = (h->fft_alpha = upload_named(h, need, "fft.pos_embed_alpha")) != nullptr;
    if (!ok) {
        if (!need.missing.empty()) return fail(h, DTTS_E_NOENT, "missing weight tensor '%s'", need.missing.c_str());
        return h->err.empty() ? fail(h, DTTS_E_HIP, "FFT blocks: weight upload failed") : DTTS_E_HIP;
    }
    h->fft_ready = true;
    return DTTS_OK;
}
} // namespace

int dtts_fft_blocks_forward(dtts_handle h, const float* x_in, const int32_t* lens_in, const float* pos_table, int n_pos, int B, int T,
                            float* y, dtts_stream stream) {
    if (!h) return DTTS_E_INVAL;
    if (!h->fft_ready) return fail(h, DTTS_E_STATE, "FFT block weights not finalized");
    const dtts_config& c = h->cfg;
    if (!x_in || !y || B <= 0 || T <= 0) return fail(h, DTTS_E_INVAL, "dtts_fft_blocks_forward: bad argument");
    if (c.fft_use_pos_embed && (!pos_table || n_pos <= T))
        return fail(h, DTTS_E_INVAL, "dtts_fft_blocks_forward: the stack uses positional embeddings, pos_table needs > T = %d rows (got %d)", T,
                    pos_table ? n_pos : 0);
    hipStream_t s = (hipStream_t)stream;
    const int C = c.hidden_size, F = 4 * C;
    const size_t rows = (size_t)B * T;
    HIPCHK(h->a_fft.reserve(rows * (size_t)(C + C + 3 * C + C + F + 1) * sizeof(float) + (size_t)B * sizeof(int) + (64 << 10), s));
    Arena& A = h->a_fft;
    float* x = A.alloc<float>(rows * C);
    float* hb = A.alloc<float>(rows * C);
    float* qkv = A.alloc<float>(rows * 3 * C);
    float* att = A.alloc<float>(rows * C);
    float* ff = A.alloc<float>(rows * F);
    int* lens = A.alloc<int>(B);
    int* pos = A.alloc<int>(rows);
    if (!x || !hb || !qkv || !att || !ff || !lens || !pos) return fail(h, DTTS_E_NOMEM, "FFT workspace");
    // padding_mask = x.abs().sum(-1).eq(0) unless the caller has the lengths (tts_modules.py:501)
    if (lens_in) HIPCHK(hipMemcpyAsync(lens, lens_in, sizeof(int) * B, hipMemcpyDeviceToDevice, s));
    else LAUNCH(rowcount_nonzero_launch(x_in, lens, B, T, C, s));
    // x = (x + alpha * positions) * nonpadding (:503-509)
    LAUNCH(fft_input_launch(x_in, c.fft_use_pos_embed ? pos_table : nullptr, n_pos, h->fft_alpha, lens, pos, x, B, T, C, s));
    const float kscale = (float)std::pow((double)c.fft_kernel_size, -0.5);
    for (size_t i = 0; i < h->fft.size(); ++i) {   // EncSALayer.forward (common_layers.py:649-673)
        const dtts_ctx::FftLayer& l = h->fft[i];
        LAUNCH(layernorm_launch(x, hb, l.g1, l.b1, 1e-5f, lens, 0, 0, B, T, C, s));
        ConvParams p = base_params(hb, C, B, T, T, qkv, 3 * C);
        p.out_lens = lens;   // tiles wholly past the utterance's end are skipped (left unwritten: the attention kernel never reads them)
        LAUNCH(conv1d_launch(l.qkv, p, s));
        // keys past the utterance's end are masked (-1e4 fill: their softmax weight underflows to exactly 0, as with
        // the reference's -inf); query rows past the end are zeroed by the residual epilogue below
        LAUNCH(mha_launch(qkv, att, lens, B, T, C, c.num_heads, s));
        p = base_params(att, C, B, T, T, x, C);
        set_res(p, 0, x, C);
        p.out_lens = lens;
        p.zero_masked = 1;
        LAUNCH(conv1d_launch(l.o, p, s));
        LAUNCH(layernorm_launch(x, hb, l.g2, l.b2, 1e-5f, lens, 0, 0, B, T, C, s));
        // TransformerFFNLayer (:558-581): the conv reads the LayerNorm output of padded frames too (= its bias), as the
        // reference's SAME-padded Conv1d does; (conv + bias) * k**-0.5 -> GELU
        p = base_params(hb, C, B, T, T, ff, F);
        p.out_lens = lens;   // dead tiles skipped: ffn_2 is 1x1 and its rows past the end are written as zeros whatever it reads
        p.out_mul = kscale;
        p.post_act = 3;
        LAUNCH(conv1d_launch(l.ffn1, p, s));
        p = base_params(ff, F, B, T, T, x, C);
        set_res(p, 0, x, C);
        p.out_lens = lens;
        p.zero_masked = 1;
        LAUNCH(conv1d_launch(l.ffn2, p, s));
    }
    if (c.fft_use_last_norm) LAUNCH(layernorm_launch(x, y, h->fft_g, h->fft_b, 1e-5f, lens, 0, 1, B, T, C, s));
    else HIPCHK(hipMemcpyAsync(y, x, rows * C * sizeof(float), hipMemcpyDeviceToDevice, s));
    return DTTS_OK;
}

int dtts_finalize_weights(dtts_handle h, int parts) {
    if (!h) return DTTS_E_INVAL;
    h->err.clear();
    int rc = DTTS_OK;
    if ((parts & DTTS_PART_ACOUSTIC) && !h->acoustic_ready) rc = build_acoustic(h);
    if (rc == DTTS_OK && (parts & DTTS_PART_VOCODER) && !h->vocoder_ready) rc = build_vocoder(h);
    if (rc == DTTS_OK && (parts & DTTS_PART_FFT) && !h->fft_ready) rc = build_fft(h);
    if (rc == DTTS_OK) {
        // host copies are no longer needed for finished parts
        for (auto it = h->w.begin(); it != h->w.end();) {
            const bool a = it->first.rfind("model.", 0) == 0 && h->acoustic_ready;
            const bool v = it->first.rfind("vocoder.", 0) == 0 && h->vocoder_ready;
            const bool f = it->first.rfind("fft.", 0) == 0 && h->fft_ready;
            it = (a || v || f) ? h->w.erase(it) : std::next(it);
        }
    }
    return rc;
}

int dtts_hifigan_hop(dtts_handle h) { return h ? h->hop : 0; }

int dtts_wav_to_int16(dtts_handle h, const float* wav, const int32_t* lens, int B, int T, int norm, int16_t* out, dtts_stream stream) {
    if (!h) return DTTS_E_INVAL;
    if (!h->vocoder_ready) return fail(h, DTTS_E_STATE, "vocoder weights not finalized");
    if (!wav || !out || B <= 0 || T <= 0) return fail(h, DTTS_E_INVAL, "dtts_wav_to_int16: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (!h->amax_bits || h->amax_cap < B) {   // tiny persistent scratch, grown outside the steady state
        if (h->amax_bits) {
            HIPCHK(hipDeviceSynchronize());
            (void)hipFree(h->amax_bits);
            h->amax_bits = nullptr;
        }
        HIPCHK(hipMalloc((void**)&h->amax_bits, sizeof(unsigned) * std::max(B, 256)));
        h->amax_cap = std::max(B, 256);
    }
    LAUNCH(wav_to_int16_launch(wav, lens, h->hop, B, (long long)T * h->hop, norm, h->amax_bits, out, s));
    return DTTS_OK;
}

__global__ void scale_lens_kernel(const int32_t* lens, int32_t* out, int B, int T, int n_stage, StageMult mult) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * n_stage) return;
    const int sidx = i / B, b = i % B;
    int l = lens ? lens[b] : T;
    l = l < 0 ? 0 : (l > T ? T : l);
    out[i] = l * mult.m[sidx];
}

int dtts_hifigan_forward(dtts_handle h, const float* mel, const int32_t* lens, int B, int T, float* wav, dtts_stream stream) {
    if (!h) return DTTS_E_INVAL;
    if (!h->vocoder_ready) return fail(h, DTTS_E_STATE, "vocoder weights not finalized");
    if (!mel || !wav || B <= 0 || T <= 0) return fail(h, DTTS_E_INVAL, "dtts_hifigan_forward: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const dtts_config& c = h->cfg;
    const int nup = c.n_upsamples, nk = c.n_resblock_kernels;
    Timed t_voc(h, DTTS_TIMER_STAGE_HIFIGAN, s);   // 'hifigan' (vocoders/hifigan.py:59): the generator forward
    if (c.vocoder_precision != DTTS_VOC_BF16X3) {
        // the fused kernels' persistent workgroups keep a per-utterance tile table (12 B per utterance) in LDS beside their tiles
        if (B > DTTS_MAX_VOCODER_BATCH) return fail(h, DTTS_E_INVAL, "dtts_hifigan_forward: B = %d exceeds %d utterances per call", B, DTTS_MAX_VOCODER_BATCH);
        if (c.vocoder_precision == DTTS_VOC_F16 && ((uintptr_t)mel & 15))
            return fail(h, DTTS_E_INVAL, "dtts_hifigan_forward: DTTS_VOC_F16 reads mel with 16-byte loads: the pointer must be 16-byte aligned");
        const int rc = hifigan_forward_fused(h, mel, lens, B, T, wav, s);
        // the detector's count follows the forward's last kernel into pinned host memory: whoever synchronises with this stream to read the
        // waveform can read it (dtts_vocoder_nonfinite) without another synchronisation
        if (rc == DTTS_OK && h->bad_dev && h->bad_host) HIPCHK(hipMemcpyAsync((void*)h->bad_host, h->bad_dev, sizeof(unsigned), hipMemcpyDeviceToHost, s));
        return rc;
    }
    // largest activation: stage i has T*prod(u[:i+1]) rows of C0/2^(i+1) channels
    size_t max_elems = (size_t)B * T * c.upsample_initial_channel;
    {
        long long rows = T;
        int ch = c.upsample_initial_channel;
        for (int i = 0; i < nup; ++i) {
            rows *= c.upsample_rates[i];
            ch /= 2;
            max_elems = std::max<size_t>(max_elems, (size_t)B * (size_t)rows * (size_t)ch);
        }
    }
    HIPCHK(h->a_voc.reserve(4 * (max_elems * sizeof(float) + 256) + (size_t)(nup + 2) * B * sizeof(int) + 8192, s));
    float* bufX = h->a_voc.alloc<float>(max_elems);
    float* bufR = h->a_voc.alloc<float>(max_elems);
    float* bufT = h->a_voc.alloc<float>(max_elems);
    float* bufS = h->a_voc.alloc<float>(max_elems);
    int* lensS = h->a_voc.alloc<int>((size_t)(nup + 1) * B);
    if (!bufX || !bufR || !bufT || !bufS || !lensS) return fail(h, DTTS_E_NOMEM, "vocoder workspace");
    {
        StageMult mult;
        mult.m[0] = 1;
        for (int i = 0; i < nup; ++i) mult.m[i + 1] = mult.m[i] * c.upsample_rates[i];
        hipLaunchKernelGGL(scale_lens_kernel, dim3((B * (nup + 1) + 255) / 256), dim3(256), 0, s, lens, lensS, B, T, nup + 1, mult);
    }
    const int TV = DTTS_TIMER_VOC_CONV;
    // conv_pre: mel [B,T,80] -> S [B,T,512]
    int Tcur = T, ch = c.upsample_initial_channel;
    {
        ConvParams p = base_params(mel, c.audio_num_mel_bins, B, T, T, bufS, ch);
        p.in_lens = lensS;
        p.out_lens = lensS;
        Timed tm(h, TV, s);
        LAUNCH(conv1d_launch(h->conv_pre, p, s));
    }
    for (int i = 0; i < nup; ++i) {
        const int u = c.upsample_rates[i];
        const int* lin = lensS + (size_t)i * B;
        const int* lout = lensS + (size_t)(i + 1) * B;
        ch /= 2;
        {   // x = ups[i](leaky_relu(x, 0.1)) as a polyphase convolution: [B,Tcur,2ch] -> [B,Tcur,u*ch] == [B,Tcur*u,ch]
            ConvParams p = base_params(bufS, 2 * ch, B, Tcur, Tcur, bufX, u * ch);
            p.in_lens = lin;
            p.out_lens = lin;
            p.pre_act = 1;
            p.pre_slope = 0.1f;
            Timed tm(h, TV, s);
            LAUNCH(conv1d_launch(h->ups[i], p, s));
        }
        Tcur *= u;
        for (int j = 0; j < nk; ++j) {
            const auto& c1 = h->rb1[(size_t)i * nk + j];
            const auto& c2 = h->rb2[(size_t)i * nk + j];
            for (int mth = 0; mth < 3; ++mth) {
                const float* xin = mth == 0 ? bufX : bufR;
                {   // xt = c1(leaky_relu(x))
                    ConvParams p = base_params(xin, ch, B, Tcur, Tcur, bufT, ch);
                    p.in_lens = lout;
                    p.out_lens = lout;
                    p.pre_act = 1;
                    p.pre_slope = 0.1f;
                    Timed tm(h, TV, s);
                    LAUNCH(conv1d_launch(c1[mth], p, s));
                }
                {   // x = c2(leaky_relu(xt)) + x ; the last one also folds xs (+)= x and the final / num_kernels
                    const bool last = mth == 2;
                    ConvParams p = base_params(bufT, ch, B, Tcur, Tcur, last ? bufS : bufR, ch);
                    p.in_lens = lout;
                    p.out_lens = lout;
                    p.pre_act = 1;
                    p.pre_slope = 0.1f;
                    set_res(p, 0, xin, ch);
                    if (last && j > 0) {
                        p.seg[0].res2 = bufS;
                        p.seg[0].ld_res2 = ch;
                    }
                    if (last && j == nk - 1) p.out_div = (float)nk;
                    Timed tm(h, TV, s);
                    LAUNCH(conv1d_launch(c2[mth], p, s));
                }
            }
        }
    }
    {   // x = tanh(conv_post(leaky_relu(x)))   (default slope 0.01, hifigan.py:138)
        const int* lout = lensS + (size_t)nup * B;
        ConvParams p = base_params(bufS, ch, B, Tcur, Tcur, wav, 1);
        p.in_lens = lout;
        p.out_lens = lout;
        p.zero_masked = 1;
        p.pre_act = 1;
        p.pre_slope = 0.01f;
        p.post_act = 2;
        Timed tm(h, TV, s);
        LAUNCH(conv1d_launch(h->conv_post, p, s));
    }
    return DTTS_OK;
}

static int encode_impl(dtts_handle h, const int64_t* word_tokens, const float* keys, const float* values,
                       const float* key_map, const int64_t* pinyin, const int64_t* pinyin_map, const int32_t* entry_ids,
                       const int64_t* pron_modified, const int64_t* mel2word, int T_m2w, int B, int T_w, int L_k, int P,
                       int32_t* T_mel_host, dtts_stream stream) {
    if (!h) return DTTS_E_INVAL;
    if (!h->acoustic_ready) return fail(h, DTTS_E_STATE, "acoustic weights not finalized");
    const bool tensors_ok = keys && values && key_map && pinyin && pinyin_map;
    if (!word_tokens || (!entry_ids && !tensors_ok) || !T_mel_host || B <= 0 || T_w <= 0 || L_k <= 0 || P <= 0 || L_k > 1024 ||
        P > 64)
        return fail(h, DTTS_E_INVAL, "dtts_text2mel_encode: bad argument (B=%d T_w=%d L_k=%d P=%d)", B, T_w, L_k, P);
    if (entry_ids && !h->t_entries) return fail(h, DTTS_E_STATE, "dtts_text2mel_encode_ids before dtts_dict_table_upload");
    hipStream_t s = (hipStream_t)stream;
    const dtts_config& c = h->cfg;
    const int C = c.hidden_size, D = c.gloss_dim, F = 4 * C;
    const size_t rows = (size_t)B * T_w;
    h->encoded = false;
    HIPCHK(h->a_enc.reserve(rows * (size_t)(12 * C + 3 * C + F + 2 * D + 3 * c.dur_chans + P + 8) * sizeof(float) +
                            (size_t)B * L_k * T_w * sizeof(float) + (size_t)B * (T_w + 8) * 4 * sizeof(int) + (64 << 10), s));
    Arena& A = h->a_enc;
    float* x = A.alloc<float>(rows * C);
    float* hb = A.alloc<float>(rows * C);
    float* qkv = A.alloc<float>(rows * 3 * C);
    float* att = A.alloc<float>(rows * C);
    float* ff = A.alloc<float>(rows * F);
    float* enc1 = A.alloc<float>(rows * C);
    float* q = A.alloc<float>(rows * C);
    float* qk = A.alloc<float>(rows * D);
    float* wv = A.alloc<float>(rows * D);
    float* v = A.alloc<float>(rows * C);
    float* pron = A.alloc<float>(rows * C);
    h->context = A.alloc<float>(rows * C);
    h->weo = A.alloc<float>(rows * C);
    h->dur = A.alloc<float>(rows);
    h->pron_attn = A.alloc<float>(rows * P);
    h->dict_attn = A.alloc<float>((size_t)B * L_k * T_w);
    float* d0 = A.alloc<float>(rows * c.dur_chans);
    float* d1 = A.alloc<float>(rows * c.dur_chans);
    h->lens = A.alloc<int>(B);
    int* ilens = A.alloc<int>(B);
    int* starts = A.alloc<int>((size_t)B * (T_w + 1));
    h->mel_lens = A.alloc<int>(B);
    int* pm_max = A.alloc<int>(1);
    if (!x || !hb || !qkv || !att || !ff || !enc1 || !q || !qk || !wv || !v || !pron || !h->context || !h->weo || !h->dur ||
        !h->pron_attn || !h->dict_attn || !d0 || !d1 || !h->lens || !ilens || !starts || !h->mel_lens || !pm_max)
        return fail(h, DTTS_E_NOMEM, "encoder workspace");
    h->B = B;
    h->T_w = T_w;
    h->L_k = L_k;
    h->P = P;
    // stage spans under the reference's profile_infer names (modules/dict_tts/model.py:50,86): 'encoder' = this whole call's
    // device work (dictionary encoder, duration predictor, length regulator; the gather-expand runs in decode here),
    // 'dict_encoder' = embedding + both relative-position encoders + S2PA
    Timed t_encoder(h, DTTS_TIMER_STAGE_ENCODER, s);
    Timed t_dict(h, DTTS_TIMER_STAGE_DICT_ENCODER, s);
    // A1: embedding * sqrt(hidden), lengths
    LAUNCH(embed_launch(word_tokens, h->word_emb, sqrtf((float)C), x, h->lens, B, T_w, C, c.word_size, s));
    // A2: semantic encoder
    int rc = run_encoder(h, h->sem, x, hb, qkv, att, ff, enc1, h->lens, B, T_w, s);
    if (rc) return rc;
    // A3: S2PA
    {
        ConvParams p = base_params(enc1, C, B, T_w, T_w, q, C);
        p.out_mul = (float)std::pow((double)D, -0.5);  // q * key_depth_per_head ** -0.5 (dict_encoder.py:45-46)
        LAUNCH(conv1d_launch(h->s2_q, p, s));
        const bool projected = entry_ids && h->t_projected;   // resident table of projected rows: logits = K . q, context = Wo sum_l w_l V_l
        if (!projected) {
            p = base_params(q, C, B, T_w, T_w, qk, D);
            LAUNCH(conv1d_launch(h->s2_kT, p, s));
        }
        if (entry_ids) LAUNCH(max_entry_pm_launch(entry_ids, h->t_pmmax, (long long)rows, pm_max, s));
        else LAUNCH(max_i64_launch(pinyin_map, (long long)rows * P, pm_max, s));
        S2paArgs a;
        memset(&a, 0, sizeof a);
        a.entry = entry_ids;
        a.t_off = h->t_off;
        a.t_keys = h->t_keys;
        a.t_values = h->t_values;
        a.t_key_map = h->t_key_map;
        a.t_poff = h->t_poff;
        a.t_pinyin = h->t_pinyin;
        a.t_pinyin_map = h->t_pinyin_map;
        a.qk = projected ? q : qk;
        a.keys = keys;
        a.values = values;
        a.key_map = key_map;
        a.pinyin = pinyin;
        a.pinyin_map = pinyin_map;
        a.pron_modified = pron_modified;
        a.pinyin_emb = h->pinyin_emb;
        a.pm_max = pm_max;
        a.lens = h->lens;
        a.wv = projected ? v : wv;
        a.dict_attn = h->dict_attn;
        a.pron_attn = h->pron_attn;
        a.pron = pron;
        a.B = B;
        a.T_w = T_w;
        a.L_k = L_k;
        a.P = P;
        a.D = projected ? C : D;
        a.H = C;
        a.n_pinyin = c.value_embedding_size;
        a.language_zh = c.language_zh;
        {
            Timed tm(h, DTTS_TIMER_S2PA, s);
            LAUNCH(s2pa_launch(a, s));
        }
        if (!projected) {
            p = base_params(wv, D, B, T_w, T_w, v, C);
            LAUNCH(conv1d_launch(h->s2_v, p, s));
        }
        p = base_params(v, C, B, T_w, T_w, h->context, C);
        p.out_lens = h->lens;
        p.zero_masked = 1;  // context * x_mask (dict_encoder.py:140)
        LAUNCH(conv1d_launch(h->s2_o, p, s));
        LAUNCH(add_launch(h->context, pron, x, (long long)rows * C, s));
    }
    // A4: linguistic encoder; * (word_tokens > 0) is the same prefix mask
    rc = run_encoder(h, h->lin, x, hb, qkv, att, ff, h->weo, h->lens, B, T_w, s);
    if (rc) return rc;
    t_dict.stop();
    // A5: duration predictor
    LAUNCH(rowcount_nonzero_launch(h->weo, ilens, B, T_w, C, s));
    {
        const float* in = h->weo;
        int cin = C;
        for (int i = 0; i < c.dur_predictor_layers; ++i) {
            ConvParams p = base_params(in, cin, B, T_w, T_w, d0, c.dur_chans);
            p.post_act = 1;
            LAUNCH(conv1d_launch(h->dur_conv[i], p, s));
            LAUNCH(layernorm_launch(d0, d1, h->dur_g[i], h->dur_b[i], 1e-5f, ilens, 0, 1, B, T_w, c.dur_chans, s));
            in = d1;  // next conv reads d1 and writes d0 again
            cin = c.dur_chans;
        }
        LAUNCH(dur_head_launch(in, h->dur_w, h->dur_bias, ilens, h->dur, B, T_w, c.dur_chans, s));
    }
    // A6/A7: durations -> mel2word
    int T_raw = 0;
    if (!mel2word) {
        LAUNCH(durations_launch(h->dur, ilens, starts, h->mel_lens, B, T_w, s));
        std::vector<int> tot(B);
        int pm_host = 0;
        HIPCHK(hipMemcpyAsync(tot.data(), h->mel_lens, sizeof(int) * B, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(&pm_host, pm_max, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));  // the one host sync of the path: T_mel sizes every later buffer
        if (pm_host > DTTS_MAX_SENSES)   // the S2PA kernel keeps DTTS_MAX_SENSES sense slots; larger indices would silently get weight 0
            return fail(h, DTTS_E_INVAL, "pinyin_map holds sense index %d; at most %d senses per word are supported", pm_host, DTTS_MAX_SENSES);
        for (int b = 0; b < B; ++b) T_raw = std::max(T_raw, tot[b]);
    } else {
        if (T_m2w <= 0) return fail(h, DTTS_E_INVAL, "mel2word given with T_m2w=%d", T_m2w);
        T_raw = T_m2w;
    }
    const int fm = c.frames_multiple;
    const int T_mel = (T_raw % fm) ? T_raw + fm - T_raw % fm : T_raw;
    const int T4 = T_mel / 4;
    const size_t mrows = (size_t)B * T_mel, qrows = (size_t)B * T4;
    const int Hd = c.fvae_enc_dec_hidden, Hf = c.prior_glow_hidden;
    HIPCHK(h->a_dec.reserve(mrows * (size_t)(C + 1 + 2 + 2 * Hd * c.fvae_dec_n_layers + 3 * Hd + 8) * sizeof(float) +
                            qrows * (size_t)(C + 2 * c.latent_size + 2 * Hf * c.prior_glow_n_layers * (1 + c.prior_glow_n_blocks) + 3 * Hf + 16) * sizeof(float) +
                            (size_t)B * T_w * 2 * Hd * c.fvae_dec_n_layers * sizeof(float) + (64 << 10), s));
    h->m2w = h->a_dec.alloc<int64_t>(mrows);
    h->x_mask = h->a_dec.alloc<float>(mrows);
    if (!h->m2w || !h->x_mask) return fail(h, DTTS_E_NOMEM, "decoder workspace");
    if (!mel2word) LAUNCH(mel2word_fill_launch(starts, h->mel_lens, ilens, h->m2w, B, T_w, T_raw, T_mel, s));
    else LAUNCH(mel2word_copy_launch(mel2word, h->m2w, h->mel_lens, B, T_m2w, T_mel, s));
    h->T_mel = T_mel;
    *T_mel_host = T_mel;
    h->encoded = true;
    return DTTS_OK;
}

int dtts_text2mel_encode(dtts_handle h, const int64_t* word_tokens, const float* keys, const float* values,
                         const float* key_map, const int64_t* pinyin, const int64_t* pinyin_map,
                         const int64_t* pron_modified, const int64_t* mel2word, int T_m2w, int B, int T_w, int L_k, int P,
                         int32_t* T_mel_host, dtts_stream stream) {
    if (h && !(keys && values && key_map && pinyin && pinyin_map))
        return fail(h, DTTS_E_INVAL, "dtts_text2mel_encode: null dictionary tensor");
    return encode_impl(h, word_tokens, keys, values, key_map, pinyin, pinyin_map, nullptr, pron_modified, mel2word, T_m2w, B, T_w,
                       L_k, P, T_mel_host, stream);
}

int dtts_text2mel_encode_ids(dtts_handle h, const int64_t* word_tokens, const int32_t* entry_ids, const int64_t* pron_modified,
                             const int64_t* mel2word, int T_m2w, int B, int T_w, int L_k, int P, int32_t* T_mel_host,
                             dtts_stream stream) {
    if (h && !entry_ids) return fail(h, DTTS_E_INVAL, "dtts_text2mel_encode_ids: null entry ids");
    return encode_impl(h, word_tokens, nullptr, nullptr, nullptr, nullptr, nullptr, entry_ids, pron_modified, mel2word, T_m2w, B,
                       T_w, L_k, P, T_mel_host, stream);
}

int dtts_dict_table_upload(dtts_handle h, int n_entries, const int32_t* tok_off, const float* keys, const float* values,
                           const float* key_map, const int32_t* pin_off, const int64_t* pinyin, const int64_t* pinyin_map) {
    if (!h || n_entries <= 0 || !tok_off || !keys || !key_map || !pin_off || !pinyin || !pinyin_map)
        return fail(h, DTTS_E_INVAL, "dtts_dict_table_upload: bad argument");
    const int D = h->cfg.gloss_dim;
    const size_t nL = (size_t)tok_off[n_entries], nP = (size_t)pin_off[n_entries];
    for (int e = 0; e < n_entries; ++e)
        if (tok_off[e + 1] < tok_off[e] || pin_off[e + 1] < pin_off[e])
            return fail(h, DTTS_E_INVAL, "dtts_dict_table_upload: offsets must be non-decreasing (entry %d)", e);
    std::vector<int> pmmax(n_entries, 0);
    for (int e = 0; e < n_entries; ++e) {
        for (int p = pin_off[e]; p < pin_off[e + 1]; ++p) pmmax[e] = std::max(pmmax[e], (int)pinyin_map[p]);
        float km = 0.f;
        for (int l = tok_off[e]; l < tok_off[e + 1]; ++l) km = std::max(km, key_map[l]);
        if (pmmax[e] > DTTS_MAX_SENSES || km > (float)DTTS_MAX_SENSES)
            return fail(h, DTTS_E_INVAL, "dtts_dict_table_upload: entry %d has sense index %d; at most %d senses per word are supported",
                        e, std::max(pmmax[e], (int)km), DTTS_MAX_SENSES);
    }
    // ---- build the NEW table completely before touching the one in use: a failed re-upload leaves the previous table working
    int dev_cur = -1;
    (void)hipGetDevice(&dev_cur);
    if (dev_cur != h->device)
        return fail(h, DTTS_E_STATE, "dtts_dict_table_upload: the current HIP device is %d, the context was created on device %d", dev_cur, h->device);
    std::vector<void*> fresh;   // the new table's allocations (released again if anything below fails)
    const char* what = nullptr;
    hipError_t herr = hipSuccess;
    auto up = [&](const void* src, size_t bytes) -> void* {
        void* d = dev_alloc(h, bytes);
        if (!d) {
            what = "device allocation";
            return nullptr;
        }
        fresh.push_back(d);
        if (bytes && (herr = hipMemcpy(d, src, bytes, hipMemcpyHostToDevice)) != hipSuccess) {
            what = "host-to-device copy";
            return nullptr;
        }
        return d;
    };
    // SURVEY 8d "resident-table path": the table holds the PROJECTED rows K = k_transform(key), V = v_transform(value)
    // (dict_encoder.py:36-39: the reference projects every gloss row of every batch; here once, at upload) — 2 x hidden_size floats per
    // row instead of 768 (+ 768), and the logit becomes k . q in the reference's own association order.  tune bit 64 keeps the raw
    // rows (round 2's table: the re-associated kernel of the tensor API reads them).
    const bool projected = !DTTS_TUNE(h, 64);
    if (projected && !h->acoustic_ready)
        return fail(h, DTTS_E_STATE, "dtts_dict_table_upload: the acoustic weights must be finalized first (the table stores k_transform / v_transform projections)");
    if (projected && nL > (size_t)INT_MAX / 2) return fail(h, DTTS_E_INVAL, "dtts_dict_table_upload: %zu gloss rows", nL);
    int* n_off = (int*)up(tok_off, sizeof(int) * (n_entries + 1));
    int* n_poff = (int*)up(pin_off, sizeof(int) * (n_entries + 1));
    int* n_pmmax = (int*)up(pmmax.data(), sizeof(int) * n_entries);
    float *n_keys = nullptr, *n_values = nullptr;
    if (projected) {
        const int C = h->cfg.hidden_size;
        float* raw = nullptr;
        hipStream_t ps = nullptr;   // the projection runs on its own stream, on the device that is current now (= the context's: its weights live there)
        if (!what && (herr = hipMalloc((void**)&raw, std::max<size_t>(nL * D * sizeof(float), 16))) != hipSuccess) what = "staging buffer allocation";
        if (!what && (herr = hipStreamCreate(&ps)) != hipSuccess) what = "hipStreamCreate";
        auto proj = [&](const float* src, const PackedConv& L) -> float* {   // [nL][D] host rows -> [nL][C] device rows
            if (what) return nullptr;
            float* out = (float*)dev_alloc(h, nL * C * sizeof(float));
            if (!out) {
                what = "device allocation";
                return nullptr;
            }
            fresh.push_back(out);
            if (nL == 0) return out;
            if ((herr = hipMemcpyAsync(raw, src, nL * D * sizeof(float), hipMemcpyHostToDevice, ps)) != hipSuccess) {
                what = "host-to-device copy";
                return nullptr;
            }
            ConvParams p = base_params(raw, D, 1, (int)nL, (int)nL, out, C);
            if ((herr = conv1d_launch(L, p, ps)) != hipSuccess) {
                what = "projection kernel launch";
                return nullptr;
            }
            if ((herr = hipStreamSynchronize(ps)) != hipSuccess) {
                what = "projection kernel";
                return nullptr;
            }
            return out;
        };
        n_keys = proj(keys, h->s2_k);
        n_values = proj(values ? values : keys, h->s2_v);
        if (ps) (void)hipStreamDestroy(ps);
        if (raw) (void)hipFree(raw);
    } else {
        n_keys = (float*)up(keys, nL * D * sizeof(float));
        n_values = values ? (float*)up(values, nL * D * sizeof(float)) : n_keys;  // the reference stores key == value
    }
    float* n_key_map = (float*)up(key_map, nL * sizeof(float));
    int64_t* n_pinyin = (int64_t*)up(pinyin, nP * sizeof(int64_t));
    int64_t* n_pinyin_map = (int64_t*)up(pinyin_map, nP * sizeof(int64_t));
    if (what || !n_off || !n_poff || !n_pmmax || !n_keys || !n_values || !n_key_map || !n_pinyin || !n_pinyin_map) {
        for (void* q : fresh) dev_free(h, q);
        return fail(h, what && strstr(what, "allocation") ? DTTS_E_NOMEM : DTTS_E_HIP, "dtts_dict_table_upload: %s failed (%s)%s",
                    what ? what : "device allocation", hipGetErrorString(herr), h->t_entries ? "; the previous table stays in use" : "");
    }
    if (h->t_entries) {   // a second upload replaces the table: release the previous one (nothing may still be using it)
        if ((herr = hipDeviceSynchronize()) != hipSuccess) {   // (the new table is released again; the previous one stays in use)
            for (void* q : fresh) dev_free(h, q);
            return fail(h, DTTS_E_HIP, "dtts_dict_table_upload: hipDeviceSynchronize failed (%s); the previous table stays in use", hipGetErrorString(herr));
        }
        void* old[] = {h->t_off, h->t_poff, h->t_pmmax, h->t_keys, h->t_values != h->t_keys ? h->t_values : nullptr, h->t_key_map, h->t_pinyin, h->t_pinyin_map};
        for (void* q : old) dev_free(h, q);
    }
    h->t_off = n_off;
    h->t_poff = n_poff;
    h->t_pmmax = n_pmmax;
    h->t_keys = n_keys;
    h->t_values = n_values;
    h->t_key_map = n_key_map;
    h->t_pinyin = n_pinyin;
    h->t_pinyin_map = n_pinyin_map;
    h->t_projected = projected;
    h->t_entries = n_entries;
    return DTTS_OK;
}

// z_p: [B][latent][z_ld] (z_ld >= T_mel/4; 0 = exactly T_mel/4) or null = drawn on the device; mel_out: [B][mel_cap][n_mel]
// (mel_cap >= T_mel; 0 = exactly T_mel), rows >= T_mel are left untouched
static int decode_impl(dtts_handle h, const float* z_p, int z_ld, float* mel_out, int mel_cap, dtts_stream stream) {
    if (!h) return DTTS_E_INVAL;
    if (!h->encoded) return fail(h, DTTS_E_STATE, "dtts_text2mel_decode called before a successful dtts_text2mel_encode");
    if (!mel_out) return fail(h, DTTS_E_INVAL, "dtts_text2mel_decode: null argument");
    hipStream_t s = (hipStream_t)stream;
    const dtts_config& c = h->cfg;
    const int B = h->B, T = h->T_mel, T4 = T / 4, C = c.hidden_size, Z = c.latent_size;
    const int Hd = c.fvae_enc_dec_hidden, Hf = c.prior_glow_hidden;
    const size_t mrows = (size_t)B * T, qrows = (size_t)B * T4;
    Timed t_fvae(h, DTTS_TIMER_STAGE_FVAE, s);   // 'fvae' (model.py:57) + the gather-expand of run_text_encoder
    Arena& A = h->a_dec;
    // (m2w and x_mask were allocated first by encode; everything below is re-allocated after them on every call)
    A.rewind();
    (void)A.alloc<int64_t>(mrows);
    (void)A.alloc<float>(mrows);
    float* g = A.alloc<float>(mrows * C);
    float* gs = A.alloc<float>(qrows * C);
    float* z = A.alloc<float>(qrows * Z);
    float* fcond = A.alloc<float>(qrows * 2 * Hf * c.prior_glow_n_layers);
    float* fh = A.alloc<float>(qrows * Hf);
    float* facts = A.alloc<float>(qrows * Hf);
    float* fout = A.alloc<float>(qrows * Hf);
    float* dx = A.alloc<float>(mrows * Hd);
    float* dacts = A.alloc<float>(mrows * Hd);
    float* dout = A.alloc<float>(mrows * Hd);
    if (!g || !gs || !z || !fcond || !fh || !facts || !fout || !dx || !dacts || !dout)
        return fail(h, DTTS_E_NOMEM, "decoder workspace");
    // A7: gather-expand (x * tgt_nonpadding is implied: padded frames gather the zero row)
    LAUNCH(expand_launch(h->weo, h->m2w, g, h->x_mask, B, h->T_w, T, C, s));
    // A8: g_sqz = Conv1d(k=8, s=4, p=2)(g)
    ConvParams p = base_params(g, C, B, T, T4, gs, C);
    if (h->g_pre_poly.w_hi) {
        VConvParams v = vparams_x3(h->g_pre_poly, g, 4 * C, 1.f, nullptr, B, T4);
        v.in_half = 1;
        v.yf = gs;
        v.ldyf = C;
        LAUNCH(vconv_launch(v, s));
    } else {
        LAUNCH(conv1d_launch(h->g_pre, p, s));
    }
    if (z_p) {
        if (z_ld && z_ld < T4) return fail(h, DTTS_E_INVAL, "prior sample holds %d steps per row, T_mel/4 = %d", z_ld, T4);
        LAUNCH(transpose_cf_to_cl_launch(z_p, z, B, Z, T4, s, z_ld));
    } else {
        LAUNCH(normal_fill_launch(z, (long long)qrows * Z, h->noise_seed + ++h->noise_counter, s));   // z_p ~ N(0,1) (fvae_semantics.py:110-111)
    }
    // A9: prior flow, reverse
    if (h->fs_w) {   // every block in one kernel (flowstack.hip); the conditioning of all blocks by one convolution
        const int n_c = h->fs_cond.C_out;
        float* cond_all = A.alloc<float>(qrows * n_c);
        float* z2 = A.alloc<float>(qrows * Z);
        if (!cond_all || !z2) return fail(h, DTTS_E_NOMEM, "decoder workspace");
        if (h->fs_cond.engine == ENG_BF16X3) {
            VConvParams v = vparams_x3(h->fs_cond, gs, C, 1.f, nullptr, B, T4);
            v.yf = cond_all;
            v.ldyf = n_c;
            LAUNCH(vconv_launch(v, s));
        } else {
            p = base_params(gs, C, B, T4, T4, cond_all, n_c);
            LAUNCH(conv1d_launch(h->fs_cond, p, s));
        }
        FlowStackParams fp;
        memset(&fp, 0, sizeof fp);
        fp.z_in = z;
        fp.z_out = z2;
        fp.cond = cond_all;
        fp.ld_cond = n_c;
        fp.w = h->fs_w;
        fp.B = B;
        fp.T4 = T4;
        fp.Z = Z;
        fp.n_flows = (int)h->flows.size();
        fp.layers = c.prior_glow_n_layers;
        fp.x3 = c.decoder_fp32 ? 0 : 1;   // split-bf16 like the decoder WaveNet unless the exact-fp32 decoder was asked for
        for (size_t i = 0; i < h->flows.size(); ++i) {
            fp.in_coff[i] = h->flows[i].in_coff;
            fp.out_coff[i] = h->flows[i].out_coff;
        }
        LAUNCH(flowstack_launch(fp, s));
        z = z2;
    } else
    for (const Flow& fl : h->flows) {
        p = base_params(z, Z, B, T4, T4, fh, Hf);
        p.x_coff = fl.in_coff;
        LAUNCH(conv1d_launch(fl.pre, p, s));
        int rc = run_wn(h, fl.wn, fh, gs, C, fcond, facts, fout, B, T4, s);
        if (rc) return rc;
        p = base_params(fout, Hf, B, T4, T4, z, Z);
        p.seg[0].coff = fl.out_coff;
        set_res(p, 0, z, Z);
        p.seg[0].coff_res = fl.out_coff;
        LAUNCH(conv1d_launch(fl.post, p, s));
    }
    // A10: decoder
    p = base_params(z, Z, B, T4, T4, dx, 4 * Hd);  // ConvTranspose1d(k=4,s=4): [B,T4,16] -> [B,T4,4*Hd] == [B,T,Hd]
    LAUNCH(conv1d_launch(h->dec_pre, p, s));
    // The decoder's conditioning is a 1x1 convolution of g, and g[b,t] is just word row mel2word[b,t] of the encoder
    // output (or the zero row): the convolution is applied to the B*T_w word rows (33x fewer than the B*T frames) and its
    // output gathered by mel2word; frames with mel2word == 0 get conv(0) = bias.  Bit-identical: every output row of this
    // kernel depends only on its own input row, summed in the same order whatever the tile shape.
    const int CW = 2 * Hd * c.fvae_dec_n_layers;
    float* cond_w = A.alloc<float>(((size_t)B * h->T_w + 1) * CW);   // row 0: conv(0) = the bias, rows 1..: the B*T_w word rows
    if (!cond_w) return fail(h, DTTS_E_NOMEM, "decoder workspace");
    if (hipMemcpyAsync(cond_w, h->dec_wn.cond.bias, (size_t)CW * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess)
        return fail(h, DTTS_E_HIP, "decoder conditioning bias row");
    p = base_params(h->weo, C, B, h->T_w, h->T_w, cond_w + CW, CW);
    LAUNCH(conv1d_launch(h->dec_wn.cond, p, s));
    int rc;
    if (!h->dec_wn.in.empty() && h->dec_wn.in[0].engine == ENG_BF16X3) {
        // split-operand layers: every layer's epilogue gathers its conditioning row by mel2word from the word-level tensor (L2-resident,
        // B*T_w rows) — the [B*T, 2*Hd*layers] expansion (221 MB written and read back at B=60) never exists
        rc = run_wn(h, h->dec_wn, dx, nullptr, C, cond_w, dacts, dout, B, T, s, h->m2w, h->T_w);
    } else {
        float* dcond = A.alloc<float>(mrows * CW);
        if (!dcond) return fail(h, DTTS_E_NOMEM, "decoder workspace");
        LAUNCH(expand_launch(cond_w + CW, h->m2w, dcond, nullptr, B, h->T_w, T, CW, s, h->dec_wn.cond.bias));
        rc = run_wn(h, h->dec_wn, dx, nullptr, C, dcond, dacts, dout, B, T, s);
    }
    if (rc) return rc;
    p = base_params(dout, Hd, B, T, T, mel_out, c.audio_num_mel_bins);
    if (mel_cap) {
        if (mel_cap < T) return fail(h, DTTS_E_INVAL, "mel_out holds %d frames per utterance, T_mel = %d", mel_cap, T);
        p.y_bstride_rows = mel_cap;   // only this layer's output lives in the caller's capacity layout
    }
    LAUNCH(conv1d_launch(h->dec_out, p, s));
    return DTTS_OK;
}

int dtts_text2mel_decode(dtts_handle h, const float* z_p, float* mel_out, dtts_stream stream) {
    return decode_impl(h, z_p, 0, mel_out, 0, stream);   // z_p == NULL: the prior sample is drawn on the device
}

// ---- the single-call forms and names of SURVEY.md 8(b)
int dtts_load_weights(dtts_handle h, const char* name, const void* host_ptr, const int64_t* shape, int ndim, int dtype) {
    return dtts_load_weight(h, name, host_ptr, shape, ndim, dtype);
}

int dtts_text2mel_plan(dtts_handle h, const int64_t* word_tokens, const float* keys, const float* values, const float* key_map,
                       const int64_t* pinyin, const int64_t* pinyin_map, const int64_t* pron_modified, const int64_t* mel2word,
                       int T_m2w, int B, int T_w, int L_k, int P, int32_t* T_mel_host, dtts_stream stream) {
    return dtts_text2mel_encode(h, word_tokens, keys, values, key_map, pinyin, pinyin_map, pron_modified, mel2word, T_m2w, B, T_w, L_k,
                                P, T_mel_host, stream);
}

static int forward_tail(dtts_handle h, const float* z_p, int z_cap, float* mel_out, int mel_cap, int T_mel, int64_t* T_mel_out,
                        float* pron_attn, float* dur, dtts_stream stream) {
    if (T_mel_out) *T_mel_out = T_mel;
    if (T_mel > mel_cap) return fail(h, DTTS_E_INVAL, "dtts_text2mel_forward: %d frames exceed the capacity %d of mel_out", T_mel, mel_cap);
    int rc = decode_impl(h, z_p, z_p ? z_cap : 0, mel_out, mel_cap, stream);
    if (rc == DTTS_OK && pron_attn) rc = dtts_text2mel_fetch(h, DTTS_OUT_PRON_ATTN, pron_attn, stream);
    if (rc == DTTS_OK && dur) rc = dtts_text2mel_fetch(h, DTTS_OUT_DUR, dur, stream);
    return rc;
}

int dtts_text2mel_forward(dtts_handle h, const int64_t* word_tokens, const float* keys, const float* values, const float* key_map,
                          const int64_t* pinyin, const int64_t* pinyin_map, const int64_t* pron_modified, const int64_t* mel2word,
                          int T_m2w, const float* z_p, int z_cap, int B, int T_w, int L_k, int P, float* mel_out, int mel_cap,
                          int64_t* T_mel_out, float* pron_attn, float* dur, dtts_stream stream) {
    if (h && (!mel_out || mel_cap <= 0)) return fail(h, DTTS_E_INVAL, "dtts_text2mel_forward: bad argument");
    int32_t T_mel = 0;
    const int rc = dtts_text2mel_encode(h, word_tokens, keys, values, key_map, pinyin, pinyin_map, pron_modified, mel2word, T_m2w, B,
                                        T_w, L_k, P, &T_mel, stream);
    return rc ? rc : forward_tail(h, z_p, z_cap, mel_out, mel_cap, T_mel, T_mel_out, pron_attn, dur, stream);
}

int dtts_text2mel_forward_ids(dtts_handle h, const int64_t* word_tokens, const int32_t* entry_ids, const int64_t* pron_modified,
                              const int64_t* mel2word, int T_m2w, const float* z_p, int z_cap, int B, int T_w, int L_k, int P,
                              float* mel_out, int mel_cap, int64_t* T_mel_out, float* pron_attn, float* dur, dtts_stream stream) {
    if (h && (!mel_out || mel_cap <= 0)) return fail(h, DTTS_E_INVAL, "dtts_text2mel_forward_ids: bad argument");
    int32_t T_mel = 0;
    const int rc = dtts_text2mel_encode_ids(h, word_tokens, entry_ids, pron_modified, mel2word, T_m2w, B, T_w, L_k, P, &T_mel, stream);
    return rc ? rc : forward_tail(h, z_p, z_cap, mel_out, mel_cap, T_mel, T_mel_out, pron_attn, dur, stream);
}

int dtts_text2mel_fetch(dtts_handle h, int what, void* dst, dtts_stream stream) {
    if (!h || !dst) return DTTS_E_INVAL;
    if (!h->encoded) return fail(h, DTTS_E_STATE, "dtts_text2mel_fetch before encode");
    hipStream_t s = (hipStream_t)stream;
    const size_t rows = (size_t)h->B * h->T_w, mrows = (size_t)h->B * h->T_mel;
    const void* src = nullptr;
    size_t bytes = 0;
    switch (what) {
        case DTTS_OUT_PRON_ATTN: src = h->pron_attn; bytes = rows * h->P * 4; break;
        case DTTS_OUT_DUR: src = h->dur; bytes = rows * 4; break;
        case DTTS_OUT_MEL2WORD: src = h->m2w; bytes = mrows * 8; break;
        case DTTS_OUT_DICT_ATTN:
            // kept as [B][T_w][L_k] (every word's weights one contiguous row, written coalesced by s2pa_kernel); the reference returns the
            // transposed view weights.permute(0, 1, 3, 2) = [B, 1, L_k, T_w] (dict_encoder.py:66): produced here, when somebody asks for it
            LAUNCH(transpose_cf_to_cl_launch(h->dict_attn, (float*)dst, h->B, h->T_w, h->L_k, s));
            return DTTS_OK;
        case DTTS_OUT_WORD_ENCODER_OUT: src = h->weo; bytes = rows * h->cfg.hidden_size * 4; break;
        case DTTS_OUT_X_MASK: src = h->x_mask; bytes = mrows * 4; break;
        case DTTS_OUT_CONTEXT: src = h->context; bytes = rows * h->cfg.hidden_size * 4; break;
        case DTTS_OUT_MEL_LENS: src = h->mel_lens; bytes = (size_t)h->B * 4; break;
        default: return fail(h, DTTS_E_INVAL, "dtts_text2mel_fetch: unknown item %d", what);
    }
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s));
    return DTTS_OK;
}

int dtts_length_regulate(dtts_handle h, const float* dur, const int32_t* ilens, int B, int T_w, int64_t* mel2word, int cap,
                         int32_t* T_max_host, dtts_stream stream) {
    if (!h || !dur || !ilens || !mel2word || !T_max_host || B <= 0 || T_w <= 0 || cap <= 0)
        return fail(h, DTTS_E_INVAL, "dtts_length_regulate: bad argument");
    hipStream_t s = (hipStream_t)stream;
    int *starts = nullptr, *total = nullptr;
    HIPCHK(hipMalloc((void**)&starts, sizeof(int) * ((size_t)B * (T_w + 1) + B)));
    total = starts + (size_t)B * (T_w + 1);
    std::vector<int> tot(B);
    hipError_t e = durations_launch(dur, ilens, starts, total, B, T_w, s);
    if (e == hipSuccess) e = hipMemcpyAsync(tot.data(), total, sizeof(int) * B, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    int T_raw = 0;
    for (int b = 0; b < B; ++b) T_raw = std::max(T_raw, tot[b]);
    *T_max_host = T_raw;
    int rc = DTTS_OK;
    if (e != hipSuccess) rc = fail(h, DTTS_E_HIP, "dtts_length_regulate: %s", hipGetErrorString(e));
    else if (T_raw > cap) rc = fail(h, DTTS_E_INVAL, "dtts_length_regulate: %d frames exceed the capacity %d", T_raw, cap);
    else {
        e = mel2word_fill_launch(starts, total, ilens, mel2word, B, T_w, cap, cap, s);  // columns >= total[b] are zero
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) rc = fail(h, DTTS_E_HIP, "dtts_length_regulate: %s", hipGetErrorString(e));
    }
    (void)hipFree(starts);
    return rc;
}

int dtts_set_noise_seed(dtts_handle h, uint64_t seed) {
    if (!h) return DTTS_E_INVAL;
    h->noise_seed = seed;
    h->noise_counter = 0x5EEDull;
    return DTTS_OK;
}

int dtts_vocoder_range_guard(dtts_handle h, int enable) {
    if (!h) return DTTS_E_INVAL;
    if (enable && h->cfg.vocoder_precision != DTTS_VOC_F16) return fail(h, DTTS_E_INVAL, "the range guard exists for DTTS_VOC_F16 only (the other modes have fp32's exponent range)");
    h->guard_on = enable != 0;
    return DTTS_OK;
}

int dtts_vocoder_clamped(dtts_handle h, int64_t* count, int reset, dtts_stream stream) {
    if (!h || !count) return DTTS_E_INVAL;
    if (!h->ovf_dev) return fail(h, DTTS_E_STATE, "vocoder weights not finalized");
    unsigned long long v = 0;
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipMemcpyAsync(&v, h->ovf_dev, sizeof v, hipMemcpyDeviceToHost, s));
    if (reset) HIPCHK(hipMemsetAsync(h->ovf_dev, 0, sizeof v, s));
    HIPCHK(hipStreamSynchronize(s));
    *count = (int64_t)v;
    return DTTS_OK;
}

int dtts_vocoder_nonfinite(dtts_handle h, int64_t* count) {
    if (!h || !count) return DTTS_E_INVAL;
    if (!h->vocoder_ready) return fail(h, DTTS_E_STATE, "vocoder weights not finalized");
    *count = h->bad_host ? (int64_t)*h->bad_host : 0;
    return DTTS_OK;
}

int dtts_vocoder_fp16_bound(dtts_handle h, float mel_abs_max, double* worst_case, double* rms_estimate) {
    if (!h || !(mel_abs_max >= 0.f)) return DTTS_E_INVAL;
    if (!h->vocoder_ready) return fail(h, DTTS_E_STATE, "vocoder weights not finalized");
    const bool f16 = h->cfg.vocoder_precision == DTTS_VOC_F16;
    if (worst_case) *worst_case = f16 ? h->wc_const + h->wc_lin * (double)mel_abs_max : 0.0;
    if (rms_estimate) *rms_estimate = f16 ? std::sqrt(h->est_const * h->est_const + h->est_lin * h->est_lin * (double)mel_abs_max * mel_abs_max) : 0.0;
    return DTTS_OK;
}

int dtts_timer_enable(dtts_handle h, int which) {
    if (!h || which < 1 || which >= DTTS_TIMER_COUNT) return DTTS_E_INVAL;
    h->timers[which].enabled = true;
    return DTTS_OK;
}

static void timer_collect(TimerSlot& t) {
    for (size_t i = 0; i + 1 < t.used; i += 2) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, t.pool[i], t.pool[i + 1]) == hipSuccess) t.ms_done += ms;
    }
    t.used = 0;
}

int dtts_timer_read(dtts_handle h, int which, double* ms_total, int64_t* launches) {
    if (!h || which < 1 || which >= DTTS_TIMER_COUNT) return DTTS_E_INVAL;
    HIPCHK(hipDeviceSynchronize());
    TimerSlot& t = h->timers[which];
    timer_collect(t);
    if (ms_total) *ms_total = t.ms_done;
    if (launches) *launches = t.launches;
    return DTTS_OK;
}

int dtts_timer_reset(dtts_handle h) {
    if (!h) return DTTS_E_INVAL;
    HIPCHK(hipDeviceSynchronize());
    for (auto& t : h->timers) {
        t.used = 0;
        t.ms_done = 0;
        t.launches = 0;
    }
    return DTTS_OK;
}

// ---- memory-safety mode (dtts_config.debug_redzone): verify every red zone of the workspaces and the weight packs
namespace {
struct RzZone { const unsigned char* p; unsigned n; unsigned id; };
__global__ void redzone_check_kernel(const RzZone* z, int nz, unsigned long long* out) {   // out[0] = damaged bytes, out[1] = lowest damaged zone id
    const RzZone q = z[blockIdx.x];
    unsigned bad = 0;
    for (unsigned i = threadIdx.x; i < q.n; i += blockDim.x) bad += q.p[i] != 0xFF ? 1u : 0u;
    if (bad) {
        atomicAdd(out, (unsigned long long)bad);
        atomicMin(out + 1, (unsigned long long)q.id);
    }
}
} // namespace

// harness self-test: damage ONE red-zone byte (the first byte after the first buffer of the first workspace in use, or after the first
// weight pack) the way an off-by-one store of a kernel would, so that a test can show dtts_debug_check notices
int dtts_debug_poke(dtts_handle h, dtts_stream stream) {
    if (!h) return DTTS_E_INVAL;
    if (!h->debug_rz) return fail(h, DTTS_E_STATE, "dtts_debug_poke: the context was not created with dtts_config.debug_redzone = 1");
    char* target = nullptr;
    for (Arena* a : {&h->a_voc, &h->a_enc, &h->a_dec, &h->a_fft})
        if (!target && a->base && !a->bufs.empty()) target = a->base + a->bufs[0].start + a->bufs[0].bytes;
    if (!target && !h->rz_static.empty()) target = h->rz_static[0].p + h->rz_static[0].bytes;
    if (!target) return fail(h, DTTS_E_STATE, "dtts_debug_poke: nothing allocated yet");
    HIPCHK(hipMemsetAsync(target, 0, 1, (hipStream_t)stream));
    return DTTS_OK;
}

int dtts_debug_check(dtts_handle h, int64_t* damaged_bytes, dtts_stream stream) {
    if (!h || !damaged_bytes) return DTTS_E_INVAL;
    *damaged_bytes = -1;
    if (!h->debug_rz) return fail(h, DTTS_E_STATE, "dtts_debug_check: the context was not created with dtts_config.debug_redzone = 1");
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipStreamSynchronize(s));
    std::vector<RzZone> zones;
    std::vector<std::string> names;
    auto zone = [&](const char* p0, size_t n, const std::string& name) {
        if (!n) return;
        zones.push_back({(const unsigned char*)p0, (unsigned)n, (unsigned)names.size()});
        names.push_back(name);
    };
    const std::pair<const char*, Arena*> arenas[] = {{"encode workspace", &h->a_enc}, {"decode workspace", &h->a_dec}, {"vocoder workspace", &h->a_voc}, {"fft workspace", &h->a_fft}};
    for (const auto& a : arenas) {
        const auto& bufs = a.second->bufs;
        const char* base = a.second->base;
        for (size_t i = 0; i < bufs.size(); ++i) {   // layout: [RZ][buffer 0][slack to 256 B][RZ][buffer 1]...[RZ .. up to the arena's end]
            const std::string me = std::string(a.first) + " buffer #" + std::to_string(i) + " (" + std::to_string(bufs[i].bytes) + " B)";
            const size_t end = bufs[i].start + bufs[i].bytes, padded_end = bufs[i].start + ((bufs[i].bytes + 255) & ~(size_t)255);
            zone(base + bufs[i].start - RZ, RZ, i ? "AFTER " + std::string(a.first) + " buffer #" + std::to_string(i - 1) + " / BEFORE " + me : "BEFORE " + me);
            zone(base + end, padded_end - end, "AFTER " + me + " (alignment slack)");
            if (i + 1 == bufs.size()) zone(base + padded_end, std::min(RZ, a.second->cap - padded_end), "AFTER " + me);
        }
    }
    for (size_t i = 0; i < h->rz_static.size(); ++i) {
        const auto& b = h->rz_static[i];
        const size_t padded = (b.bytes + 255) & ~(size_t)255;
        const std::string me = "weight pack / table #" + std::to_string(i) + " (" + std::to_string(b.bytes) + " B)";
        zone(b.p - RZ, RZ, "BEFORE " + me);
        zone(b.p + b.bytes, padded + RZ - b.bytes, "AFTER " + me);
    }
    if (zones.empty()) {
        *damaged_bytes = 0;
        return DTTS_OK;
    }
    RzZone* dz = nullptr;
    unsigned long long* dout = nullptr;
    unsigned long long hout[2] = {0ull, ~0ull};
    HIPCHK(hipMalloc((void**)&dz, zones.size() * sizeof(RzZone)));
    hipError_t e = hipMalloc((void**)&dout, sizeof hout);
    if (e == hipSuccess) e = hipMemcpy(dz, zones.data(), zones.size() * sizeof(RzZone), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dout, hout, sizeof hout, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(redzone_check_kernel, dim3((unsigned)zones.size()), dim3(256), 0, s, dz, (int)zones.size(), dout);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess) e = hipMemcpy(hout, dout, sizeof hout, hipMemcpyDeviceToHost);
    (void)hipFree(dz);
    if (dout) (void)hipFree(dout);
    if (e != hipSuccess) return fail(h, DTTS_E_HIP, "dtts_debug_check: %s", hipGetErrorString(e));
    *damaged_bytes = (int64_t)hout[0];
    if (hout[0]) h->err = "red zone damaged: " + std::to_string(hout[0]) + " bytes in " + std::to_string(zones.size()) + " zones; first: " +
                          (hout[1] < names.size() ? names[hout[1]] : std::string("?"));
    return DTTS_OK;
}

} // extern "C"
