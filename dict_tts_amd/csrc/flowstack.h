// The FVAE prior flow, reverse direction (modules/dict_tts/fvae_semantics.py:112-113 -> glow_modules.py ResidualCouplingBlock
// reverse: [Flip, ResidualCouplingLayer(mean_only)] x n_blocks), as ONE kernel: see flowstack.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

namespace dtts {

constexpr int FS_H = 64;         // prior_glow_hidden
constexpr int FS_K = 3;          // glow_kernel_size
constexpr int FS_HALF = 8;       // latent_size / 2
constexpr int FS_MAX_FLOWS = 8;
constexpr int FS_MAX_LAYERS = 8;

// packed weights of one flow block, in floats (all offsets multiples of 4)
constexpr size_t FS_IN_FRAGS = (size_t)FS_K * (FS_H / 8) * 4 * 64 * 4;    // in_layer: [tap][j][co-tile 0..3][lane] float4
constexpr size_t FS_RS_FRAGS = (size_t)(FS_H / 8) * 4 * 64 * 4;           // res_skip_layer: [j][co-tile 0..3][lane] float4 (last layer: tiles 0,1 used)
constexpr size_t FS_LAYER = FS_IN_FRAGS + 2 * FS_H + FS_RS_FRAGS + 2 * FS_H;   // frags, bias_in[128], frags, bias_rs[128]
constexpr size_t FS_PRE = (size_t)FS_H * FS_HALF + FS_H;                  // Wpre[64][8], bpre[64]
constexpr size_t FS_POST = (size_t)FS_HALF * FS_H + FS_HALF;              // Wpost'[8][64], bpost'[8]  (both negated: x1 += ...)
__host__ __device__ inline size_t fs_flow_floats(int layers) { return FS_PRE + (size_t)layers * FS_LAYER + FS_POST; }

struct FlowStackParams {
    const float* z_in;    // [B][T4][Z]
    float* z_out;         // [B][T4][Z]   (must not alias z_in: neighbouring chunks read each other's halo rows)
    const float* cond;    // [B][T4][ld_cond]: flow f (execution order), layer l at column (f * layers + l) * 2H, bias of the cond layer included
    int ld_cond;
    const float* w;       // n_flows * fs_flow_floats(layers) floats
    int B, T4, Z;
    int n_flows, layers;
    int x3;               // 1: weights packed as bf16 hi / lo fragments, contractions on three bf16 MFMAs per product; 0: exact fp32 MFMA
    int in_coff[FS_MAX_FLOWS], out_coff[FS_MAX_FLOWS];   // physical channel offsets of the logical x0 / x1 halves (flip parity)
};

// host packer: getters return the LOGICAL weights of flow block `f` in execution order
struct FlowStackHostWeights {
    // pre[c][i] (64 x 8), bpre[64]; in[l][co][ci][tap] (128 x 64 x 3), bin[l][128]; rs[l][co][ci] (128 or 64 x 64), brs[l][...]; post[o][c] (8 x 64), bpost[8]
    std::vector<float> pre, bpre, post, bpost;
    std::vector<std::vector<float>> in, bin, rs, brs;
};
void flowstack_pack(const FlowStackHostWeights& w, int layers, bool x3, std::vector<float>& out);   // appends fs_flow_floats(layers) floats

bool flowstack_supported(int hidden, int kernel, int layers, int blocks, int latent);
hipError_t flowstack_launch(const FlowStackParams& p, hipStream_t stream);

} // namespace dtts
