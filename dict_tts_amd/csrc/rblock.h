// Fused HifiGAN ResBlock1 kernel for the narrow stages: see rblock.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "voc_el.h"

namespace dtts {

struct RBlockParams {
    const float* x;        // stage input, fp32 [B][T][C] (the transposed conv's output)
    float* S;              // stage accumulator xs, fp32 [B][T][C]
    unsigned short* Sa;    // bf16 leaky_relu(xs / num_kernels, slope): next stage's input (mode 2 only)
    // the ResBlocks of this launch: one (rb[0]), or — nrb = 2..3, C <= 64 — ALL ResBlocks of the stage on the same tile (one launch per
    // stage: x is read from HBM once per tile, re-read from L2 / Infinity Cache by the other ResBlocks, and the stage sum is
    // read-modify-written through the cache instead of HBM: 7-8 passes over the stage's tensors become ~2)
    struct Set {
        const uint4* w1[3];   // convs1[m] / convs2[m] packed weights, Kp taps (zero padded)
        const uint4* w2[3];
        const float* b1[3];
        const float* b2[3];
        int dil[3];
        int K, Kp;
    } rb[3];
    int nrb;
    const int* lens;       // [B] valid rows
    int B, T;
    int K;                 // the LARGEST kernel size of the launch: the tile's halo is 6 (K - 1) rows per side
    int mode;              // what the (first) ResBlock of the launch does with the stage sum: 0: xs = r ; 1: xs += r ; 2: xs = (xs + r) / div, and emit Sa
    int last_mode;         // nrb > 1: the same for the launch's LAST ResBlock (2 when it is the stage's last, else 1); those in between accumulate (1)
    int drop_S;            // mode 2 with Sa: do not write the fp32 xs (nothing reads it after the stage)
    float div, slope;
    // fused conv_post + tanh (last stage, mode 2, C = 32): the stage output never reaches HBM, the waveform is written instead
    float* wav;            // [B][T] or null
    const float* post_w;   // conv_post weight as [7 taps][C] fp32
    const float* post_b;   // [1]
    int el;                // 16-bit operand type: EL_BF16 (rb_common.h) or EL_F16; the packed weights are in that type
    unsigned* tile_ctr;    // persistent configurations: device counter (zero at launch) for dynamic tile claiming, or null = static w, w + G, ...
    int pre_off;           // (set by the launcher) byte offset of the tile-count table in dynamic LDS
    unsigned* bad;         // always-on detector of the fused conv_post: device counter of NON-FINITE pre-tanh values (an fp16 operand overflowed upstream), or null
    unsigned long long* ovf;   // fp16 range guard: device counter of unrepresentable activations (launches the GUARD instantiation), or null
    int small_tile;        // tune bit 14: C = 64 keeps 512-row tiles at k >= 7 (A/B against the default 640)
    int pingpong;          // 1: the phase-shifted two-group form (rblock2.hip; an experiment, dtts_config.tune_flags bit 7)
    int s_private;         // 0, or the byte capacity of S when it holds one private TT-row strip per TILE (fused launch + fused conv_post)
    unsigned long long* stats;   // -DDTTS_ABLATE builds only (DTTS_RB_STATS): per-phase cycle sums of each group's wave 0 (rblock2.hip)
    int dbg;               // -DDTTS_ABLATE builds only; tuning ablations (DTTS_VCONV_DBG): 1 skip contractions, 2 skip epilogue, 4 skip the x load, 8 skip write_act
};

bool rblock_supported(int C, int K);
int rblock_padded_taps(int C, int K);
hipError_t rblock_launch(const RBlockParams& p, int C, hipStream_t stream);
// rows a fused launch with the fused conv_post needs in S (one private strip per tile), or 0 when the configuration does not fuse
long long rblock_private_rows(int C, int Kmax, int B, int T);
// rblock2.hip: two phase-shifted groups of waves per workgroup (one computing while the other rewrites / loads / stores)
bool rblock2_supported(int C, int K, bool wav);
hipError_t rblock2_launch(const RBlockParams& p, int C, hipStream_t stream);

} // namespace dtts
