// One fused ResBlock1 iteration for the C = 128 HifiGAN stage: see vpair.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "voc_el.h"

namespace dtts {

struct VPairParams {
    const float* x;       // fp32 residual stream in, [B][T][C]; must not alias y (neighbouring tiles read halos)
    float* y;             // mode 1: y = x';  mode 2: y += x';  mode 3: y = (y + x') / div and ya = bf16 leaky_relu(y, slope)
    unsigned short* ya;
    const uint4* w1;      // convs1[m] (dilated) / convs2[m] packed bf16 weights, single C_in chunk
    const uint4* w2;
    const float* b1;
    const float* b2;
    const int* lens;      // [B] valid rows
    int B, T, K, dil;
    int mode;
    int drop_y;           // mode 3 with ya: do not write the fp32 result (nothing reads it after the stage)
    float div, slope;
    int x16;              // (EL_F16 only) x is an fp16 stream [B][T][C] (written by a launch with y16); 0: fp32
    int y16;              // (EL_F16, mode 1 only) y is written as fp16 — the 16-bit inter-iteration stream of round 6; 0: fp32
    int el;               // 16-bit operand type of both convolutions: EL_BF16 (rb_common.h) or EL_F16; w1 / w2 are packed in that type
    unsigned* tile_ctr;   // persistent configurations: device counter (zero at launch) for dynamic tile claiming, or null = static w, w + G, ...
    int pre_off;          // (set by the launcher) byte offset of the tile table in dynamic LDS
    int tile_rows;        // (set by the launcher) rows of the LDS activation tile
    unsigned long long* ovf;     // fp16 range guard: device counter of unrepresentable activations (launches the GUARD instantiation), or null
    unsigned long long* stats;   // -DDTTS_ABLATE builds only: per-phase cycle sums of wave 0 (see vpair.hip), or null
    int dbg;              // -DDTTS_ABLATE builds only (DTTS_VCONV_DBG >> 8): 1 skip contractions, 2 skip epilogue, 4 skip staging, 8 skip xt write
};

bool vpair_supported(int C, int K, int dil);
hipError_t vpair_launch(const VPairParams& p, int C, hipStream_t stream);

} // namespace dtts
