// 16-bit operand type of the fused vocoder kernels (vpair.hip, rblock.hip): EL_BF16 = v_mfma_f32_32x32x16_bf16 (8-bit
// significand), EL_F16 = v_mfma_f32_32x32x16_f16 (11-bit significand, same rate) — the ResBlock stages of DTTS_VOC_F16.
#pragma once
namespace dtts {
enum { EL_BF16 = 0, EL_F16 = 1 };
}
