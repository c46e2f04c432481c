// Micro-benchmark: what does the MFMA pipe sustain on THIS chip with real operand data?
// A register-resident loop of independent v_mfma_f32_32x32x16_{f16,bf16} (no memory traffic), 2 or 3 waves per SIMD, with
// (a) all-zero operands and (b) random operands of the magnitude the vocoder sees.  Prints TFLOP/s and, from the wall time, the
// fraction of the 2.5 PFLOP/s dense peak (MI355X_MICROARCH.md).  Power management lowers the clock when the matrix cores
// toggle real data; this measures the ceiling a data-carrying kernel can reach, next to the spec peak the rooflines quote.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o tools/micro/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <bool BF>
__global__ __launch_bounds__(256) void mfma_loop(const uint4* ab, float* out, int iters) {
    const int lane = threadIdx.x;
    uint4 a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = ab[(i * 256 + lane) % 2048];
    for (int i = 0; i < 2; ++i) b[i] = ab[((4 + i) * 256 + lane) % 2048];
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (BF)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i & 3]), __builtin_bit_cast(bf16x8, b[i >> 2]), acc[i], 0, 0, 0);
            else
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i & 3]), __builtin_bit_cast(f16x8, b[i >> 2]), acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) out[0] = s;   // keep the loop alive
}

static unsigned short f2h(float f) { _Float16 h = (_Float16)f; unsigned short u; __builtin_memcpy(&u, &h, 2); return u; }
static unsigned short f2b(float f) { unsigned u; __builtin_memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    uint4* dab;
    float* dout;
    hipMalloc(&dab, 2048 * 16);
    hipMalloc(&dout, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int bf = 0; bf < 2; ++bf)
        for (int data = 0; data < 3; ++data) {
            std::vector<unsigned short> h(2048 * 8);
            srand(1);
            for (auto& v : h) {
                float x = data == 0 ? 0.f : (data == 1 ? (rand() / (float)RAND_MAX - 0.5f) * 1e-2f : (rand() / (float)RAND_MAX - 0.5f) * 2.f);
                v = bf ? f2b(x) : f2h(x);
            }
            hipMemcpy(dab, h.data(), h.size() * 2, hipMemcpyHostToDevice);
            for (int wg_per_cu = 2; wg_per_cu <= 3; ++wg_per_cu) {
                const int blocks = 256 * wg_per_cu;
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0);
                    if (bf) hipLaunchKernelGGL(mfma_loop<true>, dim3(blocks), dim3(256), 0, 0, dab, dout, iters);
                    else hipLaunchKernelGGL(mfma_loop<false>, dim3(blocks), dim3(256), 0, 0, dab, dout, iters);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    const double flop = (double)blocks * 4 * iters * 8 * 32768.0;
                    if (rep)
                        printf("%s operands %-22s %d waves/SIMD: %8.2f ms  %7.1f TFLOP/s = %5.1f %% of 2.5 PF\n", bf ? "bf16" : "f16 ",
                               data == 0 ? "all zero" : (data == 1 ? "random |x| < 5e-3" : "random |x| < 1"), wg_per_cu, ms,
                               flop / ms / 1e9, flop / ms / 1e9 / 2500.0 * 100);
                }
            }
        }
    return 0;
}
