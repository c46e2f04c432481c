// Which XCD does workgroup i of a 1-D grid land on?  (HW_REG_XCC_ID, gfx940+: hwreg 20, bits 3:0.)  Prints blockIdx -> xcc for the first
// workgroups and whether xcc == blockIdx % 8 for all of them, for a plain launch and for a persistent-style launch of 512 workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));
}
int main() {
    for (int n : {512, 4096}) {
        int* d;
        hipMalloc(&d, n * sizeof(int));
        hipLaunchKernelGGL(k, dim3(n), dim3(256), 0, 0, d);
        std::vector<int> h(n);
        hipMemcpy(h.data(), d, n * sizeof(int), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < n; ++i) bad += (h[i] & 15) != (i % 8);
        printf("grid %d: first 16 xcc ids:", n);
        for (int i = 0; i < 16; ++i) printf(" %d", h[i] & 15);
        printf("  mismatches vs blockIdx %% 8: %d\n", bad);
        hipFree(d);
    }
    return 0;
}
