// Which SIMD does each wave of a 512-thread workgroup land on?  (HW_REG_HW_ID bits 5:4 = simd_id on gfx9)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(512) void k(unsigned* out) {
    extern __shared__ char smem[];
    const unsigned id = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
    smem[threadIdx.x] = 0;
}
int main() {
    const int nb = 1024;
    unsigned* d; hipMalloc(&d, nb * 8 * 4);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    k<<<nb, 512, 150 * 1024>>>(d);
    unsigned h[nb * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int hist[8][4] = {};
    for (int b = 0; b < nb; ++b) for (int w = 0; w < 8; ++w) hist[w][(h[b * 8 + w] >> 4) & 3]++;
    for (int w = 0; w < 8; ++w) printf("wave %d: simd0 %d simd1 %d simd2 %d simd3 %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    int same04 = 0, same01 = 0;
    for (int b = 0; b < nb; ++b) {
        same04 += ((h[b * 8] >> 4) & 3) == ((h[b * 8 + 4] >> 4) & 3);
        same01 += ((h[b * 8] >> 4) & 3) == ((h[b * 8 + 1] >> 4) & 3);
    }
    printf("blocks where wave0/wave4 share a SIMD: %d of %d; wave0/wave1: %d\n", same04, nb, same01);
    for (int b = 0; b < 4; ++b) { for (int w = 0; w < 8; ++w) printf("%d ", (h[b * 8 + w] >> 4) & 3); printf("\n"); }
    return 0;
}
