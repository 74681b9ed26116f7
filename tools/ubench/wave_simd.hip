// Which SIMD does each wave of a 512- / 768- / 1024-thread workgroup land on?  (HW_REG_HW_ID bits 5:4 = simd_id on gfx9)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
template <int T>
__global__ __launch_bounds__(T) void k(unsigned* out) {
    extern __shared__ char smem[];
    const unsigned id = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (T / 64) + (threadIdx.x >> 6)] = id;
    smem[threadIdx.x] = 0;
}
template <int T>
void run() {
    const int nb = 1024, W = T / 64;
    unsigned* d; hipMalloc(&d, nb * W * 4);
    hipFuncSetAttribute((const void*)k<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    k<T><<<nb, T, 150 * 1024>>>(d);
    std::vector<unsigned> h(nb * W); hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    printf("%d threads: waves per SIMD pattern of the first 6 workgroups (wave 0 .. %d):\n", T, W - 1);
    for (int b = 0; b < 6; ++b) { for (int w = 0; w < W; ++w) printf("%d ", (h[b * W + w] >> 4) & 3); printf("\n"); }
    int rr = 0;      // workgroups in which waves w and w + 4 always share a SIMD
    for (int b = 0; b < nb; ++b) {
        bool ok = true;
        for (int w = 0; w + 4 < W; ++w) ok = ok && (((h[b * W + w] >> 4) & 3) == ((h[b * W + w + 4] >> 4) & 3));
        rr += ok;
    }
    printf("  workgroups where wave w and w + 4 share a SIMD for every w: %d of %d\n", rr, nb);
    hipFree(d);
}
int main() { run<512>(); run<768>(); run<1024>(); return 0; }
