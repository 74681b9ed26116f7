// How fast can ONE CU fill LDS from an L2-resident buffer with global_load_lds_dwordx4?  (The weight streams of conv_zs_kernel and
// attn_block_kernel, and the capacity argument against Winograd in DESIGN.md section 5, rest on this number.)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench/lds_fill.hip -o lds_fill && ./lds_fill [MB of source] [waves per WG]
// One workgroup per CU; every wave streams its 1-KiB share of successive pieces (1 KiB per wave) into a 4-slot ring, three pieces in
// flight (counted vmcnt), nothing reads the ring.  Prints GB/s per CU and chip-wide.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void fill_kernel(const char* src, size_t src_bytes, int pieces, unsigned long long* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int PIECE = WAVES * 1024;
    const size_t npos = src_bytes / PIECE;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    size_t pos = (size_t)blockIdx.x * 7 % npos;          // workgroups start at different pieces of the shared buffer
#pragma unroll 1
    for (int t = 0; t < pieces; ++t) {
        const char* s = src + pos * PIECE + w * 1024 + lane * 16;
        char* d = smem + (t & 3) * PIECE + w * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                         (__attribute__((address_space(3))) void*)d, 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        pos = pos + 1 == npos ? 0 : pos + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_memrealtime() - t0;
}

template <int WAVES>
static void run(const char* dsrc, size_t bytes, int pieces, unsigned long long* dout) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&fill_kernel<WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * WAVES * 1024);
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(fill_kernel<WAVES>, dim3(256), dim3(WAVES * 64), 4 * WAVES * 1024, 0, dsrc, bytes, pieces, dout);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(256);
        hipMemcpy(h.data(), dout, 256 * 8, hipMemcpyDeviceToHost);
        double tick = 0; for (auto v : h) tick += (double)v;
        tick /= 256.0;                                    // 10-ns ticks per workgroup
        const double per_wg = (double)pieces * WAVES * 1024;
        printf("%d waves, %.1f MB source, %d pieces of %d KiB per CU: %.1f GB/s per CU in-kernel (mean), %.2f TB/s chip by events (%.3f ms)\n",
               WAVES, bytes / 1048576.0, pieces, WAVES, per_wg / (tick * 10.0), 256.0 * per_wg / (ms * 1e-3) * 1e-12, ms);
    }
}

int main(int argc, char** argv) {
    const double mb = argc > 1 ? atof(argv[1]) : 1.84;
    const int waves = argc > 2 ? atoi(argv[2]) : 8;
    const size_t bytes = ((size_t)(mb * 1048576.0) / 16384) * 16384;
    char* dsrc; hipMalloc(&dsrc, bytes); hipMemset(dsrc, 1, bytes);
    unsigned long long* dout; hipMalloc(&dout, 256 * 8);
    const int pieces = 4096;
    if (waves == 4) run<4>(dsrc, bytes, pieces, dout); else run<8>(dsrc, bytes, pieces, dout);
    return 0;
}
