// Microbenchmark: sustained v_mfma_f32_32x32x16_f16 rate on this box (random operands, registers only),
// at 1 and 2 waves per SIMD, 10 independent accumulators per wave (the conv kernel's shape).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(512) void mfma_loop(const _Float16* in, float* out, int iters) {
    half8 a0 = *(const half8*)(in + threadIdx.x * 8), a1 = *(const half8*)(in + 4096 + threadIdx.x * 8);
    half8 b[5];
    for (int i = 0; i < 5; ++i) b[i] = *(const half8*)(in + 8192 + i * 4096 + threadIdx.x * 8);
    float16v acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16((i & 1) ? a1 : a0, b[(i >> 1) % 5], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the conv kernels' instruction: v_mfma_f32_16x16x32_f16, 40 independent accumulators per wave (8 x 5 tiles), same operand reuse
typedef float float4m __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void mfma_loop16(const _Float16* in, float* out, int iters) {
    half8 a[8], b[5];
    for (int i = 0; i < 8; ++i) a[i] = *(const half8*)(in + i * 4096 + threadIdx.x * 8);
    for (int i = 0; i < 5; ++i) b[i] = *(const half8*)(in + 32768 + i * 4096 + threadIdx.x * 8);
    float4m acc[8][5];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 5; ++j) acc[i][j] = float4m{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j], a[i], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 5; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    _Float16* in; float* out;
    const int n = 65536;
    hipMalloc(&in, n * 2); hipMalloc(&out, 256 * 512 * 4 * 4);
    _Float16* h = (_Float16*)malloc(n * 2);
    srand(1);
    for (int i = 0; i < n; ++i) h[i] = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
    hipMemcpy(in, h, n * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    if (argc > 1) {     // long run for tools/power_probe.sh: `mfma_peak <launches>` back to back, 2 waves per SIMD
        const int reps = atoi(argv[1]), iters = 20000;
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(mfma_loop<10>, dim3(256), dim3(512), 0, 0, in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flop = (double)reps * 256.0 * 8 * iters * 10.0 * 2.0 * 32 * 32 * 16;
        printf("%d launches back to back, 2 waves/SIMD: %.1f ms, %.1f TFLOP/s sustained\n", reps, ms, flop / ms / 1e9);
        if (argc > 2) {   // `mfma_peak <launches> 16`: then the same with the conv kernels' 16x16x32 instruction (40 accumulators)
            const int it16 = 10000;
            hipEventRecord(e0);
            for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(mfma_loop16, dim3(256), dim3(512), 0, 0, in, out, it16);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            flop = (double)reps * 256.0 * 8 * it16 * 40.0 * 2.0 * 16 * 16 * 32;
            printf("%d launches back to back, 16x16x32, 2 waves/SIMD: %.1f ms, %.1f TFLOP/s sustained (%s)\n", reps, ms, flop / ms / 1e9, hipGetErrorString(hipGetLastError()));
        }
        return 0;
    }
    for (int threads : {256, 512}) {
        for (int rep = 0; rep < 3; ++rep) {
            const int iters = 20000;
            hipEventRecord(e0);
            hipLaunchKernelGGL(mfma_loop<10>, dim3(256), dim3(threads), 0, 0, in, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flop = 256.0 * (threads / 64) * iters * 10.0 * 2.0 * 32 * 32 * 16;
            printf("threads/WG %d (waves/SIMD %d): %.3f ms  %.1f TFLOP/s\n", threads, threads / 256, ms, flop / ms / 1e9);
        }
    }
    return 0;
}
