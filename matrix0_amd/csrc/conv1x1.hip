// conv1x1_kernel: 1x1 conv (qkv, proj, pst) = GEMM [256 rows x Cin] x [Cin x 160] per workgroup, TWO workgroups per CU.
//
// The 1x1 convs have K = Cin = 320 only, so a workgroup's life is short and dominated by what is NOT the K loop: the
// first operand fetch, the 160 KB output tile going out, and -- with the 256 x 320 tile of conv_big_kernel, whose
// staged epilogue takes the whole LDS -- nothing else on the CU to hide them (qkv: 24 us per workgroup for 6.4 us of
// MFMA work, 305 us per launch).  Here the tile is 256 rows x 160 output channels: 4 waves (wave = board, 64 x 160 =
// the same 2 x 5 MFMA tiles and 160 accumulators per wave), 52 KB of operand buffers, an 80 KB staged output image, so
// two workgroups share a CU and one's prologue / epilogue overlaps the other's MFMA loop.  Weight traffic per row is
// unchanged (each workgroup streams the 160 columns it needs); the activations of a row block are read by twice as many
// workgroups, which the XCD-aware grid keeps on one L2.
//
// K is walked in steps of 32: activations 256 x 32 k (16 KB) and weights 160 x 32 k (10 KB) per step, double-buffered,
// global_load_lds, 64-byte LDS rows with the 16-byte chunk index XOR (row>>2)&3 (conflict-free ds_read_b128).
#include "kernel_common.h"
#include "conv_epilogue.h"

__device__ __forceinline__ void c11_glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv1x1_kernel(GemmArgs a) {
    constexpr int NT = 5;
    constexpr int A_BYTES = 256 * 64;     // 256 rows x 32 k fp16
    constexpr int W_BYTES = 160 * 64;     // 160 output channels x 32 k fp16
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* A_lds = smem;                   // [2][A_BYTES]
    char* W_lds = smem + 2 * A_BYTES;     // [2][W_BYTES]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // board within the tile
    // 1-D grid, XCD-aware: the N blocks of one row block get ids that are consecutive on ONE XCD
    const int nblk = a.Npad / 160, rblk = a.Mrows >> 8;
    int rb, nb;
    if ((rblk & 7) == 0) {
        const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
        rb = (q / nblk) * 8 + xcd;
        nb = q % nblk;
    } else {
        rb = blockIdx.x % rblk;
        nb = blockIdx.x / rblk;
    }
    const int m0 = rb * 256, n0 = nb * 160;
    const int Cin = a.Cin;
    const int NS = Cin >> 5;              // K steps
    const int half = lane >> 5;
    const int r31 = lane & 31;

    // DMA: activations piece q (1 KiB) = rows 16q..16q+15, LDS unit (row, slot) holds chunk slot ^ ((row>>2)&3);
    // wave w moves pieces w, w+4, w+8, w+12; weights [step][Npad/160][160][32] pre-swizzled: 10 linear pieces
    const uint32_t a_lane = (uint32_t)(lane >> 2) * (uint32_t)Cin * 2u + 16u * (uint32_t)((lane & 3) ^ ((lane >> 4) & 3));
    const char* a_base = reinterpret_cast<const char*>(a.in) + ((size_t)(m0 + 16 * wave) * Cin) * 2 + a_lane;
    const size_t a_q4 = (size_t)64 * Cin * 2;                       // piece q + 4 = 64 rows further
    const char* w_base = reinterpret_cast<const char*>(a.w) + (size_t)nb * W_BYTES + lane * 16;
    const size_t w_step = (size_t)nblk * W_BYTES;
    auto issue = [&](int s) __attribute__((always_inline)) {
        const char* as = a_base + (size_t)s * 64;
        char* ad = A_lds + (s & 1) * A_BYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) c11_glds16(as + i * a_q4, ad + i * 4096);
        const char* ws = w_base + (size_t)s * w_step;
        char* wd = W_lds + (s & 1) * W_BYTES;
        c11_glds16(ws + wave * 1024, wd + wave * 1024);
        c11_glds16(ws + (4 + wave) * 1024, wd + (4 + wave) * 1024);
        if (wave < 2) c11_glds16(ws + (8 + wave) * 1024, wd + (8 + wave) * 1024);
    };

    float16v acc[2][NT];
    static_for<0, 2>([&](auto mi) __attribute__((always_inline)) {
        static_for<0, NT>([&](auto ni) __attribute__((always_inline)) {
            acc[decltype(mi)::value][decltype(ni)::value] = float16v{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f,
                                                                      0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        });
    });
    const int fx = ((r31 >> 2) & 3) ^ half;                 // swizzle key ^ k-half, both operands
    const int arow_off = (wave * 64 + r31) * 64;
    const int wrow_off = r31 * 64;

    issue(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int s = 0; s < NS; ++s) {
        if (s + 1 < NS) issue(s + 1);                       // the other buffer: its last reads ended at the barrier above
        const char* Ab = A_lds + (s & 1) * A_BYTES + arow_off;
        const char* Wb = W_lds + (s & 1) * W_BYTES + wrow_off;
        static_for<0, 2>([&](auto j_) __attribute__((always_inline)) {
            constexpr int j = decltype(j_)::value;
            const int ko = 16 * (fx ^ (j << 1));
            const half8 fa0 = *reinterpret_cast<const half8*>(Ab + ko);
            const half8 fa1 = *reinterpret_cast<const half8*>(Ab + 32 * 64 + ko);
            static_for<0, NT>([&](auto ni_) __attribute__((always_inline)) {
                constexpr int ni = decltype(ni_)::value;
                const half8 fb = *reinterpret_cast<const half8*>(Wb + ni * 2048 + ko);
                acc[0][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa0, fb, acc[0][ni], 0, 0, 0);
                acc[1][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa1, fb, acc[1][ni], 0, 0, 0);
            });
        });
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();                                    // step s+1 landed; step s no longer read
    }
    conv_tile_epilogue<EPI, ACT_NONE, NT>(acc, a, smem + wave * (NT * 64 * 64), m0, n0, wave, 0, lane);
}

template <int EPI>
static hipError_t launch_conv1x1_e(const GemmArgs& a, hipStream_t st) {
    const size_t lds = 80 * 1024;      // operand buffers 52 KB; the epilogue stages the 256 x 160 fp16 tile (80 KB)
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_kernel<EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid((a.Mrows / 256) * (a.Npad / 160));
    hipLaunchKernelGGL((conv1x1_kernel<EPI>), grid, dim3(256), lds, st, a);
    return hipGetLastError();
}

// plain epilogue only (bias / scale / statistics, fp16 out, no activation); a.w in the [step][N/160][160][32] layout
hipError_t launch_conv1x1(const GemmArgs& a, hipStream_t st) {
    if (a.Cin % 32 != 0 || a.Npad % 160 != 0 || a.Mrows % 256 != 0) return hipErrorInvalidValue;
    if (a.mul != nullptr || a.out_f32 != 0 || a.epi_act != ACT_NONE || a.gn_gamma != nullptr || a.res != nullptr)
        return hipErrorInvalidValue;
    if ((size_t)a.Mrows * a.ldo * 2 >= ((size_t)1 << 32)) return hipErrorInvalidValue;   // 32-bit store offsets
    return launch_conv1x1_e<0>(a, st);
}
