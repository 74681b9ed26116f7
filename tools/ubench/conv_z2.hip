// conv_z2_kernel (experiment): conv_zs_kernel's wave tile and zero-skipping main loop with TWO independent workgroups per CU.
//
// conv_zs_kernel runs one 512-thread workgroup per CU (4 boards x 320 channels, 160 KB of LDS); its 8 waves reach their
// epilogue together, so for 11 % (GroupNorm epilogue) to 27 % (fused block tail) of a tile's time the CU's matrix pipe idles
// (stagger experiments: the epilogues are per-CU latency / issue chains, not chip-level HBM bursts).
// Here a workgroup is 4 waves (one per SIMD) = ONE board pair x 320 channels in 80 KB of LDS, and two of them share a CU:
// the epilogue of one runs under the main loop of the other.  Per-wave tile, fragment reads, MFMA order and therefore the
// results are those of conv_zs_kernel bit for bit.  What it costs: each workgroup streams the whole weight tensor for 2
// boards (2x the L2 -> LDS weight traffic per CU), the weight ring is 2 half-tiles deep instead of 4, and the two waves of
// a SIMD are not phase-locked by a barrier.
//
// LDS (81,920 B): activations [2 chunk buffers][128 rows x 128 B] = 32 KB (conv_zs's image of one board pair), weight
// half-tiles [2][20 KB], zero region 8 KB.  Phase y (one half-tile = 32 k of one tap):
//     s_waitcnt vmcnt(0)  -- this wave's DMA pieces of half-tile y (issued in phase y - 1) have landed
//     s_barrier           -- everyone's have; everyone's fragment reads of phase y - 1 are complete (they preceded its MFMAs)
//     DMA issue: half-tile y + 1 -> slot (y + 1) & 1 (5 pieces per wave) [+ one activation piece of the next chunk]
//     fragment reads of y (7-8 activation + 5 weight ds_read_b128), 35-40 MFMAs
#include "../../matrix0_amd/csrc/kernel_common.h"
#include "../../matrix0_amd/csrc/conv_zs_epilogue.h"

namespace {
constexpr int Z2_A_BYTES = 128 * 128;                 // 2 boards x 64 squares x 64 channels fp16
constexpr int Z2_WH_BYTES = 320 * 64;                 // 320 output channels x 32 k fp16
constexpr int Z2_OFF_W = 2 * Z2_A_BYTES;              // 32,768
constexpr int Z2_OFF_Z = Z2_OFF_W + 2 * Z2_WH_BYTES;  // 73,728
constexpr int Z2_LDS = Z2_OFF_Z + 8192;               // 81,920
}

__device__ __forceinline__ void z2_glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
#define Z2_FENCE() asm volatile("" ::: "memory")

#ifdef SW_STAMP
__device__ unsigned long long* g_z2_stamp;
#endif
__device__ int g_z2_delay = 0;          // 10-ns ticks

template <int EPI, int ACT>
__global__ __launch_bounds__(256, 2) void conv_z2_kernel(GemmArgs a) {
    constexpr int NG = 5, MT = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* A_lds = smem;
    char* W_lds = smem + Z2_OFF_W;
    char* Z_lds = smem + Z2_OFF_Z;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // = channel quarter wn
    const int wn = wave;
    const int m0 = blockIdx.x * 128;
    const int n0 = blockIdx.y * 320;
    const int Cin = a.Cin;
    const int nchunk = Cin >> 6;
    const int NH = nchunk * 18;
    const int c15 = lane & 15;
    const int q = lane >> 4;

    // the two workgroups of a CU must not run in lockstep (same start, same duration: their epilogues would coincide):
    // the second batch of first-round workgroups (blocks 256..511: the second slot of every CU) starts g_z2_delay x 10 ns late
    if (blockIdx.x >= 256 && blockIdx.x < 512 && g_z2_delay > 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)g_z2_delay) __builtin_amdgcn_s_sleep(16);
    }
    reinterpret_cast<uint4*>(Z_lds)[tid] = make_uint4(0, 0, 0, 0);
    reinterpret_cast<uint4*>(Z_lds)[256 + tid] = make_uint4(0, 0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // the zero region is written before the first barrier

    const char* in_bytes = reinterpret_cast<const char*>(a.in);
    const size_t w_kt_stride = (size_t)gridDim.y * (2 * Z2_WH_BYTES);
    const char* w_base = reinterpret_cast<const char*>(a.w) + (size_t)blockIdx.y * (2 * Z2_WH_BYTES) + wave * 1024;   // wave-uniform

    // activation piece qq (1 KiB = rows 8qq .. 8qq+7 of the 128-row tile), chunk `chunk`
    auto issue_A_piece = [&](int chunk, int qq) __attribute__((always_inline)) {
        const int p = 8 * qq + (lane >> 3);
        const int key = ((p >> 1) & 3) | (((p >> 6) & 1) << 2);
        const char* src = in_bytes + ((size_t)(m0 + p) * Cin + (size_t)chunk * 64) * 2 + 16 * ((lane & 7) ^ key);
        z2_glds16(src, A_lds + (chunk & 1) * Z2_A_BYTES + qq * 1024);
    };
    const uint32_t w_lane = (uint32_t)lane * 16u;
    // half-tile y: K-tile y >> 1, half y & 1; this wave's pieces wave, wave + 4, ..., wave + 16
    auto issue_W = [&](const char* wp_, int slot) __attribute__((always_inline)) {
        char* dst = W_lds + slot * Z2_WH_BYTES + wave * 1024;
#pragma unroll
        for (int i = 0; i < 5; ++i) z2_glds16(wp_ + w_lane + i * 4096, dst + i * 4096);
    };

    float4v acc[MT][NG];
    static_for<0, MT>([&](auto mi) __attribute__((always_inline)) {
        static_for<0, NG>([&](auto ni) __attribute__((always_inline)) {
            acc[decltype(mi)::value][decltype(ni)::value] = float4v{0.f, 0.f, 0.f, 0.f};
        });
    });

    const int wfx = ((4 - ((c15 >> 2) & 3)) & 3) ^ q;
    const int wrow_off = (wn * 80 + c15) * 64;
    const int lx = c15 & 7;
    const int lb = c15 >> 3;
    const int arow0 = lb * 64 + lx;

    // prologue: chunk 0 activations (4 pieces per wave) and half-tile 0
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_A_piece(0, wave * 4 + i);
    const char* w_ptr = w_base;                         // half-tile to be issued next
    issue_W(w_ptr, 0);
    w_ptr += Z2_WH_BYTES;                               // half-tile 1 = K-tile 0, half 1
    int t_next = 1;

    int y = 0;
#ifdef SW_STAMP
    const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll 1
    for (int c = 0; c < nchunk; ++c) {
        const char* Ab = A_lds + (c & 1) * Z2_A_BYTES;
        const bool next_chunk = c + 1 < nchunk;
        static_for<0, 3>([&](auto t3_) __attribute__((always_inline)) {
            constexpr int t3 = decltype(t3_)::value;     // dy = t3 - 1
            constexpr int LO = t3 == 0 ? 1 : 0;
            constexpr int HI = t3 == 2 ? 7 : 8;
            const int rbase = arow0 + (t3 == 2 ? 8 : 0) - 1;
#pragma unroll 1
            for (int dxi = 0; dxi < 3; ++dxi) {
                const int xx = lx + dxi - 1;
                const bool ok = (unsigned)xx < 8u;
                const int key = ((xx >> 1) & 3) | (lb << 2);
                const char* rowp = Ab + (rbase + dxi) * 128;
                const char* ap0 = ok ? rowp + 16 * (key ^ q) : Z_lds;
                const char* ap1 = ok ? rowp + 16 * (key ^ q ^ 4) : Z_lds;
                static_for<0, 2>([&](auto h_) __attribute__((always_inline)) {
                    constexpr int h = decltype(h_)::value;
                    const int yh = y + h;
                    const char* ap = h ? ap1 : ap0;
                    // lgkmcnt(0): hipcc sinks the last weight-fragment reads of the previous phase down to here; they must have
                    // RETURNED before the barrier, since the DMA issued behind it refills the slot they read
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    Z2_FENCE();
                    __builtin_amdgcn_s_barrier();
                    Z2_FENCE();
                    // DMA of the next half-tile into the slot whose reads ended with phase yh - 1
                    auto issue_dma = [&]() __attribute__((always_inline)) {
                        const bool more = t_next < NH;
                        if (more) issue_W(w_ptr, t_next & 1);
                        const size_t inc = (t_next & 1) ? (w_kt_stride - Z2_WH_BYTES) : (size_t)Z2_WH_BYTES;
                        w_ptr += more ? inc : 0;
                        t_next += 1;
                        // the next chunk's activations: this wave's 4 pieces in the h == 1 phases of taps (t3, dxi) = (0,1) (0,2) (1,0) (1,1)
                        if constexpr (h == 1 && t3 < 2) {
                            const int k = t3 * 3 + dxi - 1;             // 0..3 for the four phases above
                            if (next_chunk && (unsigned)k < 4u) issue_A_piece(c + 1, wave * 4 + k);
                        }
                    };
#ifndef Z2_DMA_LATE
                    issue_dma();
#endif
                    Z2_FENCE();
                    half8 fa[MT], fb[NG];
                    const char* Wb = W_lds + (yh & 1) * Z2_WH_BYTES + wrow_off + 16 * wfx;
                    static_for<LO, HI>([&](auto mi_) __attribute__((always_inline)) {
                        constexpr int mi = decltype(mi_)::value;
                        fa[mi] = *reinterpret_cast<const half8*>(ap + (mi - LO) * 1024);
                    });
                    static_for<0, NG>([&](auto ni_) __attribute__((always_inline)) {
                        constexpr int ni = decltype(ni_)::value;
                        fb[ni] = *reinterpret_cast<const half8*>(Wb + ni * 1024);
                    });
#ifdef Z2_SETPRIO
                    __builtin_amdgcn_s_setprio(1);
#endif
                    static_for<0, NG>([&](auto ni_) __attribute__((always_inline)) {
                        constexpr int ni = decltype(ni_)::value;
                        static_for<LO, HI>([&](auto mi_) __attribute__((always_inline)) {
                            constexpr int mi = decltype(mi_)::value;
                            if constexpr (EPI == 3 || EPI == 5)
                                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[mi], fb[ni], acc[mi][ni], 0, 0, 0);
                            else
                                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[ni], fa[mi], acc[mi][ni], 0, 0, 0);
                        });
#ifdef Z2_DMA_LATE
                        if constexpr (ni == Z2_DMA_LATE) { Z2_FENCE(); issue_dma(); Z2_FENCE(); }
#endif
                    });
#ifdef Z2_SETPRIO
                    __builtin_amdgcn_s_setprio(0);
#endif
                    Z2_FENCE();
                });
                y += 2;
            }
        });
    }
#ifdef SW_STAMP
    if (tid == 0) {
        unsigned long long* o = g_z2_stamp + (size_t)blockIdx.x * 4;
        o[0] = st_c0; o[1] = st_r0; o[2] = __builtin_amdgcn_s_memtime(); o[3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // every wave's reads are done before anyone stages output

#ifdef PP_NO_EPILOGUE
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NG; ++ni) asm volatile("" :: "v"(acc[mi][ni]));
#else
    zs_tile_epilogue<EPI, ACT>(acc, a, smem + wave * 20480, m0, n0, 0, wn, lane);
#endif
}

template <int EPI, int ACT>
static hipError_t launch_conv_z2_e(const GemmArgs& a, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_z2_kernel<EPI, ACT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, Z2_LDS);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid(a.Mrows / 128, a.Npad / 320);
    hipLaunchKernelGGL((conv_z2_kernel<EPI, ACT>), grid, dim3(256), Z2_LDS, st, a);
    return hipGetLastError();
}

hipError_t launch_conv_z2(const GemmArgs& a, hipStream_t st) {
    if (a.Cin % 64 != 0 || a.Npad % 320 != 0 || a.Mrows % 128 != 0) return hipErrorInvalidValue;
    if (a.mul != nullptr || a.out_f32 != 0 || a.res != nullptr) return hipErrorInvalidValue;
    if ((size_t)a.Mrows * a.ldo * 2 >= ((size_t)1 << 32)) return hipErrorInvalidValue;
    if (a.gn_gamma != nullptr) {
        if (a.epi_act == ACT_SILU) return launch_conv_z2_e<1, ACT_SILU>(a, st);
        return hipErrorInvalidValue;
    }
    return a.epi_act == ACT_NONE ? launch_conv_z2_e<0, ACT_NONE>(a, st) : hipErrorInvalidValue;
}
