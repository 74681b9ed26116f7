// conv_pp16_kernel: the 3x3 320->320 implicit-GEMM conv of tools/ubench/conv_pp.hip (same workgroup tile, LDS images, DMA ring, ping-pong
// groups and barrier protocol -- read its header) on v_mfma_f32_16x16x32_f16 instead of v_mfma_f32_32x32x16_f16.
//
// Why the shape matters: the main loop is POWER-bound, not issue-bound.  In-kernel stamps (tools/ubench, -DSW_STAMP, random
// operands) show the 32x32x16 loop at 87.7 % of the MFMA issue rate but at an in-kernel clock of 1.52-1.57 GHz; the same
// loop on 16x16x32 spends more cycles (82 %) at 1.75-1.82 GHz and finishes 5.6-7 % sooner (MI355X_MICROARCH.md, DVFS
// give-back item 7: the chip holds a higher clock on this shape).  Small cycle savings on either loop (DMA placement inside
// the compute section, early barriers, static priority) left the wall time unchanged in round 2 -- the clock falls as the MFMAs
// pack closer; round 3's move of the DMA out of the compute section altogether (-11 % cycles, conv_zs.hip) came back as -4 % wall.
//
// One phase pair per half-tile (32 k): a wave reads 4 activation fragments (16 squares x 32 k each) and 10 weight fragments
// (16 channels x 32 k), then issues 40 MFMAs; accumulators = 4 x 10 tiles of 16 squares x 16 channels (160 registers).
// Lane l = (c15 = l & 15, q = l >> 4): A fragment = row (square) c15 of the tile, k 8q..8q+7; B fragment = channel c15, same k.
// LDS images as in tools/ubench/conv_pp.hip; the weight rows' 16-byte chunk swizzle is  chunk ^ ((4 - (row >> 2)) & 3)  (pack_gemm), which is
// conflict-free for this read pattern and for the 32x32 one.
#include "kernel_common.h"
#include "conv_epilogue16.h"
#include "conv_tail16.h"

__device__ __forceinline__ void p16_glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
#define P16_FENCE() asm volatile("" ::: "memory")

#ifdef SW_STAMP
__device__ unsigned long long* g_p16_stamp;     // [blocks][4]: s_memtime / s_memrealtime at main-loop start and end
#endif

template <int EPI, int ACT>
__global__ __launch_bounds__(512) void conv_pp16_kernel(GemmArgs a) {
    constexpr int NG = 10, MT = 4;
    constexpr int A_BYTES = 256 * 128;    // 4 boards x 64 squares x 64 channels fp16, 128-byte rows
    constexpr int WH_BYTES = 320 * 64;    // 320 output channels x 32 k fp16, 64-byte rows
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* A_lds = smem;                   // [2][A_BYTES]   chunk c in buffer c&1
    char* W_lds = smem + 2 * A_BYTES;     // [4][WH_BYTES]  half-tile y in slot y&3
    char* Z_lds = W_lds + 4 * WH_BYTES;   // one all-zero square (128 B)
    char* D_lds = Z_lds + 128;            // [4][1024] sink of the filler DMA pieces

#ifdef SW_STAMP
    const unsigned long long st_entry = __builtin_amdgcn_s_memrealtime();
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3;              // board within the tile
    const int wn = wave >> 2;             // channel half = ping-pong group
    const int m0 = blockIdx.x * 256;
    const int n0 = blockIdx.y * 320;
    const int Cin = a.Cin;
    const int nchunk = Cin >> 6;
    const int NH = nchunk * 18;           // half-tiles
    const int c15 = lane & 15;
    const int q = lane >> 4;

    if (tid < 8) reinterpret_cast<uint4*>(Z_lds)[tid] = make_uint4(0, 0, 0, 0);

    const char* in_bytes = reinterpret_cast<const char*>(a.in);
    const char* w_blk = reinterpret_cast<const char*>(a.w) + (size_t)blockIdx.y * (2 * WH_BYTES) + lane * 16;
    const size_t w_kt_stride = (size_t)gridDim.y * (2 * WH_BYTES);

    auto issue_A_piece = [&](int chunk, int qq) __attribute__((always_inline)) {
        const int p = 8 * qq + (lane >> 3);             // 1-KiB piece: squares 8qq..8qq+7 of the 256-row tile
        const int cl = lane & 7;                        // LDS 16-byte chunk this lane fills
        const char* src = in_bytes + ((size_t)(m0 + p) * Cin + (size_t)chunk * 64) * 2 + 16 * (cl ^ ((p >> 1) & 7));
        p16_glds16(src, A_lds + (chunk & 1) * A_BYTES + qq * 1024);
    };
    auto issue_half = [&](int y) __attribute__((always_inline)) {      // prologue only
        const char* src = w_blk + (size_t)(y >> 1) * w_kt_stride + (size_t)(y & 1) * WH_BYTES;
        char* dst = W_lds + (y & 3) * WH_BYTES;
        p16_glds16(src + wave * 1024, dst + wave * 1024);
        p16_glds16(src + (8 + wave) * 1024, dst + (8 + wave) * 1024);
        if (wave < 4) p16_glds16(src + (16 + wave) * 1024, dst + (16 + wave) * 1024);
        else p16_glds16(w_blk, D_lds + (wave - 4) * 1024);  // filler: keeps 3 pieces per wave and half-tile (vmcnt)
    };

    float4v acc[MT][NG];
    static_for<0, MT>([&](auto mi) __attribute__((always_inline)) {
        static_for<0, NG>([&](auto ni) __attribute__((always_inline)) {
            acc[decltype(mi)::value][decltype(ni)::value] = float4v{0.f, 0.f, 0.f, 0.f};
        });
    });

    // per-lane constants of the fragment reads
    const int wfx = ((4 - ((c15 >> 2) & 3)) & 3) ^ q;                   // weight rows (64 B): chunk q ^ swizzle key
    const int wrow_off = (wn * 160 + c15) * 64;
    int prow[MT], py[MT], px[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        prow[mi] = wm * 64 + mi * 16 + c15;
        py[mi] = (prow[mi] >> 3) & 7;
        px[mi] = prow[mi] & 7;
    }

    // ---- prologue: chunk 0 activations, half-tiles 0..2 ----
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_A_piece(0, wave * 4 + i);
    issue_half(0);
    issue_half(1);
    issue_half(2);
    // only the activations and half-tile 0 have to be there for the first load section; half-tiles 1 and 2 (6 pieces per wave)
    // stay in flight and are retired by the loop's own counted waits, exactly as in the steady state.  Raw barrier: a
    // __syncthreads() here would drain the DMA queue (vmcnt(0)).
    asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wn == 1) __builtin_amdgcn_s_barrier();          // group 1 runs one barrier behind group 0

    // ---- steady-state DMA (wave-uniform state, advanced incrementally; see tools/ubench/conv_pp.hip) ----
    const uint32_t w_lane = (uint32_t)lane * 16u;
    const uint32_t a_lane = (uint32_t)(lane >> 3) * (uint32_t)Cin * 2u +
                            16u * (uint32_t)((lane & 7) ^ (((wave & 1) << 2) | (lane >> 4)));   // piece q = 4 xi + wave - 4
    const char* w_base = reinterpret_cast<const char*>(a.w) + (size_t)blockIdx.y * (2 * WH_BYTES) + wave * 1024;  // uniform
    int t_next = 3;
    const char* w_ptr = w_base + (size_t)1 * w_kt_stride + WH_BYTES;     // half-tile 3 = K-tile 1, half 1
    int w_slot = 3 * WH_BYTES;
    const char* a_ptr = in_bytes;
    int a_dst = 0, a_left = 0;
    auto issue_next = [&](auto G_) __attribute__((always_inline)) {
        constexpr int G = decltype(G_)::value;
        char* dst = W_lds + w_slot + wave * 1024;
        p16_glds16(w_ptr + w_lane, dst);
        p16_glds16(w_ptr + 8192 + w_lane, dst + 8192);
        if constexpr (G == 0) {
            p16_glds16(w_ptr + 16384 + w_lane, dst + 16384);
        } else {
            const bool have = a_left > 0;
            p16_glds16((have ? a_ptr : in_bytes) + a_lane, have ? A_lds + a_dst : D_lds + (wave - 4) * 1024);
            a_ptr += have ? (size_t)32 * Cin * 2 : 0;
            a_dst += have ? 4096 : 0;
            a_left -= have ? 1 : 0;
        }
        const bool more = t_next + 1 < NH;
        const size_t inc = (t_next & 1) ? (w_kt_stride - WH_BYTES) : (size_t)WH_BYTES;   // odd -> even: next K-tile
        w_ptr += more ? inc : 0;
        t_next += 1;
        w_slot = (t_next & 3) * WH_BYTES;
    };

    const char* abase[MT] = {Z_lds, Z_lds, Z_lds, Z_lds};
    int afx[MT] = {0, 0, 0, 0};
    int y = 0;
#ifdef SW_STAMP
    const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    auto main_loop = [&](auto G_) __attribute__((always_inline)) {
    constexpr int G = decltype(G_)::value;
#pragma unroll 1
    for (int c = 0; c < nchunk; ++c) {
        const char* Ab = A_lds + (c & 1) * A_BYTES;
        if constexpr (G == 1) {                          // this wave's 8 pieces of the next chunk's activations
            a_left = c + 1 < nchunk ? 8 : 0;
            a_ptr = in_bytes + ((size_t)(m0 + 8 * (wave - 4)) * Cin + (size_t)(c + 1) * 64) * 2;
            a_dst = ((c + 1) & 1) * A_BYTES + (wave - 4) * 1024;
        }
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            {   // fragment addresses of this tap (zero square outside the board)
                const int t3 = tap / 3;
                const int dy = t3 - 1, dx = tap - t3 * 3 - 1;
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) {
                    const int yy = py[mi] + dy, xx = px[mi] + dx;
                    const bool ok = (unsigned)yy < 8u && (unsigned)xx < 8u;
                    const int pp = prow[mi] + dy * 8 + dx;
                    abase[mi] = ok ? Ab + pp * 128 : Z_lds;
                    afx[mi] = ok ? ((pp >> 1) & 7) : 0;
                }
            }
            static_for<0, 2>([&](auto h_) __attribute__((always_inline)) {
                constexpr int h = decltype(h_)::value;
                const int yh = y + h;
                half8 fa[MT], fb[NG];
                const char* Wb = W_lds + (yh & 3) * WH_BYTES + wrow_off + 16 * wfx;
                static_for<0, MT>([&](auto mi_) __attribute__((always_inline)) {
                    constexpr int mi = decltype(mi_)::value;
                    fa[mi] = *reinterpret_cast<const half8*>(abase[mi] + 16 * (afx[mi] ^ (4 * h + q)));
                });
                static_for<0, NG>([&](auto ni_) __attribute__((always_inline)) {
                    constexpr int ni = decltype(ni_)::value;
                    fb[ni] = *reinterpret_cast<const half8*>(Wb + ni * 1024);
                });
                P16_FENCE(); issue_next(G_); P16_FENCE();      // in the load section, not between the MFMAs (conv_zs.hip)
                asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                P16_FENCE();
                __builtin_amdgcn_s_barrier();
                P16_FENCE();
                __builtin_amdgcn_s_setprio(1);
                static_for<0, NG>([&](auto ni_) __attribute__((always_inline)) {
                    constexpr int ni = decltype(ni_)::value;
                    static_for<0, MT>([&](auto mi_) __attribute__((always_inline)) {
                        constexpr int mi = decltype(mi_)::value;
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[mi], fb[ni], acc[mi][ni], 0, 0, 0);
                    });
                });
                __builtin_amdgcn_s_setprio(0);
                P16_FENCE();
                __builtin_amdgcn_s_barrier();
                P16_FENCE();
            });
            y += 2;
        }
    }
    };
    if (wn == 0) main_loop(std::integral_constant<int, 0>{});
    else main_loop(std::integral_constant<int, 1>{});
#ifdef SW_STAMP
    if (tid == 0) {
        unsigned long long* o = g_p16_stamp + (size_t)blockIdx.x * 4;
        o[0] = st_c0; o[1] = st_r0; o[2] = __builtin_amdgcn_s_memtime(); o[3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's tail refetches / fillers have landed
    if (wn == 0) __builtin_amdgcn_s_barrier();          // matches group 1's extra barrier
    __builtin_amdgcn_s_barrier();                       // every wave's DMA has landed before anyone stages output (tools/ubench/conv_pp.hip)

#ifdef SW_STAMP
    if (tid == 0) {
        unsigned long long* o = g_p16_stamp + (size_t)gridDim.x * 4 + (size_t)blockIdx.x * 4 + (size_t)(a.ksplit & 1) * gridDim.x * 8;
        unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        o[0] = st_entry; o[1] = __builtin_amdgcn_s_memrealtime(); o[2] = hw; o[3] = xcc;
    }
#endif
#ifdef PP_NO_EPILOGUE
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NG; ++ni) asm volatile("" :: "v"(acc[mi][ni]));
#else
    if constexpr (EPI == 3) conv_tail_epilogue16<ACT, false>(acc, a, smem, m0, wm, wn, wave, lane);
    else if constexpr (EPI == 5) conv_tail_epilogue16<ACT, true>(acc, a, smem, m0, wm, wn, wave, lane);
    else conv_tile_epilogue16<EPI, ACT>(acc, a, smem + wave * (5 * 64 * 64), m0, n0, wm, wn, lane);
#endif
}

template <int EPI, int ACT>
static hipError_t launch_conv_pp16_e(const GemmArgs& a, hipStream_t st) {
    const size_t lds = 160 * 1024;     // main loop 151,680 B; the epilogue stages the whole tile
    static DeviceOnce once;
    hipError_t e = once.run([] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pp16_kernel<EPI, ACT>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    if (e != hipSuccess) return e;
    dim3 grid(a.Mrows / 256, a.Npad / 320);
    hipLaunchKernelGGL((conv_pp16_kernel<EPI, ACT>), grid, dim3(512), lds, st, a);
    return hipGetLastError();
}

// 3x3 only; a.w must be in the half-tile layout (GemmArgs::w_pp).  Same contract as launch_conv_pp.
hipError_t launch_conv_pp16(const GemmArgs& a, hipStream_t st) {
    if (a.Cin % 64 != 0 || a.Npad % 320 != 0 || a.Mrows % 256 != 0) return hipErrorInvalidValue;
    if (a.mul != nullptr || a.out_f32 != 0) return hipErrorInvalidValue;      // 3x3 convs never use these
    if ((size_t)a.Mrows * a.ldo * 2 >= ((size_t)1 << 32)) return hipErrorInvalidValue;   // 32-bit store offsets
    if (a.res != nullptr) {                     // conv2 of a block with the block's tail fused (conv_tail16.h)
        if (a.N != 320 || a.Npad != 320 || a.ldo != 320 || a.bias != nullptr || a.out_stats != nullptr) return hipErrorInvalidValue;
        if (a.y2 != nullptr && a.gn_gamma == nullptr) return hipErrorInvalidValue;
        if (a.se_w1 != nullptr && (a.se_hidden < 8 || a.se_hidden > 128 || a.se_hidden % 8 != 0 || a.se_w1h == nullptr ||
                                   a.se_w2h == nullptr)) return hipErrorInvalidValue;
        if (a.pre_gamma != nullptr) {               // x += act(norm(conv(x))) (chess-feature conv) + next GroupNorm
            if (a.se_w1 != nullptr) return hipErrorInvalidValue;
            if (a.epi_act == ACT_SILU) return launch_conv_pp16_e<5, ACT_SILU>(a, st);
            if (a.epi_act == ACT_RELU) return launch_conv_pp16_e<5, ACT_RELU>(a, st);
            return hipErrorInvalidValue;
        }
        if (a.epi_act == ACT_SILU) return launch_conv_pp16_e<3, ACT_SILU>(a, st);
        if (a.epi_act == ACT_RELU) return launch_conv_pp16_e<3, ACT_RELU>(a, st);
        return hipErrorInvalidValue;
    }
    if (a.gn_gamma != nullptr) {                // conv1 of a block: GroupNorm + the network activation
        if (a.epi_act == ACT_SILU) return launch_conv_pp16_e<1, ACT_SILU>(a, st);
        if (a.epi_act == ACT_RELU) return launch_conv_pp16_e<1, ACT_RELU>(a, st);
        return hipErrorInvalidValue;
    }
    return a.epi_act == ACT_NONE ? launch_conv_pp16_e<0, ACT_NONE>(a, st) : hipErrorInvalidValue;
}
