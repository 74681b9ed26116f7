// conv_zs_kernel: the 3x3 320->320 implicit-GEMM conv of the tower with the zero padding SKIPPED.
//
// conv_pp16_kernel (read its header and tools/ubench/conv_pp.hip's: same workgroup tile of 4 boards x 320 channels, LDS images, DMA ring,
// ping-pong groups and barrier protocol) gives a wave one board x 160 channels, M-tile = 16 consecutive squares.  A 3x3 conv
// on an 8x8 board multiplies 92 of its 576 (square, tap) pairs by the zero padding; the main loop is POWER-bound (DESIGN.md
// section 5), so those MFMAs cost wall time.  Here a wave owns TWO boards x 80 channels and M-tile mi = board row y = mi of
// both boards (lanes c15 < 8: board a, squares 8 mi .. 8 mi + 7; c15 >= 8: board b).  For the three taps with dy = -1 the
// operand of tile 0 is the padding row above both boards, for dy = +1 that of tile 7 the row below: those tiles' MFMAs and
// fragment reads are simply not issued -- 6 of the 72 (tap, tile) pairs, 8.3 % of the matrix work, results bit-identical
// (the skipped products are exact zeros).  Only the left / right padding column (one lane in eight) is still read from a zero
// region.  Per half-tile (32 k) a wave reads 8 (7) activation fragments and 5 weight fragments and issues 40 (35) MFMAs.
//
// LDS activation image: [256 rows = board * 64 + square][64 channels] fp16, 128-byte rows, 16-byte chunk index XOR
// key(row) = (column & 2) | (board parity << 2).  A ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19,
// 28-31}, {32-35, 44-47, 52-59}, {36-43, 48-51, 60-63} (MI355X_MICROARCH.md, LDS): columns 0-3 of one board and 4-7 of the other with
// k-chunk q, the other halves with q ^ 1.  With this key the 16 lanes of every group touch 16 different bank quads for all
// three column shifts of a tap row (searched exhaustively; round 2's key ((column >> 1) & 3) was built for lanes 0-15 as one group
// and was 2-way conflicted for dx = -1 / +1: 31 % of the main loop's LDS cycles).  key does not change with the board row, so
// the 8 tiles of a tap are ONE base address plus immediate offsets mi * 1024.
#include "kernel_common.h"
#include "conv_zs_epilogue.h"
#include "conv_zs_tail.h"

namespace {
constexpr int ZS_A_BYTES = 256 * 128;                 // 4 boards x 64 squares x 64 channels fp16
constexpr int ZS_WH_BYTES = 320 * 64;                 // 320 output channels x 32 k fp16, 64-byte rows
constexpr int ZS_OFF_W = 2 * ZS_A_BYTES;              // [4][WH_BYTES]  half-tile y in slot y & 3
constexpr int ZS_OFF_Z = ZS_OFF_W + 4 * ZS_WH_BYTES;  // zero region: 8 squares at 1-KiB stride (the tiles' immediate offsets)
constexpr int ZS_OFF_D = ZS_OFF_Z + 8192;             // [4][1024] sink of the filler DMA pieces
constexpr int ZS_LDS_MAIN = ZS_OFF_D + 4096;          // 159,744 B
}

__device__ __forceinline__ void zs_glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// activation pieces: every byte is read once, by one workgroup.  ZS_A_AUX = 2 (nt) asks the caches not to keep them, so that the
// 1.84 MB of weights every workgroup of an XCD streams stay in that XCD's 4 MB L2 instead of being evicted by the ~10 MB of
// activations a round of 32 workgroups moves through it (measured: profiles/r04_exp_conv_activation_loads_nt.log)
#ifndef ZS_A_AUX
#define ZS_A_AUX 0
#endif
__device__ __forceinline__ void zs_glds16_act(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, ZS_A_AUX);
}
#define ZS_FENCE() asm volatile("" ::: "memory")

#ifdef SW_STAMP
__device__ unsigned long long* g_zs_stamp;      // [blocks][4]: s_memtime / s_memrealtime at main-loop start and end
#endif

template <int EPI, int ACT>
__global__ __launch_bounds__(512) void conv_zs_kernel(GemmArgs a) {
    constexpr int NG = 5, MT = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* A_lds = smem;                   // [2][A_BYTES]   chunk c in buffer c & 1
    char* W_lds = smem + ZS_OFF_W;
    char* Z_lds = smem + ZS_OFF_Z;
    char* D_lds = smem + ZS_OFF_D;

#ifdef SW_STAMP
    const unsigned long long st_entry = __builtin_amdgcn_s_memrealtime();
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 3;              // channel quarter
    const int wp = wave >> 2;             // board pair of the tile = ping-pong group
    const int m0 = blockIdx.x * 256;
    const int n0 = blockIdx.y * 320;
    const int Cin = a.Cin;
    const int nchunk = Cin >> 6;
    const int NH = nchunk * 18;           // half-tiles
    const int c15 = lane & 15;
    const int q = lane >> 4;

    reinterpret_cast<uint4*>(Z_lds)[tid] = make_uint4(0, 0, 0, 0);          // 512 x 16 B = the whole zero region

    const char* in_bytes = reinterpret_cast<const char*>(a.in);
    const char* w_blk = reinterpret_cast<const char*>(a.w) + (size_t)blockIdx.y * (2 * ZS_WH_BYTES) + lane * 16;
    const size_t w_kt_stride = (size_t)gridDim.y * (2 * ZS_WH_BYTES);

    auto issue_A_piece = [&](int chunk, int qq) __attribute__((always_inline)) {
        const int p = 8 * qq + (lane >> 3);             // 1-KiB piece: rows 8qq..8qq+7 of the 256-row tile
        const int key = (p & 2) | (((p >> 6) & 1) << 2);
        const char* src = in_bytes + ((size_t)(m0 + p) * Cin + (size_t)chunk * 64) * 2 + 16 * ((lane & 7) ^ key);
        zs_glds16_act(src, A_lds + (chunk & 1) * ZS_A_BYTES + qq * 1024);
    };
    auto issue_half = [&](int y) __attribute__((always_inline)) {      // prologue only
        const char* src = w_blk + (size_t)(y >> 1) * w_kt_stride + (size_t)(y & 1) * ZS_WH_BYTES;
        char* dst = W_lds + (y & 3) * ZS_WH_BYTES;
        zs_glds16(src + wave * 1024, dst + wave * 1024);
        zs_glds16(src + (8 + wave) * 1024, dst + (8 + wave) * 1024);
        if (wave < 4) zs_glds16(src + (16 + wave) * 1024, dst + (16 + wave) * 1024);
        else zs_glds16(w_blk, D_lds + (wave - 4) * 1024);   // filler: keeps 3 pieces per wave and half-tile (vmcnt)
    };

    float4v acc[MT][NG];
    static_for<0, MT>([&](auto mi) __attribute__((always_inline)) {
        static_for<0, NG>([&](auto ni) __attribute__((always_inline)) {
            acc[decltype(mi)::value][decltype(ni)::value] = float4v{0.f, 0.f, 0.f, 0.f};
        });
    });

    // per-lane constants of the fragment reads
    const int wfx = ((4 - ((c15 >> 2) & 3)) & 3) ^ q;                   // weight rows (64 B): chunk q ^ swizzle key
    const int wrow_off = (wn * 80 + c15) * 64;
    const int lx = c15 & 7;                                             // board column of this lane's A-operand rows
    const int lb = c15 >> 3;                                            // board of the pair
    const int arow0 = (2 * wp + lb) * 64 + lx;                          // tile row of (board, y = 0, x)

    // ---- prologue: chunk 0 activations, half-tiles 0..2 ----
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_A_piece(0, wave * 4 + i);
    issue_half(0);
    issue_half(1);
    issue_half(2);
    // only the activations and half-tile 0 have to be there for the first load section; half-tiles 1 and 2 (6 pieces per wave)
    // stay in flight and are retired by the loop's own counted waits.  Raw barrier: __syncthreads() would drain the DMA queue.
    asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wp == 1) __builtin_amdgcn_s_barrier();          // group 1 runs one barrier behind group 0

    // ---- steady-state DMA (wave-uniform state, advanced incrementally; see tools/ubench/conv_pp.hip) ----
    const uint32_t w_lane = (uint32_t)lane * 16u;
    // activation piece qq = 4 xi + wave - 4 (xi = 0..7): rows 32 xi + 8 (wave - 4) + (lane >> 3); key = ((lane >> 3) & 2) | (xi & 2) << 1
    const uint32_t a_lane0 = (uint32_t)(lane >> 3) * (uint32_t)Cin * 2u + 16u * (uint32_t)((lane & 7) ^ ((lane >> 3) & 2));
    const uint32_t a_lane1 = (uint32_t)(lane >> 3) * (uint32_t)Cin * 2u + 16u * (uint32_t)((lane & 7) ^ (((lane >> 3) & 2) | 4));
    const char* w_base = reinterpret_cast<const char*>(a.w) + (size_t)blockIdx.y * (2 * ZS_WH_BYTES) + wave * 1024;  // uniform
    int t_next = 3;
    const char* w_ptr = w_base + (size_t)1 * w_kt_stride + ZS_WH_BYTES;     // half-tile 3 = K-tile 1, half 1
    int w_slot = 3 * ZS_WH_BYTES;
    const char* a_ptr = in_bytes;
    int a_dst = 0, a_left = 0;
    auto issue_next = [&](auto G_) __attribute__((always_inline)) {
        constexpr int G = decltype(G_)::value;
        char* dst = W_lds + w_slot + wave * 1024;
#ifdef ZS_EMU8          // timing experiment: the DMA volume of an 8-board x 160-channel tile (half the weight pieces, twice the activation
                        // pieces): group 0 two weight pieces, group 1 its activation piece (+ one weight piece with ZS_EMU8 = 2); results wrong
        if constexpr (G == 0) { zs_glds16(w_ptr + w_lane, dst); zs_glds16(w_ptr + 8192 + w_lane, dst + 8192); }
        else if (ZS_EMU8 == 2) zs_glds16(w_ptr + w_lane, dst);
#else
        zs_glds16(w_ptr + w_lane, dst);
        zs_glds16(w_ptr + 8192 + w_lane, dst + 8192);
#endif
#if defined(ZS_SKIP_DMA) || defined(ZS_EMU8)      // timing experiments: fewer DMA pieces per wave and half-tile (results wrong)
        if constexpr (G == 2) {
#else
        if constexpr (G == 0) {
#endif
            zs_glds16(w_ptr + 16384 + w_lane, dst + 16384);
#if defined(ZS_SKIP_DMA)
        } else if constexpr (G == 3) {
#elif defined(ZS_EMU8)
        } else if constexpr (G == 1) {
#else
        } else {
#endif
            const bool have = a_left > 0;
            const uint32_t al = ((8 - a_left) & 2) ? a_lane1 : a_lane0;     // pieces 2, 3, 6, 7 of this wave: odd boards
            zs_glds16_act((have ? a_ptr : in_bytes) + al, have ? A_lds + a_dst : D_lds + (wave - 4) * 1024);
            a_ptr += have ? (size_t)32 * Cin * 2 : 0;
            a_dst += have ? 4096 : 0;
            a_left -= have ? 1 : 0;
        }
        const bool more = t_next + 1 < NH;
        const size_t inc = (t_next & 1) ? (w_kt_stride - ZS_WH_BYTES) : (size_t)ZS_WH_BYTES;   // odd -> even: next K-tile
#ifndef ZS_W_SAME      // timing experiment: every half-tile re-reads the same 20 KB of weights (L1 / L2 hits; results wrong)
        w_ptr += more ? inc : 0;
#else
        (void)more; (void)inc;
#endif
        t_next += 1;
        w_slot = (t_next & 3) * ZS_WH_BYTES;
    };

    int y = 0;
#ifdef SW_STAMP
    const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    auto main_loop = [&](auto G_) __attribute__((always_inline)) {
    constexpr int G = decltype(G_)::value;
#pragma unroll 1
    for (int c = 0; c < nchunk; ++c) {
        const char* Ab = A_lds + (c & 1) * ZS_A_BYTES;
        if constexpr (G == 1) {                          // this wave's 8 pieces of the next chunk's activations
            a_left = c + 1 < nchunk ? 8 : 0;
            a_ptr = in_bytes + ((size_t)(m0 + 8 * (wave - 4)) * Cin + (size_t)(c + 1) * 64) * 2;
            a_dst = ((c + 1) & 1) * ZS_A_BYTES + (wave - 4) * 1024;
        }
        static_for<0, 3>([&](auto t3_) __attribute__((always_inline)) {
            constexpr int t3 = decltype(t3_)::value;     // dy = t3 - 1
            constexpr int LO = t3 == 0 ? 1 : 0;          // first / one-past-last M-tile with an operand on the board
            constexpr int HI = t3 == 2 ? 7 : 8;
            // row of tile LO's operand for dx = -1:  dy = -1: tile 1 reads board row 0;  dy = 0: tile 0 row 0;  dy = +1: tile 0 row 1
            const int rbase = arow0 + (t3 == 2 ? 8 : 0) - 1;
#pragma unroll 1
            for (int dxi = 0; dxi < 3; ++dxi) {
                const int xx = lx + dxi - 1;
                const bool ok = (unsigned)xx < 8u;
                // a lane beside the board reads zeros at the bank slot the wrapped column (x + 8 / x - 8) would have used: the 16
                // lanes of a read group then touch 16 different slots for every tap
                const int key = (xx & 2) | (lb << 2);
                const char* rowp = ok ? Ab + (rbase + dxi) * 128 : Z_lds + (xx & 1) * 128;
                const char* ap0 = rowp + 16 * (key ^ q);
                const char* ap1 = rowp + 16 * (key ^ q ^ 4);
                static_for<0, 2>([&](auto h_) __attribute__((always_inline)) {
                    constexpr int h = decltype(h_)::value;
                    const int yh = y + h;
                    const char* ap = h ? ap1 : ap0;
                    half8 fa[MT], fb[NG];
                    const char* Wb = W_lds + (yh & 3) * ZS_WH_BYTES + wrow_off + 16 * wfx;
                    static_for<LO, HI>([&](auto mi_) __attribute__((always_inline)) {
                        constexpr int mi = decltype(mi_)::value;
                        fa[mi] = *reinterpret_cast<const half8*>(ap + (mi - LO) * 1024);
                    });
                    static_for<0, NG>([&](auto ni_) __attribute__((always_inline)) {
                        constexpr int ni = decltype(ni_)::value;
                        fb[ni] = *reinterpret_cast<const half8*>(Wb + ni * 1024);
                    });
                    // This wave's share of the half-tile three ahead goes out HERE, in the load section (which has ~300 cycles of slack
                    // under the partner group's 40 MFMAs), not between this wave's own MFMAs: an LDS-DMA instruction takes 40-60
                    // cycles to issue, and in the compute section the SIMD's matrix pipe idles for them (three per half-tile: the
                    // main loop was 127.4 k cycles per tile with the DMA after the second channel tile's MFMAs, 113.4 k here; the
                    // chip gives part of it back as clock, 1.81 -> 1.71 GHz: -4 % wall).  The slot's previous tenant (the
                    // half-tile before this one) was read into registers by both wave groups at least a phase ago.
                    ZS_FENCE(); issue_next(G_); ZS_FENCE();
                    // all but this wave's three youngest pieces (the ones just issued) have landed
#if defined(ZS_EMU8)
                    if constexpr (G == 0 || ZS_EMU8 == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
#elif defined(ZS_SKIP_DMA)
                    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
#else
                    asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
#endif
                    ZS_FENCE();
                    __builtin_amdgcn_s_barrier();
                    ZS_FENCE();
                    __builtin_amdgcn_s_setprio(1);
                    static_for<0, NG>([&](auto ni_) __attribute__((always_inline)) {
                        constexpr int ni = decltype(ni_)::value;
                        static_for<LO, HI>([&](auto mi_) __attribute__((always_inline)) {
                            constexpr int mi = decltype(mi_)::value;
                            // operand order = accumulator layout (conv_zs_epilogue.h): weights as A for the plain / GroupNorm epilogues
                            // (a lane gets 4 consecutive channels of a square), activations as A for the fused tail
                            if constexpr (EPI == 3 || EPI == 5)
                                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[mi], fb[ni], acc[mi][ni], 0, 0, 0);
                            else
                                acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[ni], fa[mi], acc[mi][ni], 0, 0, 0);
                        });
                    });
                    __builtin_amdgcn_s_setprio(0);
                    ZS_FENCE();
                    __builtin_amdgcn_s_barrier();
                    ZS_FENCE();
                });
                y += 2;
            }
        });
    }
    };
    if (wp == 0) main_loop(std::integral_constant<int, 0>{});
    else main_loop(std::integral_constant<int, 1>{});
#ifdef SW_STAMP
    if (tid == 0) {
        unsigned long long* o = g_zs_stamp + (size_t)blockIdx.x * 4;
        o[0] = st_c0; o[1] = st_r0; o[2] = __builtin_amdgcn_s_memtime(); o[3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's tail refetches / fillers have landed
    if (wp == 0) __builtin_amdgcn_s_barrier();          // matches group 1's extra barrier
    __builtin_amdgcn_s_barrier();                       // every wave's DMA has landed before anyone stages output (tools/ubench/conv_pp.hip)

#ifdef ZS_PREFETCH
    // L2 prefetch of the first activation chunk of the tile that the XCD's NEXT round of workgroups brings (workgroup ids go to the
    // XCDs round-robin, so tile blockIdx.x + 256 k lands on this XCD): one 4-byte load per row = one per 128-byte line of the
    // chunk, issued before the epilogue, retired (by the compiler's own wait) at the very end.
    float zs_pf = 0.f;
    {
        const int nrow = m0 + 256 * ZS_PREFETCH + (tid & 255);
        if (tid < 256 && nrow < a.Mrows)
            zs_pf = *reinterpret_cast<const volatile float*>(in_bytes + (size_t)nrow * Cin * 2);
    }
#endif
#ifdef PP_NO_EPILOGUE
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NG; ++ni) asm volatile("" :: "v"(acc[mi][ni]));
#else
    if constexpr (EPI == 3) zs_tail_epilogue<ACT, false>(acc, a, smem, m0, wp, wn, wave, lane);
    else if constexpr (EPI == 5) zs_tail_epilogue<ACT, true>(acc, a, smem, m0, wp, wn, wave, lane);
    else zs_tile_epilogue<EPI, ACT>(acc, a, smem + wave * 20480, m0, n0, wp, wn, lane);
#endif
#ifdef ZS_PREFETCH
    asm volatile("" :: "v"(zs_pf));
#endif
#ifdef SW_STAMP     // timeline of the workgroup (10-ns ticks, wave 0) and where it ran: [blocks][4] behind the main-loop stamps
    if (tid == 0) {
        unsigned long long* o = g_zs_stamp + (size_t)gridDim.x * 4 + (size_t)blockIdx.x * 4;
        unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        o[0] = st_entry; o[1] = __builtin_amdgcn_s_memrealtime(); o[2] = hw; o[3] = xcc;
    }
#endif
}

template <int EPI, int ACT>
static hipError_t launch_conv_zs_e(const GemmArgs& a, hipStream_t st) {
    const size_t lds = 160 * 1024;     // main loop 159,744 B; the epilogue stages the whole tile (8 x 20 KiB)
    static DeviceOnce once;             // the attribute is per device: a process may drive several GPUs (arena, tests)
    hipError_t e = once.run([] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_zs_kernel<EPI, ACT>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    if (e != hipSuccess) return e;
    dim3 grid(a.Mrows / 256, a.Npad / 320);
    hipLaunchKernelGGL((conv_zs_kernel<EPI, ACT>), grid, dim3(512), lds, st, a);
    return hipGetLastError();
}

// true: launch_conv_zs takes these arguments (the dispatcher falls back to conv_pp16_kernel otherwise)
bool conv_zs_supports(const GemmArgs& a) {
    if (a.res != nullptr && a.se_w1 != nullptr && (a.se_wf == nullptr || a.se_hidden > ZS_SE_HMAX)) return false;   // fragment-order weights
    return true;
}

// 3x3 only; a.w must be in the half-tile layout (GemmArgs::w_pp).  Same contract as launch_conv_pp16.
hipError_t launch_conv_zs(const GemmArgs& a, hipStream_t st) {
    if (a.Cin % 64 != 0 || a.Npad % 320 != 0 || a.Mrows % 256 != 0) return hipErrorInvalidValue;
    if (a.mul != nullptr || a.out_f32 != 0 || !conv_zs_supports(a)) return hipErrorInvalidValue;
    if ((size_t)a.Mrows * a.ldo * 2 >= ((size_t)1 << 32)) return hipErrorInvalidValue;   // 32-bit store offsets
    if (a.res != nullptr) {                     // conv2 of a block with the block's tail fused (conv_zs_tail.h)
        if (a.N != 320 || a.Npad != 320 || a.ldo != 320 || a.bias != nullptr || a.out_stats != nullptr) return hipErrorInvalidValue;
        if (a.y2 != nullptr && a.gn_gamma == nullptr) return hipErrorInvalidValue;
        if (a.se_w1 != nullptr && (a.se_hidden < 1 || a.se_hidden > ZS_SE_HMAX)) return hipErrorInvalidValue;
        if (a.pre_gamma != nullptr) {               // x += act(norm(conv(x))) (chess-feature conv) + next GroupNorm
            if (a.se_w1 != nullptr) return hipErrorInvalidValue;
            if (a.epi_act == ACT_SILU) return launch_conv_zs_e<5, ACT_SILU>(a, st);
            if (a.epi_act == ACT_RELU) return launch_conv_zs_e<5, ACT_RELU>(a, st);
            return hipErrorInvalidValue;
        }
        if (a.epi_act == ACT_SILU) return launch_conv_zs_e<3, ACT_SILU>(a, st);
        if (a.epi_act == ACT_RELU) return launch_conv_zs_e<3, ACT_RELU>(a, st);
        return hipErrorInvalidValue;
    }
    if (a.gn_gamma != nullptr) {                // conv1 of a block: GroupNorm + the network activation
        if (a.epi_act == ACT_SILU) return launch_conv_zs_e<1, ACT_SILU>(a, st);
        if (a.epi_act == ACT_RELU) return launch_conv_zs_e<1, ACT_RELU>(a, st);
        return hipErrorInvalidValue;
    }
    return a.epi_act == ACT_NONE ? launch_conv_zs_e<0, ACT_NONE>(a, st) : hipErrorInvalidValue;
}
