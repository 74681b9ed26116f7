// conv_pp_kernel: 3x3 implicit-GEMM conv on MFMA, 256 rows (4 boards) x 320 output channels per workgroup, with the
// two halves of the workgroup running one barrier apart ("ping-pong"): while waves 0-3 issue their 10 MFMAs of a
// 16-deep k slice, waves 4-7 (their SIMD partners) read the next fragments from LDS and issue the weight DMA, and
// vice versa.  Weights stream through a ring of four half-K-tiles (320 x 32 k) filled by global_load_lds three
// half-tiles ahead and retired with a counted s_waitcnt vmcnt (never 0 inside the loop); the activations of a 64-channel
// chunk stay resident in LDS for all nine taps (tap = shifted row address, off-board = one zero square).
//
// Barrier numbering (b = 0,1,...; every wave executes the same count, waves 4-7 start with one extra):
//   waves 0-3: L(y,a) ends at barrier 4y+2a, C(y,a) at 4y+2a+1     y = half-tile, a = k slice within it
//   waves 4-7: L(y,a) ends at barrier 4y+2a+1, C(y,a) at 4y+2a+2
//   L = fragment reads of slice (y,a) [+ DMA issue and vmcnt when a == 1],  C = the slice's 10 MFMAs
// (Past the last half-tile the loop keeps issuing: the last one again, into slots nobody reads, so that every
//  L(y,1) issues exactly three pieces and the counted wait never changes.)
// RAW: half-tile y is first read in L(y,0) of waves 0-3, after barrier 4y-1; every wave retires its DMA pieces of y
//      with vmcnt(6) at the end of L(y-1,1), i.e. before barrier 4y-2 (waves 0-3) / 4y-1 (waves 4-7).
// WAR: the pieces of half-tile y+3 overwrite the slot of y-1 and are issued in L(y,1), after barrier 4y+1 (4y+2);
//      the last reads of y-1 (waves 4-7, L(y-1,1)) are retired by the lgkmcnt wait of C(y-1,1), before barrier 4y.
#include "../../matrix0_amd/csrc/kernel_common.h"
#include "../../matrix0_amd/csrc/conv_epilogue.h"
#include "../../matrix0_amd/csrc/conv_tail.h"

__device__ __forceinline__ void pp_glds16(const void* gsrc, void* lds_wave_base) {
    // LDS destination = wave-uniform base + lane*16 (hardware); the global source is per lane
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

#define PP_FENCE() asm volatile("" ::: "memory")

#if defined(PP_NO_SETPRIO) || defined(PP_LPRIO) || defined(PP_STATIC_PRIO)
#define PP_SETPRIO(x) do {} while (0)
#else
#define PP_SETPRIO(x) __builtin_amdgcn_s_setprio(x)
#endif
#ifdef PP_LPRIO
#define PP_LSETPRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define PP_LSETPRIO(x) do {} while (0)
#endif

// ablation switches of tools/ubench/conv_pp_bench.hip (timing experiments only; results are wrong with them)
#ifdef PP_NO_BARRIER
#define PP_BARRIER() do {} while (0)
#else
#define PP_BARRIER() __builtin_amdgcn_s_barrier()
#endif
#ifndef PP_EARLY_BAR
#define PP_EARLY_BAR 0
#endif
#ifndef PP_M16_ISSUE_AT
#define PP_M16_ISSUE_AT 2
#endif
#ifndef PP_MFMA_REP
#define PP_MFMA_REP 1
#endif

#ifdef PP_TRACE
// cycle stamps of one workgroup's waves 0 and 4 (tools/ubench/conv_pp_bench.hip): [wave>>2][kt-PP_TRACE_KT0][phase][4]
__device__ unsigned long long* g_pp_trace;
#define PP_STAMP(slot)                                                                                   \
    do {                                                                                                 \
        if (trace_on && kt >= PP_TRACE_KT0 && kt < PP_TRACE_KT0 + 4) {                                   \
            const unsigned long long t_ = __builtin_readcyclecounter();                                  \
            if (lane == 0) g_pp_trace[(((wave >> 2) * 4 + (kt - PP_TRACE_KT0)) * 4 + j) * 4 + (slot)] = t_; \
        }                                                                                                \
    } while (0)
#else
#define PP_STAMP(slot) do {} while (0)
#endif

#ifdef SW_STAMP
__device__ unsigned long long* g_pp_stamp;      // [blocks][4]: s_memtime / s_memrealtime at main-loop start and end
#endif

template <int EPI, int ACT>
__global__ __launch_bounds__(512) void conv_pp_kernel(GemmArgs a) {
    constexpr int NT = 5;
    constexpr int A_BYTES = 256 * 128;    // 4 boards x 64 squares x 64 channels fp16, 128-byte rows
    constexpr int WH_BYTES = 320 * 64;    // 320 output channels x 32 k fp16, 64-byte rows
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* A_lds = smem;                   // [2][A_BYTES]   chunk c in buffer c&1
    char* W_lds = smem + 2 * A_BYTES;     // [4][WH_BYTES]  half-tile y in slot y&3
    char* Z_lds = W_lds + 4 * WH_BYTES;   // one all-zero square (128 B)
    char* D_lds = Z_lds + 128;            // [4][1024] sink of the filler DMA pieces

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3;              // board within the tile
    const int wn = wave >> 2;             // N half = ping-pong group
    const int m0 = blockIdx.x * 256;
    const int n0 = blockIdx.y * 320;
    const int Cin = a.Cin;
    const int nchunk = Cin >> 6;
    const int NH = nchunk * 18;           // half-tiles
    const int half = lane >> 5;
    const int r31 = lane & 31;

    if (tid < 8) reinterpret_cast<uint4*>(Z_lds)[tid] = make_uint4(0, 0, 0, 0);

    const char* in_bytes = reinterpret_cast<const char*>(a.in);
    // packed weights: [chunk*9+tap][Npad/320][half][320][32] (pack_gemm), half-tile y of this N block:
    const char* w_blk = reinterpret_cast<const char*>(a.w) + (size_t)blockIdx.y * (2 * WH_BYTES) + lane * 16;
    const size_t w_kt_stride = (size_t)gridDim.y * (2 * WH_BYTES);

    auto issue_A_piece = [&](int chunk, int q) __attribute__((always_inline)) {
        const int p = 8 * q + (lane >> 3);              // 1-KiB piece: squares 8q..8q+7 of the 256-row tile
        const int cl = lane & 7;                        // LDS 16-byte chunk this lane fills
        const char* src = in_bytes + ((size_t)(m0 + p) * Cin + (size_t)chunk * 64) * 2 + 16 * (cl ^ ((p >> 1) & 7));
        pp_glds16(src, A_lds + (chunk & 1) * A_BYTES + q * 1024);
    };
    // the three DMA pieces this wave contributes to half-tile y; `xi`/`c0` = position of the issuing phase
    auto issue_half = [&](int y, int c0, int xi) __attribute__((always_inline)) {
#ifdef PP_SLIM_DMA
        {   // timing experiment: the three pieces with no address arithmetic and no branches (wrong data)
            pp_glds16(w_blk, W_lds + wave * 1024);
            pp_glds16(w_blk, W_lds + (8 + wave) * 1024);
            pp_glds16(w_blk, D_lds + (wave & 3) * 1024);
            return;
        }
#endif
        const char* src = w_blk + (size_t)(y >> 1) * w_kt_stride + (size_t)(y & 1) * WH_BYTES;
        char* dst = W_lds + (y & 3) * WH_BYTES;
        pp_glds16(src + wave * 1024, dst + wave * 1024);
        pp_glds16(src + (8 + wave) * 1024, dst + (8 + wave) * 1024);
        if (wave < 4) {
            pp_glds16(src + (16 + wave) * 1024, dst + (16 + wave) * 1024);
        } else if (xi < 8 && c0 + 1 < nchunk) {
            issue_A_piece(c0 + 1, xi * 4 + (wave - 4));   // next chunk's activations, 32 pieces over 8 half-tiles
        } else {
            pp_glds16(w_blk, D_lds + (wave - 4) * 1024);  // filler: keeps 3 pieces per wave and half-tile (vmcnt)
        }
    };

#ifdef PP_MFMA16
    // timing experiment (tools/ubench/conv_pp_bench.hip): the same fragment reads feeding 40 v_mfma_f32_16x16x32_f16 per
    // half-tile instead of 20 v_mfma_f32_32x32x16_f16 -- same FLOP, same LDS bytes, same accumulator registers (results wrong)
    typedef float float4v __attribute__((ext_vector_type(4)));
    float4v acc16[4][2 * NT];
    static_for<0, 4>([&](auto mi) __attribute__((always_inline)) {
        static_for<0, 2 * NT>([&](auto ni) __attribute__((always_inline)) {
            acc16[decltype(mi)::value][decltype(ni)::value] = float4v{0.f, 0.f, 0.f, 0.f};
        });
    });
#endif
    float16v acc[2][NT];
    static_for<0, 2>([&](auto mi) __attribute__((always_inline)) {
        static_for<0, NT>([&](auto ni) __attribute__((always_inline)) {
            acc[decltype(mi)::value][decltype(ni)::value] = float16v{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f,
                                                                      0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        });
    });

    // per-lane constants of the fragment reads
    const int wfx = ((4 - ((r31 >> 2) & 3)) & 3) ^ half;                // weight rows (64 B): swizzle key (pack_gemm) ^ k-half
    const int wrow_off = (wn * NT * 32 + r31) * 64;
    int prow[2], py[2], px[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        prow[mi] = wm * 64 + mi * 32 + r31;
        py[mi] = (prow[mi] >> 3) & 7;
        px[mi] = prow[mi] & 7;
    }

    // ---- prologue: chunk 0 activations, half-tiles 0..2 ----
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_A_piece(0, wave * 4 + i);
    issue_half(0, nchunk, 99);
    issue_half(1, nchunk, 99);
    issue_half(2, nchunk, 99);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (wn == 1) __builtin_amdgcn_s_barrier();          // group 1 runs one barrier behind group 0
#ifdef PP_STATIC_PRIO
    if (wn == 1) __builtin_amdgcn_s_setprio(1);
#endif

    // ---- steady-state DMA: everything that addresses it is wave-uniform (SGPR) and advanced incrementally; the only
    // per-lane parts are two constant 32-bit offsets (the address arithmetic and branches of the first version of this
    // loop cost 30% of the kernel: 499 -> 340 us in the stand-alone bench when replaced by fixed addresses) ----
    const uint32_t w_lane = (uint32_t)lane * 16u;
    const uint32_t a_lane = (uint32_t)(lane >> 3) * (uint32_t)Cin * 2u +
                            16u * (uint32_t)((lane & 7) ^ (((wave & 1) << 2) | (lane >> 4)));   // piece q = 4 xi + wave - 4
    const char* w_base = reinterpret_cast<const char*>(a.w) + (size_t)blockIdx.y * (2 * WH_BYTES) + wave * 1024;  // uniform
    int t_next = 3;                                      // half-tile the next issue fetches
    const char* w_ptr = w_base + (size_t)1 * w_kt_stride + WH_BYTES;     // half-tile 3 = K-tile 1, half 1
    int w_slot = 3 * WH_BYTES;                           // LDS offset of slot t_next & 3
    const char* a_ptr = in_bytes;                        // group 1: next activation piece of the next chunk
    int a_dst = 0, a_left = 0;
    auto issue_next = [&](auto G_) __attribute__((always_inline)) {
        constexpr int G = decltype(G_)::value;
        char* dst = W_lds + w_slot + wave * 1024;
        pp_glds16(w_ptr + w_lane, dst);
        pp_glds16(w_ptr + 8192 + w_lane, dst + 8192);
        if constexpr (G == 0) {
            pp_glds16(w_ptr + 16384 + w_lane, dst + 16384);
        } else {
            const bool have = a_left > 0;
            pp_glds16((have ? a_ptr : in_bytes) + a_lane, have ? A_lds + a_dst : D_lds + (wave - 4) * 1024);
            a_ptr += have ? (size_t)32 * Cin * 2 : 0;
            a_dst += have ? 4096 : 0;
            a_left -= have ? 1 : 0;
        }
        // advance to the next half-tile; past the end the last one is fetched again into a slot nobody reads
        const bool more = t_next + 1 < NH;
        const size_t inc = (t_next & 1) ? (w_kt_stride - WH_BYTES) : (size_t)WH_BYTES;   // odd -> even: next K-tile
        w_ptr += more ? inc : 0;
        t_next += 1;
        w_slot = (t_next & 3) * WH_BYTES;
    };

    half8 fa0 = {}, fa1 = {}, fb[NT] = {};
    const char* abase[2] = {Z_lds, Z_lds};
    int afx[2] = {0, 0};
    int y = 0;
#ifdef PP_TRACE
    const bool trace_on = blockIdx.x == PP_TRACE_BLOCK && (wave & 3) == 0;
    int kt = 0;
#endif
    // the loop is instantiated once per ping-pong group: the third DMA piece differs (weights / activations)
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
            // fragment addresses of tap `tp` in activation buffer `Abuf` (zero square outside the board)
            auto tap_addr = [&](const char* Abuf, int tp, const char* (&ab)[2], int (&af)[2]) __attribute__((always_inline)) {
                const int t3 = tp / 3;
                const int dy = t3 - 1, dx = tp - t3 * 3 - 1;
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    const int yy = py[mi] + dy, xx = px[mi] + dx;
                    const bool ok = (unsigned)yy < 8u && (unsigned)xx < 8u;
                    const int pp = prow[mi] + dy * 8 + dx;
                    ab[mi] = ok ? Abuf + pp * 128 : Z_lds;
                    af[mi] = ok ? (((pp >> 1) & 7) ^ half) : 0;
                }
            };
#ifdef PP_SLIM_TAP
            if (c == 0 && tap == 0) tap_addr(Ab, 4, abase, afx);   // timing experiment: no per-tap address work (wrong data)
#else
            tap_addr(Ab, tap, abase, afx);
#endif
#ifndef PP_KS1
            // One load / compute phase pair per HALF-TILE (two k16 slices: 14 fragment reads, 20 MFMAs): half the barriers of
            // the first form (one pair per k16 slice, -DPP_KS1: 457 vs 446 us stand-alone, forward 26.47 vs 26.23 ms).  The
            // half-tile DMA is issued in the MFMA shadow; its counted wait sits at the end of the load section BEFORE the
            // half-tile's own (only the previous issue may still be in flight): both groups pass that wait before either
            // reads the half-tile (group 1's load section ends at the barrier that opens group 0's next one).
            static_for<0, 2>([&](auto h_) __attribute__((always_inline)) {
                constexpr int h = decltype(h_)::value;
                const int yh = y + h;
                half8 ga0[2], ga1[2], gb[2][NT];
                static_for<0, 2>([&](auto u_) __attribute__((always_inline)) {
                    constexpr int u = decltype(u_)::value;
                    constexpr int j = 2 * h + u;
                    const char* Wb = W_lds + (yh & 3) * WH_BYTES + wrow_off + 16 * (wfx ^ (u << 1));
                    ga0[u] = *reinterpret_cast<const half8*>(abase[0] + 16 * (afx[0] ^ (j << 1)));
                    ga1[u] = *reinterpret_cast<const half8*>(abase[1] + 16 * (afx[1] ^ (j << 1)));
                    static_for<0, NT>([&](auto ni_) __attribute__((always_inline)) {
                        constexpr int ni = decltype(ni_)::value;
                        gb[u][ni] = *reinterpret_cast<const half8*>(Wb + ni * 2048);
                    });
                });
                asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                PP_FENCE();
                PP_BARRIER();
                PP_FENCE();
                PP_SETPRIO(1);
#ifdef PP_MFMA16
                static_for<0, 2 * NT>([&](auto n_) __attribute__((always_inline)) {
                    constexpr int n = decltype(n_)::value;
                    const half8 bfrag = gb[n / NT][n % NT];
                    acc16[0][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ga0[0], bfrag, acc16[0][n], 0, 0, 0);
                    acc16[1][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ga0[1], bfrag, acc16[1][n], 0, 0, 0);
                    acc16[2][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ga1[0], bfrag, acc16[2][n], 0, 0, 0);
                    acc16[3][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ga1[1], bfrag, acc16[3][n], 0, 0, 0);
                    if constexpr (n == PP_M16_ISSUE_AT) { PP_FENCE(); issue_next(G_); PP_FENCE(); }
#if PP_EARLY_BAR > 0
                    // the phase-end barrier PP_EARLY_BAR MFMA groups before the last MFMA: its release latency then runs under
                    // this wave's remaining MFMAs instead of leaving the matrix pipe idle (MFMAs touch no LDS: every LDS / DMA
                    // ordering the barrier provides is unchanged)
                    if constexpr (n == 2 * NT - 1 - PP_EARLY_BAR) { PP_FENCE(); PP_SETPRIO(0); PP_BARRIER(); PP_FENCE(); }
#endif
                });
#else
                static_for<0, 2>([&](auto u_) __attribute__((always_inline)) {
                    constexpr int u = decltype(u_)::value;
                    static_for<0, NT>([&](auto ni_) __attribute__((always_inline)) {
                        constexpr int ni = decltype(ni_)::value;
                        acc[0][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ga0[u], gb[u][ni], acc[0][ni], 0, 0, 0);
                        acc[1][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ga1[u], gb[u][ni], acc[1][ni], 0, 0, 0);
                        if constexpr (u == 0 && ni == 2) { PP_FENCE(); issue_next(G_); PP_FENCE(); }
#if PP_EARLY_BAR > 0
                        if constexpr (u == 1 && ni == NT - 1 - PP_EARLY_BAR) { PP_FENCE(); PP_SETPRIO(0); PP_BARRIER(); PP_FENCE(); }
#endif
                    });
                });
#endif
#if PP_EARLY_BAR == 0
                PP_SETPRIO(0);
                PP_FENCE();
                PP_BARRIER();
                PP_FENCE();
#endif
            });
#else
            static_for<0, 4>([&](auto j_) __attribute__((always_inline)) {
                constexpr int j = decltype(j_)::value;      // 16-deep k slice of the K-tile
                constexpr int h = j >> 1;
                const int yh = y + h;
                const char* Wb = W_lds + (yh & 3) * WH_BYTES + wrow_off + 16 * (wfx ^ ((j & 1) << 1));
                // ---- L: fragment reads (+ DMA) ----
                PP_STAMP(0);
                PP_LSETPRIO(2);
#ifndef PP_NO_LDSREAD
                fa0 = *reinterpret_cast<const half8*>(abase[0] + 16 * (afx[0] ^ (j << 1)));
                fa1 = *reinterpret_cast<const half8*>(abase[1] + 16 * (afx[1] ^ (j << 1)));
                static_for<0, NT>([&](auto ni_) __attribute__((always_inline)) {
                    constexpr int ni = decltype(ni_)::value;
                    fb[ni] = *reinterpret_cast<const half8*>(Wb + ni * 2048);
                });
#else
                asm volatile("" : "+v"(fa0), "+v"(fa1), "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]), "+v"(fb[4]) : "v"(Wb), "v"(abase[0]), "v"(abase[1]));
#endif
                if constexpr ((j & 1) == 1) {
#ifndef PP_NO_DMA
#ifdef PP_DMA_IN_L
                    issue_next(G_);                     // first placement: in the load section (476 vs 455 us stand-alone)
#endif
#ifndef PP_NO_VMWAIT
                    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
#endif
#endif
                }
#ifdef PP_DUMMY_VALU
                {   // timing experiment: extra VALU work in the load section
                    int dv = lane;
#pragma unroll
                    for (int u = 0; u < PP_DUMMY_VALU; ++u) asm volatile("v_add_u32 %0, %0, 1" : "+v"(dv));
                    asm volatile("" :: "v"(dv));
                }
#endif
                PP_LSETPRIO(0);
                PP_STAMP(1);
                PP_FENCE();
                PP_BARRIER();
                PP_FENCE();
                // ---- C: the slice's MFMAs ----
                PP_STAMP(2);
                PP_SETPRIO(1);
                static_for<0, PP_MFMA_REP>([&](auto) __attribute__((always_inline)) {
                static_for<0, NT>([&](auto ni_) __attribute__((always_inline)) {
                    constexpr int ni = decltype(ni_)::value;
                    acc[0][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa0, fb[ni], acc[0][ni], 0, 0, 0);
                    acc[1][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa1, fb[ni], acc[1][ni], 0, 0, 0);
#if !defined(PP_DMA_IN_L) && !defined(PP_NO_DMA)
                    // the next half-tile's DMA is issued in the shadow of the even slice's MFMAs (the issuing wave is
                    // otherwise waiting for the matrix pipe there); its counted wait stays in the odd slice's load
                    // section.  The slot it overwrites (half-tile y-1 resp. y) was last read in a load section that both
                    // groups have left by now.
                    if constexpr ((j & 1) == 0 && ni == 2) { PP_FENCE(); issue_next(G_); PP_FENCE(); }
#endif
                });
                });
                PP_SETPRIO(0);
                PP_STAMP(3);
                PP_FENCE();
                PP_BARRIER();
                PP_FENCE();
            });
#endif
            y += 2;
#ifdef PP_TRACE
            ++kt;
#endif
        }
    }
    };
#ifdef SW_STAMP
    const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (wn == 0) main_loop(std::integral_constant<int, 0>{});
    else main_loop(std::integral_constant<int, 1>{});
#ifdef SW_STAMP
    if (tid == 0) {
        unsigned long long* o = g_pp_stamp + (size_t)blockIdx.x * 4;
        o[0] = st_c0; o[1] = st_r0; o[2] = __builtin_amdgcn_s_memtime(); o[3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's tail refetches / fillers have landed
    if (wn == 0) __builtin_amdgcn_s_barrier();          // matches group 1's extra barrier
    // One more barrier, with every wave's vmcnt(0) before it: group 1 reaches its wait only AFTER the barrier that
    // releases group 0 above, so without this group 0 could stage its output tile while group 1's last DMA pieces are
    // still on their way into the same LDS bytes.
    __builtin_amdgcn_s_barrier();

#ifdef PP_MFMA16
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2 * NT; ++ni) asm volatile("" :: "v"(acc16[mi][ni]));
#endif
#ifdef PP_NO_EPILOGUE
    asm volatile("" :: "v"(acc[0][0]), "v"(acc[0][1]), "v"(acc[0][2]), "v"(acc[0][3]), "v"(acc[0][4]));
    asm volatile("" :: "v"(acc[1][0]), "v"(acc[1][1]), "v"(acc[1][2]), "v"(acc[1][3]), "v"(acc[1][4]));
#else
    // every wave is past its last LDS read and DMA wait here (the barrier just above / the loop's final one)
    if constexpr (EPI == 3) conv_tail_epilogue<ACT, false>(acc, a, smem, m0, wm, wn, wave, lane);
    else if constexpr (EPI == 5) conv_tail_epilogue<ACT, true>(acc, a, smem, m0, wm, wn, wave, lane);
    else conv_tile_epilogue<EPI, ACT, NT>(acc, a, smem + wave * (NT * 64 * 64), m0, n0, wm, wn, lane);
#endif
}

template <int EPI, int ACT>
static hipError_t launch_conv_pp_e(const GemmArgs& a, hipStream_t st) {
    const size_t lds = 160 * 1024;     // main loop 151,680 B; the epilogue stages the whole 256 x 320 fp16 tile
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_pp_kernel<EPI, ACT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid(a.Mrows / 256, a.Npad / 320);
    hipLaunchKernelGGL((conv_pp_kernel<EPI, ACT>), grid, dim3(512), lds, st, a);
    return hipGetLastError();
}

// 3x3 only; a.w must be in the half-tile layout (GemmArgs::w_pp).
hipError_t launch_conv_pp(const GemmArgs& a, hipStream_t st) {
    if (a.Cin % 64 != 0 || a.Npad % 320 != 0 || a.Mrows % 256 != 0) return hipErrorInvalidValue;
    if (a.mul != nullptr || a.out_f32 != 0) return hipErrorInvalidValue;      // 3x3 convs never use these
    if ((size_t)a.Mrows * a.ldo * 2 >= ((size_t)1 << 32)) return hipErrorInvalidValue;   // 32-bit store offsets
    if (a.res != nullptr) {                     // conv2 of a block with the block's tail fused (conv_tail.h)
        if (a.N != 320 || a.Npad != 320 || a.ldo != 320 || a.bias != nullptr || a.out_stats != nullptr) return hipErrorInvalidValue;
        if (a.y2 != nullptr && a.gn_gamma == nullptr) return hipErrorInvalidValue;
        if (a.se_w1 != nullptr && (a.se_hidden < 4 || a.se_hidden > 128 || a.se_hidden % 4 != 0)) return hipErrorInvalidValue;
        if (a.pre_gamma != nullptr) {               // x += act(norm(conv(x))) (chess-feature conv) + next GroupNorm
            if (a.se_w1 != nullptr) return hipErrorInvalidValue;
            if (a.epi_act == ACT_SILU) return launch_conv_pp_e<5, ACT_SILU>(a, st);
            if (a.epi_act == ACT_RELU) return launch_conv_pp_e<5, ACT_RELU>(a, st);
            return hipErrorInvalidValue;
        }
        if (a.epi_act == ACT_SILU) return launch_conv_pp_e<3, ACT_SILU>(a, st);
        if (a.epi_act == ACT_RELU) return launch_conv_pp_e<3, ACT_RELU>(a, st);
        return hipErrorInvalidValue;
    }
    if (a.gn_gamma != nullptr) {                // conv1 of a block: GroupNorm + the network activation
        if (a.epi_act == ACT_SILU) return launch_conv_pp_e<1, ACT_SILU>(a, st);
        if (a.epi_act == ACT_RELU) return launch_conv_pp_e<1, ACT_RELU>(a, st);
        return hipErrorInvalidValue;
    }
    return a.epi_act == ACT_NONE ? launch_conv_pp_e<0, ACT_NONE>(a, st) : hipErrorInvalidValue;
}
