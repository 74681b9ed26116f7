// attn_block4_kernel: one whole ChessAttention block of the tower (resnet.py:133-181: qkv 1x1 -> per-head scores / softmax / PV ->
// proj 1x1 -> residual add -> LayerNorm) plus the pre-activation GroupNorm of the residual block that follows, in ONE kernel for
// the 320-channel trunk -- round 4: 16 ROLE-SPECIALISED waves per board pair (128 token rows), four per SIMD, 128 registers.
//
// Round 3's kernel (attn_block.hip) ran every phase in all 8 waves at once -- qkv GEMM (matrix pipe + LDS reads), staging, softmax
// (VALU), proj GEMM -- and its ablations showed the parts to be additive.  Here:
//   waves 8-15 ("G", two per SIMD): the qkv GEMM.  A wave owns 16 token rows whose trunk values live in REGISTERS for the whole
//              kernel as the B fragments of the GEMM (10 k-steps x 4 registers; the k order inside a k-step is permuted so that the
//              same registers are, element for element, the residual of the wave's accumulator tiles).  Per period g: the six
//              channel tiles (q, k, v of two heads) of group g+1, one weight piece [16 channels][320 k] = 10 KB each -- ten MFMAs
//              into ONE accumulator tile, converted to fp16 as soon as it is complete and written to the Q / K / V^T staging
//              buffers one barrier later (the A-waves read all their fragments in the first third of their period).
//   waves 0-7  ("A", two per SIMD): the attention of group g, one (board, head, query half) per wave: S^T = K Q^T and
//              O^T = V^T P^T on MFMA 32x32x16, softmax arithmetic in between (relative-position bias requested a period ahead, in
//              registers, from a table pre-arranged in accumulator order); O goes to the [128 tokens][320] O buffer.
//   the end:   proj = O[128 x 320] Wproj^T as a K = 320 GEMM from the O buffer, split over ALL 16 waves (16 tokens x 160 channels
//              each: 40 accumulators; the residual comes from the G-waves' registers, for the A-waves' half it is re-read from
//              global memory), then LayerNorm (the two halves of a token row exchange their sums through LDS), the next block's
//              GroupNorm + activation, outputs through an LDS image in 16-byte stores.
// Synchronisation: ONE barrier per TWO weight pieces (40 intervals per board pair; round 3: 70 barriers + 10 phase barriers).
// The weights are one stream of 80 pieces of 10 KB in consumption order through a 4-slot ring: the pieces of interval i+1 are
// requested (global_load_lds, by the G-waves) right after the barrier of interval i into the slots interval i-1 used, and every
// wave waits with vmcnt(0) before a barrier -- no counted waits, so the kernel_common.h rule holds trivially.
#include "../../matrix0_amd/csrc/kernel_common.h"
#include "../../matrix0_amd/csrc/conv_epilogue.h"

typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));

namespace {
constexpr int A4_PIECE = 10240;
constexpr int A4_NPIECES = 80;                            // 60 qkv (group, tile) + 20 proj (k-step, channel half)
constexpr int A4_NIV = 40;                                // intervals of two pieces
constexpr int A4_O = 0;                                   // [128][640 B], 16-byte chunk ^ (row>>1)&7 within 128 B: first the trunk
                                                          // rows (prologue), then O, then the y / y2 images
constexpr int A4_Q = 81920;                               // [2 heads][128 tokens][16] fp16; a token's two 16-byte halves at
                                                          // half ^ (token >> 3 & 1)
constexpr int A4_K = A4_Q + 8192;                         // same layout
constexpr int A4_VT = A4_K + 8192;                        // [4 units][16][68]
constexpr int A4_VROW = 68;
constexpr int A4_RING = A4_VT + 4 * 16 * A4_VROW * 2;     // 107008: 4 slots; after the main loop: GroupNorm partials
constexpr int A4_PAR = A4_RING + 4 * A4_PIECE;            // 147968: LayerNorm gamma, beta, next GroupNorm gamma, beta [4][320] f32
constexpr int A4_XCH = A4_PAR + 4 * 320 * 4;              // 153088: LayerNorm partial sums [2 channel halves][128 tokens] float2
constexpr int A4_LDS = A4_XCH + 2 * 128 * 8;              // 155136
constexpr int A4_THREADS = 1024;
}

__device__ __forceinline__ void a4_dma16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// 64 bytes per lane from global memory that the compiler does not track (the caller waits: vmcnt(0))
__device__ __forceinline__ void a4_load64(half8& b0, half8& b1, half8& b2, half8& b3, const half8* p) {
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\t"
                 "global_load_dwordx4 %2, %4, off offset:32\n\tglobal_load_dwordx4 %3, %4, off offset:48"
                 : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3) : "v"(p) : "memory");
}
template <int OFF>
__device__ __forceinline__ void a4_load8(half4v& d, const char* p) {
    asm volatile("global_load_dwordx2 %0, %1, off offset:%2" : "=v"(d) : "v"(p), "n"(OFF) : "memory");
}
// one 16-byte LDS read the compiler does not track (the caller waits: a4_arrived)
template <int OFF>
__device__ __forceinline__ void a4_lds16(half8& d, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
// ... and its wait: at most N younger LDS operations outstanding; the operand ties the first use to this point
template <int N>
__device__ __forceinline__ void a4_arrived(half8& f) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void a4_arrived5(half8& f0, half8& f1, half8& f2, half8& f3, half8& f4) {
    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4) : "n"(N) : "memory");
}
template <int CTRL>
__device__ __forceinline__ float a4_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float a4_row_sum(float v) {      // sum over the 16 lanes of a DPP row (every lane gets the total)
    v += a4_dpp<0xB1>(v);
    v += a4_dpp<0x4E>(v);
    v += a4_dpp<0x141>(v);
    v += a4_dpp<0x140>(v);
    return v;
}

#if defined(A4_STAMP) || defined(A4_STAMP2)
__device__ unsigned long long* g_a4_stamp;        // [blocks][2 roles][16] s_memtime stamps (tools/ubench/attn_block4_bench.hip)
#ifdef A4_STAMP
#define A4_ST(role, k) do { if (lane == 0 && (role ? w == 8 : w == 0)) g_a4_stamp[((size_t)blockIdx.x * 2 + role) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define A4_ST(role, k) do {} while (0)
#endif
#else
#define A4_ST(role, k) do {} while (0)
#endif
// A4_DRY: every barrier of the main loop becomes a counter (results are garbage): all 16 waves must report the same count before
// the real kernel is ever launched (a mismatch would hang the workgroup).
#ifdef A4_STAMP2     // arrival at / release from the barriers of intervals 18-23 (period 5 and the start of 6), kept in registers
#define A4_T2(k) do { if (iv >= 18 && iv < 24) ts2[2 * (iv - 18) + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define A4_T2(k) do {} while (0)
#endif
#ifdef A4_DRY
#define A4_BARRIER() do { ++nbar; } while (0)
#else
#define A4_BARRIER() __builtin_amdgcn_s_barrier()
#endif

template <int ACT>
__global__ __launch_bounds__(A4_THREADS) void attn_block4_kernel(AttnBlockArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);             // 0-7: attention waves, 8-15: GEMM waves
    const int l15 = lane & 15, lq = lane >> 4, r31 = lane & 31, half = lane >> 5;
    const size_t b0 = (size_t)blockIdx.x * 2;
    const char* xg = reinterpret_cast<const char*>(a.x) + b0 * 64 * 640;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const uint32_t ring_a = lds0 + A4_RING;
    const int tw = w & 7;                                               // token tile of this wave in the proj and the epilogue
    const int token = 16 * tw + l15;
    const int tsw = (token >> 1) & 7;
    int nbar = 0;
    (void)nbar;
#ifdef A4_STAMP2
    unsigned long long ts2[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    A4_ST(0, 0); A4_ST(1, 0);
#ifndef A4_NO_PRIO
    if (w & 4) __builtin_amdgcn_s_setprio(1);           // waves 4-7 / 12-15: the later-dispatched wave of each role on its SIMD
#endif

    // ---- prologue: the two boards' trunk rows into the O buffer region (80 x 1 KB over the 16 waves)
#pragma unroll
    for (int n = 0; n < 5; ++n) {
        const int idx = w + 16 * n;
        const int q = idx * 64 + lane;
        const int row = q / 40, pos = q - row * 40;
        const int src = (pos & ~7) | ((pos ^ (row >> 1)) & 7);
        a4_dma16(xg + row * 640 + src * 16, smem + A4_O + idx * 1024);
    }
    // proj accumulators of this wave: 16 tokens x 160 channels (G-waves: channels 0-159, A-waves: 160-319)
    float4v oc[10];
    // proj piece: [160 channels][32 k] in 64-byte rows; a ds_read_b128 is served in lane groups {0-3, 12-15, 20-27}, {4-11, 16-19,
    // 28-31}, ...: chunk ^ (4 - quad) & 3 gives the 16 lanes of a group 16 different bank quads
    const uint32_t wpo = (uint32_t)(l15 * 64 + ((lq ^ (4 - (l15 >> 2))) & 3) * 16);
    const uint32_t orow_a = lds0 + A4_O + token * 640;
    // proj pieces: k-step g (the two heads of group g) of this wave's 160 channels, ten MFMAs each.  A run of NPC pieces (all in the
    // ring: the pieces of one interval) is software-pipelined in halves of 5 channel tiles: the fragments of half h + 1 are read
    // while the MFMAs of half h are issued.
    auto proj_run = [&](auto npc_, const uint32_t pb0, const uint32_t pb1, const int g0, const int g1) __attribute__((always_inline)) {
        constexpr int NPC = decltype(npc_)::value, NH = 2 * NPC;
        half8 of[2], pw[2][5];
        auto load = [&](auto h_) __attribute__((always_inline)) {
            constexpr int h = decltype(h_)::value, S = h & 1, pc = h >> 1;
            const uint32_t pbase = (pc ? pb1 : pb0) + wpo;
            if constexpr ((h & 1) == 0) {
                const int c = 4 * (pc ? g1 : g0) + lq;
                a4_lds16<0>(of[pc], orow_a + (uint32_t)(((c & ~7) | ((c ^ tsw) & 7)) * 16));
            }
            static_for<0, 5>([&](auto jj_) __attribute__((always_inline)) {
                constexpr int jj = decltype(jj_)::value;
                a4_lds16<(5 * (h & 1) + jj) * 1024>(pw[S][jj], pbase);
            });
        };
        load(std::integral_constant<int, 0>{});
        static_for<0, NH>([&](auto h_) __attribute__((always_inline)) {
            constexpr int h = decltype(h_)::value, S = h & 1, pc = h >> 1;
            if constexpr (h + 1 < NH) {
                load(std::integral_constant<int, h + 1>{});
                // the fragments of half h are there once at most the reads of half h + 1 are outstanding (5, +1 with its O fragment)
                if constexpr (((h + 1) & 1) == 0) a4_arrived5<6>(pw[S][0], pw[S][1], pw[S][2], pw[S][3], pw[S][4]);
                else a4_arrived5<5>(pw[S][0], pw[S][1], pw[S][2], pw[S][3], pw[S][4]);
            } else {
                a4_arrived5<0>(pw[S][0], pw[S][1], pw[S][2], pw[S][3], pw[S][4]);
            }
            asm volatile("" : "+v"(of[pc]));
            static_for<0, 5>([&](auto jj_) __attribute__((always_inline)) {
                constexpr int jj = decltype(jj_)::value;
                oc[5 * (h & 1) + jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pw[S][jj], of[pc], oc[5 * (h & 1) + jj], 0, 0, 0);
            });
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    auto proj_piece = [&](const uint32_t pbase, const int g) __attribute__((always_inline)) {
        proj_run(std::integral_constant<int, 1>{}, pbase, pbase, g, g);
    };

    if (w < 8) {
        // =====================================================================================================================
        // attention waves: unit (board, head-in-group) = w >> 1, query half = w & 1
        // =====================================================================================================================
        if (tid < 320) {
            float* par = reinterpret_cast<float*>(smem + A4_PAR);
            par[tid] = a.ln_g[tid]; par[320 + tid] = a.ln_b[tid];
            par[640 + tid] = a.y2 ? a.gn2_gamma[tid] : 0.f; par[960 + tid] = a.y2 ? a.gn2_beta[tid] : 0.f;
        }
        const int au = w >> 1, aboard = au >> 1, ahl = au & 1, aqt = w & 1;
        const int aq = aqt * 32 + r31;
        // visibility of key (kt, r) from query aq as a multiplicand, accumulator order: key = kt*32 + 8(r>>2) + 4 half + (r&3)
        half2v visp[16];
        {
            const uint64_t m = a.mask[aq];
            static_for<0, 32>([&](auto i_) __attribute__((always_inline)) {
                constexpr int i = decltype(i_)::value;
                constexpr int kt = i >> 4, r = i & 15;
                const int key = kt * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
                visp[i >> 1][i & 1] = (_Float16)(float)((m >> key) & 1);
            });
        }
        float wm_, wu_;   // output weights of the masked / unmasked branch (resnet.py:154-174)
        if (a.mix > 0.f && a.mix < 1.f) { wm_ = 1.f - a.mix; wu_ = 1.f - (1.f - a.mix); }
        else if (a.mix >= 1.f) { wm_ = 1.f; wu_ = 0.f; }
        else { wm_ = 0.f; wu_ = 1.f; }
        const float isd = a.inv_sqrt_d * 1.44269504088896f;
        const float clampv = 50.f * 1.44269504088896f;
        const float16v zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int hsw = 16 * (half ^ ((r31 >> 3) & 1));                                     // this lane's half of its token's row
        const char* const Kb = smem + A4_K + ahl * 4096 + aboard * 64 * 32;
        const char* const Qp = smem + A4_Q + ahl * 4096 + (aboard * 64 + aq) * 32 + hsw;
        const _Float16* const vrow = reinterpret_cast<const _Float16*>(smem + A4_VT) + (au * 16 + l15) * A4_VROW;
        const int orow = aboard * 64 + aq;
        const uint32_t obase = lds0 + A4_O + orow * 640;
        const int osw = (orow >> 1) & 7;
        int iv = 0;                                                     // interval about to start
#ifdef A4_DMA_A
        // the weight stream is moved by attention waves 0-3 (the waves with the most slack at the barriers: an LDS-DMA instruction
        // costs its wave ~150 cycles of issue): two pieces = 20 x 1 KB, contiguous in the packed stream and in the ring
        const char* wsrc = reinterpret_cast<const char*>(a.wpack) + w * 1024 + lane * 16;
        char* const ring_w = smem + A4_RING + w * 1024;
        auto issue_interval = [&](int k) __attribute__((always_inline)) {
            if (w < 4) {
                const char* s = wsrc + (size_t)k * (2 * A4_PIECE);
                char* d = ring_w + (k & 1) * (2 * A4_PIECE);
                a4_dma16(s, d); a4_dma16(s + 4096, d + 4096); a4_dma16(s + 8192, d + 8192); a4_dma16(s + 12288, d + 12288);
                a4_dma16(s + 16384, d + 16384);
            }
        };
        issue_interval(0);
#endif
        // an interval boundary of these waves: everything they have in flight is done (LDS reads / writes, the bias request) ->
        // barrier.  Returns the ring address of the interval's first piece.
        // (sched_barrier: hipcc moves vector and matrix instructions across s_barrier freely -- without it nearly all of a period's
        // arithmetic ended up in ONE of its three intervals)
        auto sync = [&]() __attribute__((always_inline)) {
            __builtin_amdgcn_sched_barrier(0);
            A4_T2(0);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            A4_BARRIER();
            asm volatile("" ::: "memory");
            A4_T2(1);
            __builtin_amdgcn_sched_barrier(0);
#ifdef A4_DMA_A
            if (iv + 1 < A4_NIV) issue_interval(iv + 1);
#endif
            const uint32_t sb = ring_a + (uint32_t)((iv & 1) * 2 * A4_PIECE);
            ++iv;
            return sb;
        };
        half8 bias8[4];
        // relative-position bias of (head, query half) in accumulator order: 64 B per lane, requested in the last third of the period
        // before (its registers are free once the scores are done).  (inline asm: at the first use of an ordinary load's result
        // hipcc waits wherever that use lands)
        auto bias_request = [&](const int g) __attribute__((always_inline)) {
            const half8* bp = reinterpret_cast<const half8*>(a.bias) + ((size_t)((2 * g + ahl) * 2 + aqt) * 64 + lane) * 4;
            a4_load64(bias8[0], bias8[1], bias8[2], bias8[3], bp);
        };
        // the attention of group g in 3 chunks, an interval boundary in front of each
        auto attend = [&](const int g) __attribute__((always_inline)) {
#ifdef A4_NO_ATTN        // timing experiment: the attention waves only keep the cadence
            sync(); sync(); sync();
            return;
#endif
            float16v st[2];
            float e[2][16];
            half8 vf[2][2];
            sync();
            asm volatile("" : "+v"(bias8[0]), "+v"(bias8[1]), "+v"(bias8[2]), "+v"(bias8[3]) :: "memory");
            {   // chunk 0: ALL fragments of this unit (the G-waves write the next group's Q, K, V^T from the next interval on);
                // S^T = K Q^T; first half of the scores
                const half8 kf0 = *reinterpret_cast<const half8*>(Kb + r31 * 32 + hsw);
                const half8 kf1 = *reinterpret_cast<const half8*>(Kb + (32 + r31) * 32 + hsw);
                const half8 qfr = *reinterpret_cast<const half8*>(Qp);
                static_for<0, 2>([&](auto kt_) __attribute__((always_inline)) {
                    static_for<0, 2>([&](auto jb_) __attribute__((always_inline)) {
                        constexpr int kt = decltype(kt_)::value, jb = decltype(jb_)::value;
                        const half4v lo = *reinterpret_cast<const half4v*>(vrow + kt * 32 + 16 * jb + 4 * half);
                        const half4v hi = *reinterpret_cast<const half4v*>(vrow + kt * 32 + 16 * jb + 8 + 4 * half);
                        vf[kt][jb] = half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    });
                });
                st[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf0, qfr, zero16, 0, 0, 0);
                st[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf1, qfr, zero16, 0, 0, 0);
            }
            float su = 0.f, sm = 0.f;
            auto scores = [&](auto kt_) __attribute__((always_inline)) {
                constexpr int kt = decltype(kt_)::value;
                static_for<0, 16>([&](auto r_) __attribute__((always_inline)) {
                    constexpr int r = decltype(r_)::value;
                    constexpr int bi = kt * 16 + r;
                    float d = st[kt][r] * isd + (float)bias8[bi >> 3][bi & 7];
                    d = __builtin_amdgcn_fmed3f(d, -clampv, clampv);
                    const float eu = __builtin_amdgcn_exp2f(d);
                    e[kt][r] = eu;
                    su += eu;
                    sm += eu * (float)visp[bi >> 1][bi & 1];
                });
            };
            using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
            scores(I0{});
#ifndef A4_CHUNK_A      // all scores in the first interval (the G-waves' lightest: no staging writes), the rest split over the other two
            scores(I1{});
            sync();
#else
            sync();
            scores(I1{});                                               // chunk 1: the other half, the sums, P V of the first 32 keys
#endif
            su += __shfl_xor(su, 32);
            sm += __shfl_xor(sm, 32);
            const float cu = wu_ / su, cm = wm_ / sm;
            float16v oacc = zero16;
            auto pv = [&](auto kt_) __attribute__((always_inline)) {    // P, O^T += V^T P^T over 32 keys
                static_for<0, 2>([&](auto jb_) __attribute__((always_inline)) {
                    constexpr int kt = decltype(kt_)::value, jb = decltype(jb_)::value;
                    half8 pf;
                    static_for<0, 8>([&](auto u_) __attribute__((always_inline)) {
                        constexpr int u = decltype(u_)::value;
                        constexpr int r = 8 * jb + u, bi = kt * 16 + r;
                        const float vis = (float)visp[bi >> 1][bi & 1];
                        pf[u] = (_Float16)(e[kt][r] * (vis * cm + cu));
                    });
                    oacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[kt][jb], pf, oacc, 0, 0, 0);
                });
            };
            pv(I0{});
            sync();
            if (g < 9) bias_request(g + 1);                             // chunk 2
            pv(I1{});
            {   // O^T: lane = query, regs 0..7 = head dims (r&3) + 8*(r>>2) + 4*half -> 16 contiguous bytes after one
                // exchange; into the O buffer: row = token, 16-byte chunk 4 g + 2 head-in-group + half (swizzled as the rows are)
                union { half2v h2[2]; uint32_t u[2]; } lo4, hi4, rcv;
                lo4.h2[0] = half2v{(_Float16)oacc[0], (_Float16)oacc[1]}; lo4.h2[1] = half2v{(_Float16)oacc[2], (_Float16)oacc[3]};
                hi4.h2[0] = half2v{(_Float16)oacc[4], (_Float16)oacc[5]}; hi4.h2[1] = half2v{(_Float16)oacc[6], (_Float16)oacc[7]};
                rcv.u[0] = __shfl_xor(half ? lo4.u[0] : hi4.u[0], 32);
                rcv.u[1] = __shfl_xor(half ? lo4.u[1] : hi4.u[1], 32);
                typedef uint32_t uint4v __attribute__((ext_vector_type(4)));
                uint4v ov;
                if (half == 0) ov = uint4v{lo4.u[0], lo4.u[1], rcv.u[0], rcv.u[1]};
                else ov = uint4v{rcv.u[0], rcv.u[1], hi4.u[0], hi4.u[1]};
                const int c = 4 * g + 2 * ahl + half;
                const uint32_t oaddr = obase + (uint32_t)(((c & ~7) | ((c ^ osw) & 7)) * 16);
                // (inline asm: before an ordinary LDS store hipcc may wait for memory operations it does not need)
                asm volatile("ds_write_b128 %0, %1" :: "v"(oaddr), "v"(ov) : "memory");
            }
        };

        bias_request(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // prologue barrier: the trunk rows are in LDS (the G-waves
        A4_BARRIER();                                                   // turn them into B fragments)
        asm volatile("" ::: "memory");
        sync(); sync(); sync();                                         // intervals 0-2: the G-waves compute qkv(0)
        A4_ST(0, 1);
#pragma unroll 1
        for (int g = 0; g < 10; ++g) {                                  // period g = intervals 3 g + 3 ...: attention of group g
            attend(g);                                                  // (group 9: beside the first proj pieces)
            if (g == 4) A4_ST(0, 2);
            if (g == 5) A4_ST(0, 3);
            if (g == 8) A4_ST(0, 4);
        }
        A4_ST(0, 5);
        // ---- proj, channels 160-319 of this wave's 16 tokens: intervals 33-36 carry one piece of these waves (k-steps 0-3) as
        // their second piece, intervals 37-39 two (k-steps 4-9)
        {
            uint32_t sb = sync();                                       // interval 33
            // the residual of these accumulators (channels 160-319 of the wave's tokens) is requested now and added at the end
            half4v xr[10];
            {
                const char* xp = xg + (size_t)token * 640 + (160 + 4 * lq) * 2;
                static_for<0, 10>([&](auto j_) __attribute__((always_inline)) { constexpr int j = decltype(j_)::value; a4_load8<32 * j>(xr[j], xp); });
            }
            static_for<0, 10>([&](auto j_) __attribute__((always_inline)) { oc[decltype(j_)::value] = float4v{0.f, 0.f, 0.f, 0.f}; });
            proj_piece(sb + A4_PIECE, 0);
#pragma unroll 1
            for (int g = 1; g < 4; ++g) { sb = sync(); proj_piece(sb + A4_PIECE, g); }
#pragma unroll 1
            for (int g = 4; g < 10; g += 2) { sb = sync(); proj_run(std::integral_constant<int, 2>{}, sb, sb + A4_PIECE, g, g + 1); }
            static_for<0, 10>([&](auto j_) __attribute__((always_inline)) {     // (every sync since the request waited vmcnt(0))
                constexpr int j = decltype(j_)::value;
                asm volatile("" : "+v"(xr[j]));
                static_for<0, 4>([&](auto r_) __attribute__((always_inline)) { constexpr int r = decltype(r_)::value; oc[j][r] += (float)xr[j][r]; });
            });
        }
        A4_ST(0, 6);
    } else {
        // =====================================================================================================================
        // GEMM waves
        // =====================================================================================================================
        const int gw = w - 8;
        // two pieces = 20 x 1 KB, contiguous in the packed stream and in the ring (slots 0,1 / 2,3): wave gw moves the 1 KB units
        // gw, gw + 8 and (gw < 4) gw + 16
        const char* wsrc = reinterpret_cast<const char*>(a.wpack) + gw * 1024 + lane * 16;
        char* const ring_w = smem + A4_RING + gw * 1024;
        auto issue_interval = [&](int k) __attribute__((always_inline)) {
            const char* s = wsrc + (size_t)k * (2 * A4_PIECE);
            char* d = ring_w + (k & 1) * (2 * A4_PIECE);
#if !defined(A4_G_NODMA) && !defined(A4_DMA_A)
            a4_dma16(s, d);
            a4_dma16(s + 8192, d + 8192);
            if (gw < 4) a4_dma16(s + 16384, d + 16384);
#else
            (void)s; (void)d;
#endif
        };
        issue_interval(0);
        int iv = 0;
        // interval boundary: this wave's LDS reads / writes are done and its parts of the interval's two pieces have landed ->
        // barrier -> request the next interval's pieces into the slots the previous interval used
        auto sync = [&]() __attribute__((always_inline)) {
            __builtin_amdgcn_sched_barrier(0);
            A4_T2(0);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            A4_BARRIER();
            asm volatile("" ::: "memory");
            A4_T2(1);
            __builtin_amdgcn_sched_barrier(0);
            if (iv + 1 < A4_NIV) issue_interval(iv + 1);
            const uint32_t sb = ring_a + (uint32_t)((iv & 1) * 2 * A4_PIECE);
            ++iv;
            return sb;
        };
        // qkv piece [16 channels][320 k]: 640-byte rows, 16-byte chunk ^ (row>>1)&7 within 128 B; k-step s = chunks 4 s .. 4 s + 3
        const int xs = (l15 >> 1) & 7;
        const uint32_t fo0 = (uint32_t)(l15 * 640 + ((lq ^ xs) & 7) * 16);            // even k-steps (+ 128 (s >> 1))
        const uint32_t fo1 = (uint32_t)(l15 * 640 + (((4 + lq) ^ xs) & 7) * 16);      // odd k-steps
        half8 Xf[10];
        half4v held[2];                                                               // a finished tile pair waiting for its barrier
        const float4v zero4 = {0.f, 0.f, 0.f, 0.f};
        const int stq = token * 32 + (((lq >> 1) ^ (l15 >> 3)) & 1) * 16 + (lq & 1) * 8;      // Q / K staging offset of this lane
        auto write_qk = [&](char* base) __attribute__((always_inline)) {
            *reinterpret_cast<half4v*>(base + stq) = held[0];
            *reinterpret_cast<half4v*>(base + 4096 + stq) = held[1];
        };
        // V transposed: [unit][dim][token].  Neighbouring lanes = neighbouring tokens exchange half of their values (DPP), so that the
        // even lane holds dims 4 lq, 4 lq + 1 and the odd lane dims 4 lq + 2, 4 lq + 3 of BOTH tokens: two 4-byte stores per lane
        // instead of four 2-byte ones (sub-dword LDS stores cost ~60 cycles each with all waves at them)
        auto write_v = [&](auto hl_, const half4v h) __attribute__((always_inline)) {
            constexpr int hl = decltype(hl_)::value;
            const int unit = (token >> 6) * 2 + hl, sq = token & 63;
            union { half2v h2; uint32_t u; } p01, p23;
            p01.h2 = half2v{h[0], h[1]}; p23.h2 = half2v{h[2], h[3]};
            const bool odd = (lane & 1) != 0;
            const uint32_t own = odd ? p23.u : p01.u;
            const uint32_t recv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(odd ? p01.u : p23.u), 0xB1, 0xF, 0xF, true);   // lane ^ 1
            const uint32_t ta = odd ? recv : own, tb = odd ? own : recv;          // first / second token of the pair
            const uint32_t w0 = __builtin_amdgcn_perm(tb, ta, 0x05040100u);       // (dim r, token), (dim r, token + 1)
            const uint32_t w1 = __builtin_amdgcn_perm(tb, ta, 0x07060302u);       // dim r + 1
            uint32_t* vt = reinterpret_cast<uint32_t*>(reinterpret_cast<_Float16*>(smem + A4_VT) + (unit * 16 + 4 * lq + (odd ? 2 : 0)) * A4_VROW + (sq & ~1));
            vt[0] = w0; vt[A4_VROW / 2] = w1;
        };
        // one interval of the qkv GEMM of a head group: channel tiles 2 I and 2 I + 1 (I = 0: q of the two heads, 1: k, 2: v), each
        // ten MFMAs over the whole K into one accumulator; the fragments of a half piece (5 k-steps) are read while the MFMAs of the
        // half before are issued
        auto qkv_interval = [&](auto I_) __attribute__((always_inline)) {
            constexpr int I = decltype(I_)::value;
            const uint32_t sb = sync();
#ifndef A4_G_NOSTAGE
            if constexpr (I == 1) write_qk(smem + A4_Q);                // the tiles of the interval before: their readers are done
            if constexpr (I == 2) write_qk(smem + A4_K);
#else
            asm volatile("" :: "v"(held[0]), "v"(held[1]));
#endif
#ifdef A4_NO_GEMM        // timing experiment: the GEMM waves only keep the cadence
            return;
#endif
            half8 wf[2][5];
            auto load = [&](auto h_) __attribute__((always_inline)) {
                constexpr int h = decltype(h_)::value, S = h & 1;
                const uint32_t pb = sb + (uint32_t)((h >> 1) * A4_PIECE);
#ifndef A4_G_NOLOAD
                static_for<0, 5>([&](auto q_) __attribute__((always_inline)) {
                    constexpr int q = decltype(q_)::value, s = 5 * (h & 1) + q;
                    a4_lds16<128 * (s >> 1)>(wf[S][q], pb + ((s & 1) ? fo1 : fo0));
                });
#else
                static_for<0, 5>([&](auto q_) __attribute__((always_inline)) { asm volatile("" : "+v"(wf[S][decltype(q_)::value])); });
                (void)pb;
#endif
            };
            float4v qa = zero4;
            auto mma = [&](auto h_) __attribute__((always_inline)) {
                constexpr int h = decltype(h_)::value, S = h & 1;
#ifndef A4_G_NOMMA
                static_for<0, 5>([&](auto q_) __attribute__((always_inline)) {
                    constexpr int q = decltype(q_)::value, s = 5 * (h & 1) + q;
                    qa = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[S][q], Xf[s], qa, 0, 0, 0);
                });
#else
                static_for<0, 5>([&](auto q_) __attribute__((always_inline)) { asm volatile("" : "+v"(qa) : "v"(wf[S][decltype(q_)::value])); });
#endif
                __builtin_amdgcn_sched_barrier(0);
            };
            auto arrived5 = [&](auto S_, auto n_) __attribute__((always_inline)) {
                constexpr int S = decltype(S_)::value;
                a4_arrived5<decltype(n_)::value>(wf[S][0], wf[S][1], wf[S][2], wf[S][3], wf[S][4]);
            };
            using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
            using I3 = std::integral_constant<int, 3>; using I5 = std::integral_constant<int, 5>;
            load(I0{}); load(I1{});
            arrived5(I0{}, I5{}); mma(I0{});
            load(I2{});
            arrived5(I1{}, I5{}); mma(I1{});
            const half4v t0 = {(_Float16)qa[0], (_Float16)qa[1], (_Float16)qa[2], (_Float16)qa[3]};
            qa = zero4;
            load(I3{});
            arrived5(I0{}, I5{}); mma(I2{});
            arrived5(I1{}, I0{}); mma(I3{});
            const half4v t1 = {(_Float16)qa[0], (_Float16)qa[1], (_Float16)qa[2], (_Float16)qa[3]};
            if constexpr (I < 2) { held[0] = t0; held[1] = t1; }
#ifndef A4_G_NOSTAGE
            else { write_v(I0{}, t0); write_v(I1{}, t1); }              // V^T: its readers finished two intervals ago
#else
            else { held[0] = t0; held[1] = t1; }
#endif
        };
        auto qkv_group = [&]() __attribute__((always_inline)) {
            qkv_interval(std::integral_constant<int, 0>{});
            qkv_interval(std::integral_constant<int, 1>{});
            qkv_interval(std::integral_constant<int, 2>{});
        };

        // after the prologue barrier the trunk rows are in LDS -> B fragments (k-step s, element e < 4: channel 32 s + 4 lq + e,
        // e >= 4: 32 s + 16 + 4 lq + (e - 4); the weights are packed in the same k order)
        auto load_trunk = [&]() __attribute__((always_inline)) {
            const char* xrow = smem + A4_O + token * 640 + (lq & 1) * 8;
            static_for<0, 10>([&](auto s_) __attribute__((always_inline)) {
                constexpr int s = decltype(s_)::value;
                const int c0 = 4 * s + (lq >> 1), c1 = c0 + 2;
                const half4v lo = *reinterpret_cast<const half4v*>(xrow + ((c0 & ~7) | ((c0 ^ tsw) & 7)) * 16);
                const half4v hi = *reinterpret_cast<const half4v*>(xrow + ((c1 & ~7) | ((c1 ^ tsw) & 7)) * 16);
                Xf[s] = half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            });
        };
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        A4_BARRIER();
        asm volatile("" ::: "memory");
        load_trunk();

        qkv_group();                                               // intervals 0-2: qkv(0)
        A4_ST(1, 1);
#pragma unroll 1
        for (int g = 0; g < 9; ++g) {                              // period g: qkv(g + 1) while the A-waves attend to group g
            qkv_group();
            if (g == 4) A4_ST(1, 2);
            if (g == 5) A4_ST(1, 3);
        }
        A4_ST(1, 4);
        // ---- proj, channels 0-159 of this wave's 16 tokens, accumulators from the residual in registers: intervals 30-32 two pieces
        // (k-steps 0-5), intervals 33-36 one (k-steps 6-9, the first piece), intervals 37-39 none
        static_for<0, 10>([&](auto j_) __attribute__((always_inline)) {
            constexpr int j = decltype(j_)::value;
            static_for<0, 4>([&](auto r_) __attribute__((always_inline)) {
                constexpr int r = decltype(r_)::value;
                oc[j][r] = (float)Xf[j >> 1][(j & 1) * 4 + r];
            });
        });
        {
            uint32_t sb = sync();                                  // interval 30 (its K tiles: none held -- the last group is done)
#pragma unroll 1
            for (int g = 0; g < 6; g += 2) {
                if (g) sb = sync();
                proj_run(std::integral_constant<int, 2>{}, sb, sb + A4_PIECE, g, g + 1);
            }
#pragma unroll 1
            for (int g = 6; g < 10; ++g) { sb = sync(); proj_piece(sb, g); }
            sync(); sync(); sync();                                // intervals 37-39: the A-waves' last pieces
        }
        A4_ST(1, 5);
    }
    // every wave has left the ring and the O buffer
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    A4_BARRIER();
    A4_ST(1, 6); A4_ST(0, 7);
#ifdef A4_STAMP2
    if (lane == 0 && (w & 3) == 0)
        for (int k = 0; k < 12; ++k) g_a4_stamp[((size_t)blockIdx.x * 4 + (w >> 2)) * 16 + k] = ts2[k];
#endif
#ifdef A4_DRY
    if (lane == 0) reinterpret_cast<int*>(a.y)[blockIdx.x * 16 + w] = nbar;
    return;
#endif

    // ---- epilogue, all 16 waves: a wave holds 16 tokens x 160 channels (the residual is already in)
    const int chalf = w < 8 ? 1 : 0;
    const int ch0 = chalf * 160 + 4 * lq;                          // channel of oc[j][r]: ch0 + 16 j + r
    float s1 = 0.f, s2 = 0.f;
    static_for<0, 10>([&](auto j_) __attribute__((always_inline)) {
        constexpr int j = decltype(j_)::value;
        static_for<0, 4>([&](auto r_) __attribute__((always_inline)) {
            constexpr int r = decltype(r_)::value;
            const float v = oc[j][r];
            s1 += v; s2 += v * v;
        });
    });
    s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
    float2* xch = reinterpret_cast<float2*>(smem + A4_XCH);
    if (lq == 0) xch[chalf * 128 + token] = make_float2(s1, s2);
    __syncthreads();
    {
        const float2 p0 = xch[token], p1 = xch[128 + token];       // channels 0-159, 160-319: the same sum order in both waves
        s1 = p0.x + p1.x; s2 = p0.y + p1.y;
    }
    const float cnt = (float)a.ln_count;
    const float mean = s1 / cnt;
    float var = s2 / cnt - mean * mean;
    var = var > 0.f ? var : 0.f;
    const float rstd = rsqrtf(var + 1e-5f);
    const float* par = reinterpret_cast<const float*>(smem + A4_PAR);
    float2* scr = reinterpret_cast<float2*>(smem + A4_RING);            // [16 waves][10][4] GroupNorm partials
    float2* tot = scr + 16 * 10 * 4;                                    // [2 boards][20] (mean, rstd)
    const float nmr = -mean * rstd;
    char* xrow = smem + A4_O + token * 640 + (lq & 1) * 8;
    static_for<0, 10>([&](auto j_) __attribute__((always_inline)) {
        constexpr int j = decltype(j_)::value;
        const float4 gm = *reinterpret_cast<const float4*>(par + ch0 + 16 * j);
        const float4 bt = *reinterpret_cast<const float4*>(par + 320 + ch0 + 16 * j);
        const float gmv[4] = {gm.x, gm.y, gm.z, gm.w}, btv[4] = {bt.x, bt.y, bt.z, bt.w};
        float p1 = 0.f, p2 = 0.f;
        half4v h;
        static_for<0, 4>([&](auto r_) __attribute__((always_inline)) {
            constexpr int r = decltype(r_)::value;
            const float y = fmaf(fmaf(oc[j][r], rstd, nmr), gmv[r], btv[r]);      // (v - mean) rstd gamma + beta, two FMAs
            p1 += y; p2 += y * y;
            h[r] = (_Float16)y;
        });
        const int chunk = chalf * 20 + 2 * j + (lq >> 1);
        const int pos = (chunk & ~7) | ((chunk ^ tsw) & 7);
        *reinterpret_cast<half4v*>(xrow + pos * 16) = h;                // y image, the layout of the rows
        p1 = a4_row_sum(p1); p2 = a4_row_sum(p2);
        if (l15 == 0) scr[(w * 10 + j) * 4 + lq] = make_float2(p1, p2);
    });
    __syncthreads();
    // the image leaves in 16-byte stores, the swizzle undone on the way: 128 rows x 40 chunks over 1024 threads
    auto flush = [&](_Float16* outp) __attribute__((always_inline)) {
        char* og = reinterpret_cast<char*>(outp) + b0 * 64 * 640;
#pragma unroll
        for (int n = 0; n < 5; ++n) {
            const int q = n * 1024 + tid;
            const int row = q / 40, pos = q - row * 40;
            const int src = (pos & ~7) | ((pos ^ (row >> 1)) & 7);
            const uint4 v = *reinterpret_cast<const uint4*>(smem + A4_O + row * 640 + pos * 16);
            *reinterpret_cast<uint4*>(og + row * 640 + src * 16) = v;
        }
    };
    flush(a.y);
    A4_ST(1, 7);
    if (a.y2 == nullptr) return;
    // ---- second output: act(GroupNorm16(y)) for the next residual block (statistics per board and 16-channel group)
    if (tid < 40) {
        const int bd = tid / 20, J = tid - bd * 20;
        const int wbase = (J >= 10 ? 0 : 8) + 4 * bd, jj = J >= 10 ? J - 10 : J;
        float s = 0.f, ss = 0.f;
        for (int ww = 0; ww < 4; ++ww)
            for (int q = 0; q < 4; ++q) { const float2 v = scr[((wbase + ww) * 10 + jj) * 4 + q]; s += v.x; ss += v.y; }
        const float mu = s * (1.f / 1024.f);
        float vr = ss * (1.f / 1024.f) - mu * mu;
        vr = vr > 0.f ? vr : 0.f;
        tot[tid] = make_float2(mu, rsqrtf(vr + 1e-5f));
    }
    __syncthreads();            // (also: every thread's flush reads of the y image are done before it is overwritten below)
    // per (board, channel) scale and shift over the gamma / beta slots (the second GroupNorm's parameters are dead after this)
    {
        float* parw = reinterpret_cast<float*>(smem + A4_PAR);
        float scv[2] = {0.f, 0.f}, shv[2] = {0.f, 0.f};
        if (tid < 320) {
            const float g2 = parw[640 + tid], b2 = parw[960 + tid];
#pragma unroll
            for (int bd = 0; bd < 2; ++bd) {
                const float2 mr = tot[bd * 20 + (tid >> 4)];
                scv[bd] = g2 * mr.y; shv[bd] = b2 - mr.x * scv[bd];
            }
        }
        __syncthreads();
        if (tid < 320) { parw[tid] = scv[0]; parw[320 + tid] = shv[0]; parw[640 + tid] = scv[1]; parw[960 + tid] = shv[1]; }
        __syncthreads();
    }
    static_for<0, 10>([&](auto j_) __attribute__((always_inline)) {
        constexpr int j = decltype(j_)::value;
        const float4 gm = *reinterpret_cast<const float4*>(par + (tw >> 2) * 640 + ch0 + 16 * j);
        const float4 bt = *reinterpret_cast<const float4*>(par + (tw >> 2) * 640 + 320 + ch0 + 16 * j);
        const float gmv[4] = {gm.x, gm.y, gm.z, gm.w}, btv[4] = {bt.x, bt.y, bt.z, bt.w};
        const int chunk = chalf * 20 + 2 * j + (lq >> 1);
        const int pos = (chunk & ~7) | ((chunk ^ tsw) & 7);
        half4v h = *reinterpret_cast<const half4v*>(xrow + pos * 16);
        static_for<0, 4>([&](auto r_) __attribute__((always_inline)) {
            constexpr int r = decltype(r_)::value;
            h[r] = (_Float16)act_fast<ACT>((float)h[r] * gmv[r] + btv[r]);
        });
        *reinterpret_cast<half4v*>(xrow + pos * 16) = h;
    });
    __syncthreads();
    flush(a.y2);
    A4_ST(1, 8);
}

hipError_t launch_attn_block4(const AttnBlockArgs& a, hipStream_t st) {
    if (a.B <= 0 || a.B % 2 != 0 || a.ln_count <= 0 || a.ln_count > 320) return hipErrorInvalidValue;
    if (a.y2 != nullptr && a.act != ACT_SILU && a.act != ACT_RELU) return hipErrorInvalidValue;
    static DeviceOnce once;
    hipError_t e = once.run([] {
        hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_block4_kernel<ACT_SILU>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, A4_LDS);
        if (r != hipSuccess) return r;
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_block4_kernel<ACT_RELU>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, A4_LDS);
    });
    if (e != hipSuccess) return e;
    const dim3 grid((unsigned)(a.B / 2));
    if (a.act == ACT_RELU) hipLaunchKernelGGL(attn_block4_kernel<ACT_RELU>, grid, dim3(A4_THREADS), A4_LDS, st, a);
    else hipLaunchKernelGGL(attn_block4_kernel<ACT_SILU>, grid, dim3(A4_THREADS), A4_LDS, st, a);
    return hipGetLastError();
}

// 80 pieces of 10 KB in consumption order
size_t attn_block4_pack_bytes() { return (size_t)A4_NPIECES * A4_PIECE; }
// stream position of a piece: qkv tile j (0-5: q0 q1 k0 k1 v0 v1) of head group g, or proj k-step g, channel half hh
int attn_block4_qkv_pos(int g, int j) { return 6 * g + j; }
int attn_block4_proj_pos(int g, int hh) {
    if (hh == 0) return g < 6 ? 60 + g : 66 + 2 * (g - 6);
    return g < 4 ? 67 + 2 * g : 74 + (g - 4);
}
// channel that sits at k-slot kl (0..31) of a qkv k-step: slot (lq = kl >> 3, e = kl & 7) holds channel 4 lq + e (e < 4) or
// 16 + 4 lq + (e - 4) of the k-step's 32, matching the register-resident trunk fragments
int attn_block4_qkv_kperm(int kl) { const int lq = kl >> 3, e = kl & 7; return e < 4 ? 4 * lq + e : 16 + 4 * lq + (e - 4); }
