// attn_block_kernel: one whole ChessAttention block of the tower (resnet.py:133-181: qkv 1x1 -> per-head
// scores / softmax / PV -> proj 1x1 -> residual add -> LayerNorm) plus the pre-activation GroupNorm of the residual
// block that follows, in ONE kernel for the 320-channel trunk.
//
// Why: as four kernels (qkv GEMM, attn_core, proj GEMM, ew_board) the block moved 2.5 GB through HBM per launch at
// 4096 boards -- the [B][64][960] qkv tensor alone is written and read back (1 GB) -- and took 856 us against 310 us of
// traffic; here only the trunk is read (168 MB) and the outputs are written.
//
// Workgroup = 2 boards = 128 token rows, 8 waves.  The boards' trunk rows X [128][320] stay in LDS for the whole
// kernel (A operand of every qkv GEMM and the residual at the end).  The 20 heads are processed in 10 groups of 2:
//   1. qkv GEMM of the group  [128 x 320] x [320 x 96]   (96 = q,k,v of 2 heads; 5 weight pieces of K = 64)
//      -> Q, K as [head][token][16] and V transposed [unit][16][token] in LDS (fp16, as the split path's qkv tensor);
//   2. attention of the 4 (board, head) units, one per wave pair (a wave owns 32 queries): S^T = K Q^T and
//      O^T = V^T P^T on MFMA 32x32x16 exactly as attn_core_kernel; the relative-position bias of the wave's
//      (head, query half) arrives in registers from a table pre-arranged in accumulator order; O overwrites Q;
//   3. proj GEMM accumulate  out[128 x 320] += O[128 x 32] x Wproj[32 x 320]  (2 weight pieces) into 80 accumulator
//      registers per lane that live across all groups (a wave owns 16 tokens x all 320 channels, so LayerNorm needs no
//      cross-wave reduction and GroupNorm group j is accumulator tile j).
// All GEMM tiles are MFMA 16x16x32 in the "swapped" orientation (A = weights, B = activations): the accumulator then
// holds 4 consecutive channels of one token per lane = one 8-byte LDS write.
// The weights of the whole block are one stream of 70 pieces of 12 KB (host-packed in LDS image order, net.hip)
// through a 4-slot ring filled by global_load_lds (two DMA instructions per wave and piece, so every wave's vmcnt
// bookkeeping is identical); one barrier per piece.  The proj pieces of a group are consumed together with the qkv pieces
// of the next one as one software-pipelined sequence: the fragments of the next half-piece are read from LDS (untracked
// inline-asm reads, counted lgkmcnt waits) while the MFMAs of the current one issue.
#include "kernel_common.h"
#include "conv_epilogue.h"

typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));

namespace {
constexpr int AB_PIECE = 12288;
constexpr int AB_PIECES_PER_GROUP = 7;
constexpr int AB_GROUPS = 10;
constexpr int AB_X = 0;                                   // [128][640 B], 16-byte chunk ^ (row>>1)&7 within 128 B
constexpr int AB_QK = 81920;                              // Q [2][128][16] | K [2][128][16]   (O overlays Q); a token's two 16-byte
                                                          // halves are stored at half ^ (token >> 3 & 1): conflict-free ds_read_b128 of
                                                          // 32 consecutive tokens, the staging's 8-byte writes 2-way instead of 4-way
constexpr int AB_VT = AB_QK + 16384;                      // [4 units][16][68]
constexpr int AB_VROW = 68;
constexpr int AB_RING = AB_VT + 4 * 16 * AB_VROW * 2;     // 107008
constexpr int AB_PAR = AB_RING + 4 * AB_PIECE;            // 156160: LayerNorm gamma, beta, next GroupNorm gamma, beta [4][320] f32
constexpr int AB_LDS = AB_PAR + 4 * 320 * 4;              // 161280
}

__device__ __forceinline__ void ab_dma16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// 64 bytes per lane from global memory that the compiler does not track (the caller waits: vmcnt)
__device__ __forceinline__ void ab_load64(half8& b0, half8& b1, half8& b2, half8& b3, const half8* p) {
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\t"
                 "global_load_dwordx4 %2, %4, off offset:32\n\tglobal_load_dwordx4 %3, %4, off offset:48"
                 : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3) : "v"(p) : "memory");
}
// one 16-byte LDS read the compiler does not track (the caller waits: lgkmcnt)
template <int OFF>
__device__ __forceinline__ void ab_lds16(half8& d, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
// ... and the wait: at most N younger LDS reads outstanding; the operands tie their first use to this point
template <int N>
__device__ __forceinline__ void ab_lds_arrived(half8& f0, half8& f1, half8& f2, half8& f3, half8& f4, half8& f5) {
    static_assert(N == 0 || N == 5, "");
    if constexpr (N == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5) :: "memory");
    else asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5) :: "memory");
}
template <int CTRL>
__device__ __forceinline__ float ab_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over the 16 lanes of a DPP row (every lane gets the total; fixed order)
__device__ __forceinline__ float ab_row_sum(float v) {
    v += ab_dpp<0xB1>(v);      // quad_perm [1,0,3,2]
    v += ab_dpp<0x4E>(v);      // quad_perm [2,3,0,1]
    v += ab_dpp<0x141>(v);     // row_half_mirror
    v += ab_dpp<0x140>(v);     // row_mirror
    return v;
}

#ifndef AB_DBG
#define AB_DBG 0
#endif
#define AB_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#if defined(AB_STAMP) || defined(AB_STAMP2) || defined(AB_STAMP3)
__device__ unsigned long long* g_ab_stamp;        // [blocks][16] s_memtime stamps of wave 0 (tools/ubench/attn_block_bench.hip)
#ifdef AB_STAMP
#define AB_ST(k) do { if (tid == 0) g_ab_stamp[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define AB_ST(k) do {} while (0)
#endif
#else
#define AB_ST(k) do {} while (0)
#endif
#define AB_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#ifdef AB_STAMP3      // ten stamps inside ONE piece (piece 3 of group 5's sequence), same mechanism as AB_STAMP2
#define AB_T3(k) do { if (g == 5) ts[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define AB_T3(k) do {} while (0)
#endif
#ifdef AB_STAMP2      // stamps kept in scalar registers and stored at the end of the kernel (waves 0 and 4): no store, no wait at the stamp
#define AB_TS(k) do { if (g == 5) ts[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define AB_TS(k) do {} while (0)
#endif

template <int ACT>
__global__ __launch_bounds__(512) void attn_block_kernel(AttnBlockArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4, r31 = lane & 31, half = lane >> 5;
    const int wm = w >> 1, wn = w & 1;
    const size_t b0 = (size_t)blockIdx.x * 2;
    const char* xg = reinterpret_cast<const char*>(a.x) + b0 * 64 * 640;

    AB_ST(0);
    // ---- prologue: the two boards' rows and the first three weight pieces
#pragma unroll
    for (int n = 0; n < 10; ++n) {
        const int idx = w * 10 + n;
        const int q = idx * 64 + lane;
        const int row = q / 40, pos = q - row * 40;
        const int src = (pos & ~7) | ((pos ^ (row >> 1)) & 7);
        ab_dma16(xg + row * 640 + src * 16, smem + AB_X + idx * 1024);
    }
    // a piece is 12 x 1 KB: every wave issues one full 16-byte DMA and one with its upper 32 lanes masked off (1.5 KB per
    // wave), so the count of outstanding vector-memory operations is the same in all 8 waves.  (global_load_lds_dwordx3
    // would give 16 x 768 B, but on gfx950 it places lane i's 12 bytes at base + 16 i.  Twelve full instructions -- waves 0-3
    // two, waves 4-7 one, with per-wave wait counts -- measured the same or slower.)
    const char* wsrc = reinterpret_cast<const char*>(a.wpack) + w * 1536 + lane * 16;
    char* const ring_w = smem + AB_RING + w * 1536;
    auto issue = [&](int t) __attribute__((always_inline)) {
        const char* s = wsrc + (size_t)t * AB_PIECE;
        char* d = ring_w + (t & 3) * AB_PIECE;
        ab_dma16(s, d);
        if (lane < 32) ab_dma16(s + 1024, d + 1024);
    };
    // AB_WAIT(4) = all but this wave's two youngest pieces have landed.  Only LDS-DMA operations may be outstanding at a
    // counted wait: loads into registers and loads into LDS do not retire in one order (measured: a later DMA retired
    // before an earlier register load and vmcnt(1) let a wave read its bias registers early), so a count over both kinds
    // proves nothing about either.  The bias loads below are therefore issued after a sequence's last counted wait and
    // waited for with vmcnt(0).
    issue(0); issue(1); issue(2);
    if (tid < 320) {
        float* par = reinterpret_cast<float*>(smem + AB_PAR);
        par[tid] = a.ln_g[tid]; par[320 + tid] = a.ln_b[tid];
        par[640 + tid] = a.y2 ? a.gn2_gamma[tid] : 0.f; par[960 + tid] = a.y2 ? a.gn2_beta[tid] : 0.f;
    }

#if AB_DBG == 3
    AB_WAIT(0);
    __syncthreads();
    if (blockIdx.x == 0)
        for (int i = tid; i < 3 * AB_PIECE / 4; i += 512)
            reinterpret_cast<uint32_t*>(a.y)[i] = reinterpret_cast<const uint32_t*>(smem + AB_RING)[i];
    return;
#endif
    // ---- attention role of this wave: unit = (board, head-in-group), query half
    const int au = w >> 1, aboard = au >> 1, ahl = au & 1, aqt = w & 1;
    const int aq = aqt * 32 + r31;
    // visibility of key (kt, r) from query aq as a multiplicand, accumulator order: key = kt*32 + 8(r>>2) + 4 half + (r&3)
    // (fp16 pairs: 16 registers; the products below take them as the fp16 operand of a mixed-precision FMA)
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

    // ---- per-lane LDS offsets
    const int xrow0 = 32 * wm + l15;
    const int xsw = (xrow0 >> 1) & 7;
    const int xe0 = ((lq ^ xsw) & 7) * 16, xe1 = (((4 + lq) ^ xsw) & 7) * 16;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const uint32_t xa[2] = {lds0 + AB_X + xrow0 * 640 + xe0, lds0 + AB_X + xrow0 * 640 + xe1};      // rows xrow0 and (+16 * 640) xrow0 + 16
    const int wsw = (l15 >> 1) & 7;
    const int wq0 = (3 * wn * 16 + l15) * 128 + ((lq ^ wsw) & 7) * 16;            // qkv piece, kk = 0
    const int wq1 = (3 * wn * 16 + l15) * 128 + (((4 + lq) ^ wsw) & 7) * 16;      // kk = 1
    // proj piece: 64-byte rows; a ds_read_b128 is served in lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (one row
    // quad of lq = 0 / 2 next to two of lq = 1 / 3): chunk ^ (4 - quad) & 3 gives the 16 lanes of a group 16 different bank quads
    const int wpo = l15 * 64 + ((lq ^ (4 - (l15 >> 2))) & 3) * 16;
    const uint32_t ring_a = lds0 + AB_RING;
    const uint32_t of_a = lds0 + AB_QK + (lq >> 1) * 4096 + (16 * w + l15) * 32 + ((lq & 1) ^ (l15 >> 3)) * 16;   // this wave's O rows (proj operand)

    const float4v zero4 = {0.f, 0.f, 0.f, 0.f};
    const float16v zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float4v oc[20];
    static_for<0, 20>([&](auto j_) __attribute__((always_inline)) { oc[decltype(j_)::value] = zero4; });

    // The block's 70 weight pieces are consumed as one sequence per head group: [proj piece 0, 1 of the PREVIOUS group,] qkv
    // piece 0..4 of this group, each piece in two halves (one k-step of the qkv GEMM / 80 output channels of the proj).
    // The fragments of half u+1 (weights from the ring, trunk rows / O from LDS) are read into the second register set before
    // the MFMAs of half u are issued, so a half's LDS reads run under the matrix work of the half before it instead of in
    // front of their own (both waves of a SIMD sit at the same barrier: nothing else would overlap them).  Per piece
    // boundary: this wave's reads of piece i are complete (lgkmcnt) and its parts of piece i+1 have landed (vmcnt) ->
    // barrier -> read the first half of piece i+1 -> DMA piece i+4 into the slot of piece i -> MFMAs of the last half of i.
#if defined(AB_STAMP2) || defined(AB_STAMP3)
    unsigned long long ts[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    half8 fs[2][5];
    half8 of;
    float4v qa[2][3];
    half8 bias8[4];
    auto sequence = [&](auto hp_, auto hq_, const int T, const int g) __attribute__((always_inline)) {
        constexpr bool HP = decltype(hp_)::value, HQ = decltype(hq_)::value;
        constexpr int NP = (HP ? 2 : 0) + (HQ ? 5 : 0), NU = 2 * NP;
        constexpr int BU = NU - 3;                            // the half under which the bias is requested: after the last counted wait
        // (fragment reads are inline asm with counted lgkmcnt waits below: after any inline asm hipcc's own waits are
        // lgkmcnt(0), which would put every half's reads in front of the MFMAs of the half before it again)
        auto load = [&](auto u_) __attribute__((always_inline)) {
            constexpr int u = decltype(u_)::value, S = u & 1, i = u >> 1, kk = u & 1;
            const uint32_t slot = ring_a + (uint32_t)(((T + i) & 3) * AB_PIECE);
            if constexpr (HP && i < 2) {
                if constexpr (u == 0) ab_lds16<0>(of, of_a);
                const uint32_t pa = slot + wpo;
                static_for<0, 5>([&](auto jj_) __attribute__((always_inline)) {
                    constexpr int jj = decltype(jj_)::value;
                    ab_lds16<(5 * kk + jj) * 1024>(fs[S][jj], pa);
                });
            } else {
                constexpr int p = i - (HP ? 2 : 0);
                ab_lds16<128 * p>(fs[S][3], xa[kk]);
                ab_lds16<128 * p + 16 * 640>(fs[S][4], xa[kk]);
                const uint32_t wa = slot + (kk ? wq1 : wq0);
                static_for<0, 3>([&](auto j_) __attribute__((always_inline)) {
                    constexpr int j = decltype(j_)::value;
                    ab_lds16<j * 2048>(fs[S][j], wa);
                });
            }
        };
        // the set of half u is in registers once at most N younger LDS reads are outstanding
        auto arrived = [&](auto u_, auto n_) __attribute__((always_inline)) {
            constexpr int S = decltype(u_)::value & 1;
            ab_lds_arrived<decltype(n_)::value>(fs[S][0], fs[S][1], fs[S][2], fs[S][3], fs[S][4], of);
        };
        auto mma = [&](auto u_) __attribute__((always_inline)) {
            constexpr int u = decltype(u_)::value, S = u & 1, i = u >> 1, kk = u & 1;
            if constexpr (HP && i < 2) {
                static_for<0, 5>([&](auto jj_) __attribute__((always_inline)) {
                    constexpr int c = 10 * i + 5 * kk + decltype(jj_)::value;
                    oc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fs[S][decltype(jj_)::value], of, oc[c], 0, 0, 0);
                });
            } else {
                static_for<0, 3>([&](auto j_) __attribute__((always_inline)) {
                    constexpr int j = decltype(j_)::value;
                    qa[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fs[S][j], fs[S][3], qa[0][j], 0, 0, 0);
                    qa[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fs[S][j], fs[S][4], qa[1][j], 0, 0, 0);
                });
            }
        };
        if constexpr (HP && HQ) AB_TS(0);
        AB_LGKM0();                                           // this wave's O rows of the previous group are in LDS
        AB_WAIT(4);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (HP && HQ) AB_TS(1);
        load(std::integral_constant<int, 0>{});
        issue(T + 3);
        static_for<0, NU>([&](auto u_) __attribute__((always_inline)) {
            constexpr int u = decltype(u_)::value;
            if constexpr (u + 1 < NU) {
                if constexpr (u & 1) {
                    if constexpr (HP && HQ && u == 7) AB_T3(7);
                    arrived(u_, std::integral_constant<int, 0>{});
                    if constexpr (HP && HQ && u == 7) AB_T3(8);
                    AB_WAIT(4);
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    if constexpr (HP && HQ && u == 5) AB_T3(0);
                    if constexpr (HP && HQ && u == 7) AB_T3(9);
                    if constexpr (HP && HQ && (u == 1 || u == 3 || u == 7 || u == 11)) { if (g == 5) AB_ST(u == 1 ? 12 : u == 3 ? 13 : u == 7 ? 14 : 15); }
                    if constexpr (HP && HQ) AB_TS(2 + (u >> 1));
                }
                load(std::integral_constant<int, u + 1>{});
                if constexpr (HP && HQ && u == 5) AB_T3(1);
                if constexpr (HP && HQ && u == 6) AB_T3(4);
                if constexpr (u & 1) issue(T + (u >> 1) + 4);
                else arrived(u_, std::integral_constant<int, 5>{});
                if constexpr (HP && HQ && u == 5) AB_T3(2);
                if constexpr (HP && HQ && u == 6) AB_T3(5);
            } else {
                arrived(u_, std::integral_constant<int, 0>{});
            }
            if constexpr (HQ && u == BU) {
                // relative-position bias of (head, query half) in accumulator order: 64 B per lane, requested after the
                // sequence's last counted wait (see above) and waited for with vmcnt(0) before the scores
                // (inline asm: at the first use of an ordinary load's result hipcc waits vmcnt(0) wherever that use lands)
                const half8* bp = reinterpret_cast<const half8*>(a.bias) + ((size_t)((2 * g + ahl) * 2 + aqt) * 64 + lane) * 4;
                ab_load64(bias8[0], bias8[1], bias8[2], bias8[3], bp);
            }
            mma(u_);
            __builtin_amdgcn_sched_barrier(0);                // the next half's wait stays behind these MFMAs
            if constexpr (HP && HQ && u == 5) AB_T3(3);
            if constexpr (HP && HQ && u == 6) AB_T3(6);
            if constexpr (HP && HQ && u == NU - 1) AB_TS(8);
        });
    };

    auto zero_qa = [&]() __attribute__((always_inline)) {
        static_for<0, 2>([&](auto i_) __attribute__((always_inline)) {
            static_for<0, 3>([&](auto j_) __attribute__((always_inline)) { qa[decltype(i_)::value][decltype(j_)::value] = zero4; });
        });
    };
    auto stage_and_attend = [&](const int g) __attribute__((always_inline)) {
        if (g == 5) AB_ST(5);
        // ---- stage q, k (token-major) and v (transposed) as fp16
        static_for<0, 2>([&](auto i_) __attribute__((always_inline)) {
            static_for<0, 3>([&](auto j_) __attribute__((always_inline)) {
                constexpr int i = decltype(i_)::value, j = decltype(j_)::value;
                const int J = 3 * wn + j, type = J >> 1, hl = J & 1;          // wave-uniform
                const int token = 32 * wm + 16 * i + l15;
                const half4v h = {(_Float16)qa[i][j][0], (_Float16)qa[i][j][1], (_Float16)qa[i][j][2], (_Float16)qa[i][j][3]};
                if (type < 2) {
                    *reinterpret_cast<half4v*>(smem + AB_QK + type * 8192 + hl * 4096 + token * 32 + (((lq >> 1) ^ (l15 >> 3)) & 1) * 16 + (lq & 1) * 8) = h;
                } else {
                    // V transposed: [unit][dim][token].  Two-byte stores (one per dim and lane) cost ~60 cycles each with all
                    // waves at it (sub-dword LDS writes); instead neighbouring lanes = neighbouring tokens exchange half of their
                    // values (DPP), so that the even lane holds dims 4 lq, 4 lq + 1 and the odd lane dims 4 lq + 2, 4 lq + 3 of
                    // BOTH tokens: two 4-byte stores per lane, the same bytes in the same places
                    const int unit = (token >> 6) * 2 + hl, sq = token & 63;
                    union { half2v h2; uint32_t u; } p01, p23;
                    p01.h2 = half2v{h[0], h[1]}; p23.h2 = half2v{h[2], h[3]};
                    const bool odd = (lane & 1) != 0;
                    const uint32_t own = odd ? p23.u : p01.u;
                    const uint32_t recv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(odd ? p01.u : p23.u), 0xB1, 0xF, 0xF, true);   // lane ^ 1
                    const uint32_t ta = odd ? recv : own, tb = odd ? own : recv;          // first / second token of the pair
                    const uint32_t w0 = __builtin_amdgcn_perm(tb, ta, 0x05040100u);       // (dim r, token), (dim r, token + 1)
                    const uint32_t w1 = __builtin_amdgcn_perm(tb, ta, 0x07060302u);       // dim r + 1
                    uint32_t* vt = reinterpret_cast<uint32_t*>(reinterpret_cast<_Float16*>(smem + AB_VT) +
                                                               (unit * 16 + 4 * lq + (odd ? 2 : 0)) * AB_VROW + (sq & ~1));
                    vt[0] = w0; vt[AB_VROW / 2] = w1;
                }
            });
        });
        AB_LGKM0();                                           // raw barrier: __syncthreads() would drain the weight DMA
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (g == 5) AB_ST(6);
        AB_TS(9);
        // ---- 2. attention of this wave's (board, head, query half)
        if (AB_DBG != 2) {
            const char* Kb = smem + AB_QK + 8192 + ahl * 4096 + aboard * 64 * 32;
            const int hsw = 16 * (half ^ ((r31 >> 3) & 1));                                     // this lane's half of its token's row
            const half8 kf0 = *reinterpret_cast<const half8*>(Kb + r31 * 32 + hsw);
            const half8 kf1 = *reinterpret_cast<const half8*>(Kb + (32 + r31) * 32 + hsw);
            char* Qp = smem + AB_QK + ahl * 4096 + (aboard * 64 + aq) * 32 + hsw;            // also where O goes
            const half8 qfr = *reinterpret_cast<const half8*>(Qp);
            half8 vf[2][2];
            {
                const _Float16* vrow = reinterpret_cast<const _Float16*>(smem + AB_VT) + (au * 16 + l15) * AB_VROW;
                static_for<0, 2>([&](auto kt_) __attribute__((always_inline)) {
                    static_for<0, 2>([&](auto jb_) __attribute__((always_inline)) {
                        constexpr int kt = decltype(kt_)::value, jb = decltype(jb_)::value;
                        const half4v lo = *reinterpret_cast<const half4v*>(vrow + kt * 32 + 16 * jb + 4 * half);
                        const half4v hi = *reinterpret_cast<const half4v*>(vrow + kt * 32 + 16 * jb + 8 + 4 * half);
                        vf[kt][jb] = half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    });
                });
            }
            float16v st[2];
            st[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf0, qfr, zero16, 0, 0, 0);
            st[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf1, qfr, zero16, 0, 0, 0);
            // the bias and every piece issued before it (those landed long ago); the operands tie the registers to this point
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(bias8[0]), "+v"(bias8[1]), "+v"(bias8[2]), "+v"(bias8[3]) :: "memory");
            float e[2][16];
            float su = 0.f, sm = 0.f;
            static_for<0, 2>([&](auto kt_) __attribute__((always_inline)) {
                static_for<0, 16>([&](auto r_) __attribute__((always_inline)) {
                    constexpr int kt = decltype(kt_)::value, r = decltype(r_)::value;
                    constexpr int bi = kt * 16 + r;
                    float d = st[kt][r] * isd + (float)bias8[bi >> 3][bi & 7];
                    d = __builtin_amdgcn_fmed3f(d, -clampv, clampv);
                    const float eu = __builtin_amdgcn_exp2f(d);
                    e[kt][r] = eu;
                    su += eu;
                    sm += eu * (float)visp[bi >> 1][bi & 1];
                });
            });
            su += __shfl_xor(su, 32);
            sm += __shfl_xor(sm, 32);
            const float cu = wu_ / su, cm = wm_ / sm;
            float16v oacc = zero16;
            static_for<0, 2>([&](auto kt_) __attribute__((always_inline)) {
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
            });
            // O^T: lane = query, regs 0..7 = head dims (r&3) + 8*(r>>2) + 4*half -> 16 contiguous bytes after one exchange
            union { half2v h2[2]; uint32_t u[2]; } lo4, hi4, rcv;
            lo4.h2[0] = half2v{(_Float16)oacc[0], (_Float16)oacc[1]}; lo4.h2[1] = half2v{(_Float16)oacc[2], (_Float16)oacc[3]};
            hi4.h2[0] = half2v{(_Float16)oacc[4], (_Float16)oacc[5]}; hi4.h2[1] = half2v{(_Float16)oacc[6], (_Float16)oacc[7]};
            rcv.u[0] = __shfl_xor(half ? lo4.u[0] : hi4.u[0], 32);
            rcv.u[1] = __shfl_xor(half ? lo4.u[1] : hi4.u[1], 32);
            typedef uint32_t uint4v __attribute__((ext_vector_type(4)));
            uint4v ov;
            if (half == 0) ov = uint4v{lo4.u[0], lo4.u[1], rcv.u[0], rcv.u[1]};
            else ov = uint4v{rcv.u[0], rcv.u[1], hi4.u[0], hi4.u[1]};
            // (inline asm: before an ordinary LDS store hipcc waits for every LDS-DMA in flight, vmcnt(0))
            asm volatile("ds_write_b128 %0, %1" :: "v"((uint32_t)(uintptr_t)Qp), "v"(ov) : "memory");
        }
        if (g == 5) AB_ST(7);
    };

    // qkv of group 0 | 9 x (attention of group g, then proj of g and qkv of g + 1 as one sequence) | attention and proj of group 9
    if (AB_DBG != 1) {
        zero_qa();
        sequence(std::false_type{}, std::true_type{}, 0, 0);
#pragma unroll 1
        for (int g = 0; g < AB_GROUPS - 1; ++g) {
            stage_and_attend(g);
            if (g == 4) AB_ST(8);
            if (g == 5) AB_ST(11);
            zero_qa();
            sequence(std::true_type{}, std::true_type{}, 7 * g + 5, g + 1);
        }
        stage_and_attend(AB_GROUPS - 1);
        sequence(std::true_type{}, std::false_type{}, 7 * AB_GROUPS - 2, AB_GROUPS);
    }
    AB_ST(2);
#if defined(AB_STAMP2) || defined(AB_STAMP3)
    if (lane == 0 && (w & 3) == 0)
        for (int k = 0; k < 10; ++k) g_ab_stamp[((size_t)blockIdx.x * 2 + (w >> 2)) * 16 + k] = ts[k];
#endif
    // every wave's DMA (the three pad pieces included) has landed and every wave has left the ring before it is reused
    AB_WAIT(0);
    __syncthreads();

    // ---- epilogue: residual + LayerNorm (per token: the wave holds all 320 channels of its 16 tokens)
    const int token = 16 * w + l15;
    const int tsw = (token >> 1) & 7;
    char* xrow = smem + AB_X + token * 640 + (lq & 1) * 8;
    float s1 = 0.f, s2 = 0.f;
    static_for<0, 20>([&](auto j_) __attribute__((always_inline)) {
        constexpr int j = decltype(j_)::value;
        const int chunk = 2 * j + (lq >> 1);
        const int pos = (chunk & ~7) | ((chunk ^ tsw) & 7);
        const half4v xv = *reinterpret_cast<const half4v*>(xrow + pos * 16);
        static_for<0, 4>([&](auto r_) __attribute__((always_inline)) {
            constexpr int r = decltype(r_)::value;
            const float v = oc[j][r] + (float)xv[r];
            oc[j][r] = v;
            s1 += v; s2 += v * v;
        });
    });
    s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
    const float cnt = (float)a.ln_count;
    const float mean = s1 / cnt;
    float var = s2 / cnt - mean * mean;
    var = var > 0.f ? var : 0.f;
    const float rstd = rsqrtf(var + 1e-5f);
    const float* par = reinterpret_cast<const float*>(smem + AB_PAR);
    float2* scr = reinterpret_cast<float2*>(smem + AB_RING);            // [8 waves][20][4] GroupNorm partials
    float2* tot = scr + 8 * 20 * 4;                                     // [2 boards][20] (mean, rstd)
    const float nmr = -mean * rstd;
    static_for<0, 20>([&](auto j_) __attribute__((always_inline)) {
        constexpr int j = decltype(j_)::value;
        const float4 gm = *reinterpret_cast<const float4*>(par + 16 * j + 4 * lq);
        const float4 bt = *reinterpret_cast<const float4*>(par + 320 + 16 * j + 4 * lq);
        const float gmv[4] = {gm.x, gm.y, gm.z, gm.w}, btv[4] = {bt.x, bt.y, bt.z, bt.w};
        float p1 = 0.f, p2 = 0.f;
        half4v h;
        static_for<0, 4>([&](auto r_) __attribute__((always_inline)) {
            constexpr int r = decltype(r_)::value;
            const float y = fmaf(fmaf(oc[j][r], rstd, nmr), gmv[r], btv[r]);      // (v - mean) rstd gamma + beta, two FMAs
            p1 += y; p2 += y * y;
            h[r] = (_Float16)y;
        });
        const int chunk = 2 * j + (lq >> 1);
        const int pos = (chunk & ~7) | ((chunk ^ tsw) & 7);
        *reinterpret_cast<half4v*>(xrow + pos * 16) = h;                // over this lane's own x values
        p1 = ab_row_sum(p1); p2 = ab_row_sum(p2);
        if (l15 == 0) scr[(w * 20 + j) * 4 + lq] = make_float2(p1, p2);
    });
    // the wave's 16 rows are contiguous in the output: linear 16-byte reads of the LDS image, swizzle undone on the way
    auto flush = [&](_Float16* outp) __attribute__((always_inline)) {
        char* og = reinterpret_cast<char*>(outp) + (b0 * 64 + 16 * w) * 640;
#pragma unroll
        for (int n = 0; n < 10; ++n) {
            const int q = n * 64 + lane;
            const int rl = q / 40, pos = q - rl * 40;
            const int grow = 16 * w + rl;
            const int src = (pos & ~7) | ((pos ^ (grow >> 1)) & 7);
            const uint4 v = *reinterpret_cast<const uint4*>(smem + AB_X + grow * 640 + pos * 16);
            *reinterpret_cast<uint4*>(og + rl * 640 + src * 16) = v;
        }
    };
    AB_ST(3);
    flush(a.y);
    AB_ST(4);
    if (a.y2 == nullptr) return;
    // ---- second output: act(GroupNorm16(y)) for the next residual block (statistics per board and 16-channel group)
    __syncthreads();
    if (tid < 40) {
        const int bd = tid / 20, j = tid - bd * 20;
        float s = 0.f, ss = 0.f;
        for (int ww = 0; ww < 4; ++ww)
            for (int q = 0; q < 4; ++q) { const float2 v = scr[((bd * 4 + ww) * 20 + j) * 4 + q]; s += v.x; ss += v.y; }
        const float mu = s * (1.f / 1024.f);
        float vr = ss * (1.f / 1024.f) - mu * mu;
        vr = vr > 0.f ? vr : 0.f;
        tot[tid] = make_float2(mu, rsqrtf(vr + 1e-5f));
    }
    __syncthreads();
    // per (board, channel) scale and shift over the gamma / beta slots (the second GroupNorm's parameters are dead after this)
    {
        float* parw = reinterpret_cast<float*>(smem + AB_PAR);
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
    static_for<0, 20>([&](auto j_) __attribute__((always_inline)) {
        constexpr int j = decltype(j_)::value;
        const float4 gm = *reinterpret_cast<const float4*>(par + (w >> 2) * 640 + 16 * j + 4 * lq);
        const float4 bt = *reinterpret_cast<const float4*>(par + (w >> 2) * 640 + 320 + 16 * j + 4 * lq);
        const float gmv[4] = {gm.x, gm.y, gm.z, gm.w}, btv[4] = {bt.x, bt.y, bt.z, bt.w};
        const int chunk = 2 * j + (lq >> 1);
        const int pos = (chunk & ~7) | ((chunk ^ tsw) & 7);
        half4v h = *reinterpret_cast<const half4v*>(xrow + pos * 16);
        static_for<0, 4>([&](auto r_) __attribute__((always_inline)) {
            constexpr int r = decltype(r_)::value;
            h[r] = (_Float16)act_fast<ACT>((float)h[r] * gmv[r] + btv[r]);
        });
        *reinterpret_cast<half4v*>(xrow + pos * 16) = h;
    });
    AB_ST(9);
    flush(a.y2);
    AB_ST(10);
}

hipError_t launch_attn_block(const AttnBlockArgs& a, hipStream_t st) {
    if (a.B <= 0 || a.B % 2 != 0 || a.ln_count <= 0 || a.ln_count > 320) return hipErrorInvalidValue;
    if (a.y2 != nullptr && a.act != ACT_SILU && a.act != ACT_RELU) return hipErrorInvalidValue;
    static DeviceOnce once;
    hipError_t e = once.run([] {
        hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_block_kernel<ACT_SILU>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, AB_LDS);
        if (r != hipSuccess) return r;
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_block_kernel<ACT_RELU>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, AB_LDS);
    });
    if (e != hipSuccess) return e;
    const dim3 grid((unsigned)(a.B / 2));
    if (a.act == ACT_RELU) hipLaunchKernelGGL(attn_block_kernel<ACT_RELU>, grid, dim3(512), AB_LDS, st, a);
    else hipLaunchKernelGGL(attn_block_kernel<ACT_SILU>, grid, dim3(512), AB_LDS, st, a);
    return hipGetLastError();
}

size_t attn_block_pack_bytes() { return (size_t)(AB_GROUPS * AB_PIECES_PER_GROUP + 3) * AB_PIECE; }
