// attn_block2_kernel: one whole ChessAttention block of the tower (resnet.py:133-181: qkv 1x1 -> per-head scores / softmax / PV ->
// proj 1x1 -> residual add -> LayerNorm) plus the pre-activation GroupNorm of the residual block that follows, in ONE kernel for
// the 320-channel trunk -- round 4: ROLE-SPECIALISED waves.
//
// Round 3's kernel (attn_block.hip) ran every phase in all 8 waves at once: qkv GEMM (matrix pipe + LDS reads), staging, softmax
// (VALU), proj GEMM; its own ablations showed the parts to be additive -- the matrix pipe idles during the softmax, the vector
// ALUs during the GEMMs.  Here a workgroup (2 boards = 128 token rows) has 12 waves, three per SIMD:
//   waves 4-11 ("G", two per SIMD): the GEMMs.  Per period g:  proj of group g-1 (O[128 x 32] x Wproj[32 x 320], accumulated in 80
//              registers per lane: a wave owns 16 tokens x all 320 channels), then the qkv GEMM of group g+1
//              ([128 x 320] x [320 x 96]; a wave owns 32 tokens x 48 channels), then Q, K (token-major) and V (transposed) of
//              group g+1 to LDS as fp16.  After the last group: residual, LayerNorm, next block's GroupNorm + activation.
//   waves 0-3  ("A", one per SIMD): the attention of group g = 2 heads x 2 boards, one (board, head) unit per wave, all 64
//              queries in two halves: S^T = K Q^T and O^T = V^T P^T on MFMA 32x32x16, softmax arithmetic in between; the
//              relative-position bias arrives in registers from a table pre-arranged in accumulator order; O overwrites Q.
// So on every SIMD one wave's exp2 / clamp / mask arithmetic runs beside two waves' MFMA + fragment-read streams.
// All waves meet at ONE barrier per weight piece (70 per board pair): the weights of the block are one stream of 70 host-packed
// 12 KB pieces (in consumption order: qkv(0), qkv(1), [proj(g-1), qkv(g+1)] for g = 1..8, proj(8), proj(9)) through a 3-slot
// LDS ring filled two pieces ahead by global_load_lds, one 1 KB instruction per wave and piece.  An A-wave's work of a period is
// cut into 7 chunks, one per piece of that period.
// Buffers: the Q / O regions alternate (Q(g) and, over it, O(g) in region g & 1), K and V^T are single: an A-wave reads its K, V
// and Q fragments in the first chunk of its period, the G-waves write the next group's only after the period's last piece.
#include "../../matrix0_amd/csrc/kernel_common.h"
#include "../../matrix0_amd/csrc/conv_epilogue.h"

typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));

namespace {
constexpr int A2_PIECE = 12288;
constexpr int A2_NPIECES = 70;
constexpr int A2_SLOTS = 3;
constexpr int A2_X = 0;                                   // [128][640 B], 16-byte chunk ^ (row>>1)&7 within 128 B
constexpr int A2_QO = 81920;                              // 2 regions x [2 heads][128 tokens][16] fp16; a token's two 16-byte halves
                                                          // at half ^ (token >> 3 & 1)
constexpr int A2_K = A2_QO + 16384;                       // [2 heads][128 tokens][16], same row layout
constexpr int A2_VT = A2_K + 8192;                        // [4 units][16][68]
constexpr int A2_VROW = 68;
constexpr int A2_RING = A2_VT + 4 * 16 * A2_VROW * 2;     // 115200
constexpr int A2_PAR = A2_RING + A2_SLOTS * A2_PIECE;     // 152064: LayerNorm gamma, beta, next GroupNorm gamma, beta [4][320] f32
constexpr int A2_LDS = A2_PAR + 4 * 320 * 4;              // 157184
constexpr int A2_THREADS = 768;
}

__device__ __forceinline__ void a2_dma16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// 64 bytes per lane from global memory that the compiler does not track (the caller waits: vmcnt(0))
__device__ __forceinline__ void a2_load64(half8& b0, half8& b1, half8& b2, half8& b3, const half8* p) {
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\t"
                 "global_load_dwordx4 %2, %4, off offset:32\n\tglobal_load_dwordx4 %3, %4, off offset:48"
                 : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3) : "v"(p) : "memory");
}
// one 16-byte LDS read the compiler does not track (the caller waits: lgkmcnt)
template <int OFF>
__device__ __forceinline__ void a2_lds16(half8& d, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void a2_lds_arrived(half8& f0, half8& f1, half8& f2, half8& f3, half8& f4, half8& f5) {
    static_assert(N == 0 || N == 5, "");
    if constexpr (N == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5) :: "memory");
    else asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5) :: "memory");
}
template <int CTRL>
__device__ __forceinline__ float a2_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float a2_row_sum(float v) {
    v += a2_dpp<0xB1>(v);
    v += a2_dpp<0x4E>(v);
    v += a2_dpp<0x141>(v);
    v += a2_dpp<0x140>(v);
    return v;
}

#ifdef A2_STAMP
__device__ unsigned long long* g_a2_stamp;        // [blocks][2 roles][16] s_memtime stamps (tools/ubench/attn_block2_bench.hip)
#define A2_ST(role, k) do { if (lane == 0 && (role ? w == 4 : w == 0)) g_a2_stamp[((size_t)blockIdx.x * 2 + role) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define A2_ST(role, k) do {} while (0)
#endif
// A2_DRY: every s_barrier of the main loop is replaced by a counter (results are garbage): the counts of all 12 waves, written to
// y, must agree before the real kernel is ever launched (a mismatch would hang the workgroup).
#ifdef A2_DRY
#define A2_BARRIER() do { ++nbar; } while (0)
#else
#define A2_BARRIER() __builtin_amdgcn_s_barrier()
#endif

template <int ACT>
__global__ __launch_bounds__(A2_THREADS) void attn_block2_kernel(AttnBlockArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);             // 0-3: attention waves, 4-11: GEMM waves
    const int l15 = lane & 15, lq = lane >> 4, r31 = lane & 31, half = lane >> 5;
    const size_t b0 = (size_t)blockIdx.x * 2;
    const char* xg = reinterpret_cast<const char*>(a.x) + b0 * 64 * 640;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    int nbar = 0;
    (void)nbar;

    // ---- weight stream: piece t -> ring slot t % 3; every wave moves 1 KB of every piece
    const char* wsrc = reinterpret_cast<const char*>(a.wpack) + w * 1024 + lane * 16;
    char* const ring_w = smem + A2_RING + w * 1024;
    int ts = 0;                      // next piece whose boundary this wave executes
    int tslot = 0;                   // ts % 3
    auto issue_at = [&](int t, int slot) __attribute__((always_inline)) {
        a2_dma16(wsrc + (size_t)t * A2_PIECE, ring_w + slot * A2_PIECE);
    };
    auto slot_plus = [](int s, int d) __attribute__((always_inline)) { int r = s + d; return r >= A2_SLOTS ? r - A2_SLOTS : r; };

    // ---- prologue: the two boards' rows (80 pieces of 1 KB over the 12 waves), then weight pieces 0 and 1
    A2_ST(0, 0); A2_ST(1, 0);
#pragma unroll
    for (int n = 0; n < 7; ++n) {
        const int idx = w + 12 * n;
        if (idx < 80) {
            const int q = idx * 64 + lane;
            const int row = q / 40, pos = q - row * 40;
            const int src = (pos & ~7) | ((pos ^ (row >> 1)) & 7);
            a2_dma16(xg + row * 640 + src * 16, smem + A2_X + idx * 1024);
        }
    }
    issue_at(0, 0); issue_at(1, 1);

    if (w < 4) {
        // =====================================================================================================================
        // attention waves
        // =====================================================================================================================
        {   // parameters of the epilogue: LDS (ordinary loads: these waves wait vmcnt(0) at every boundary anyway)
            float* par = reinterpret_cast<float*>(smem + A2_PAR);
            for (int i = tid; i < 320; i += 256) {
                par[i] = a.ln_g[i]; par[320 + i] = a.ln_b[i];
                par[640 + i] = a.y2 ? a.gn2_gamma[i] : 0.f; par[960 + i] = a.y2 ? a.gn2_beta[i] : 0.f;
            }
        }
        const int au = w, aboard = au >> 1, ahl = au & 1;
        // visibility of key (kt, r) from this lane's query of each half, accumulator order: key = kt*32 + 8(r>>2) + 4 half + (r&3)
        half2v visp[2][16];
        static_for<0, 2>([&](auto qt_) __attribute__((always_inline)) {
            constexpr int qt = decltype(qt_)::value;
            const uint64_t m = a.mask[qt * 32 + r31];
            static_for<0, 32>([&](auto i_) __attribute__((always_inline)) {
                constexpr int i = decltype(i_)::value;
                constexpr int kt = i >> 4, r = i & 15;
                const int key = kt * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
                visp[qt][i >> 1][i & 1] = (_Float16)(float)((m >> key) & 1);
            });
        });
        float wm_, wu_;   // output weights of the masked / unmasked branch (resnet.py:154-174)
        if (a.mix > 0.f && a.mix < 1.f) { wm_ = 1.f - a.mix; wu_ = 1.f - (1.f - a.mix); }
        else if (a.mix >= 1.f) { wm_ = 1.f; wu_ = 0.f; }
        else { wm_ = 0.f; wu_ = 1.f; }
        const float isd = a.inv_sqrt_d * 1.44269504088896f;
        const float clampv = 50.f * 1.44269504088896f;
        const float16v zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int hsw = 16 * (half ^ ((r31 >> 3) & 1));                                     // this lane's half of its token's row

        // boundary of piece ts: everything this wave has in flight has landed (its register loads too: only vmcnt(0) is safe
        // with both kinds outstanding, kernel_common.h), its LDS writes are done -> barrier -> request piece ts + 2
        auto bnd = [&]() __attribute__((always_inline)) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            A2_BARRIER();
            asm volatile("" ::: "memory");
            issue_at(ts + 2, slot_plus(tslot, 2));
            ++ts; tslot = slot_plus(tslot, 1);
        };

        half8 kf0, kf1, qfr[2], vf[2][2], bias8[4];
        float16v st[2];
        float e[2][16];
        float su, sm, cu, cm;
        char *Qp0 = nullptr, *Qp1 = nullptr;
        auto bias_load = [&](int g, int qt) __attribute__((always_inline)) {
            const half8* bp = reinterpret_cast<const half8*>(a.bias) + ((size_t)((2 * g + ahl) * 2 + qt) * 64 + lane) * 4;
            a2_load64(bias8[0], bias8[1], bias8[2], bias8[3], bp);
        };
        auto scores = [&](auto qt_, auto kt_) __attribute__((always_inline)) {
            constexpr int qt = decltype(qt_)::value, kt = decltype(kt_)::value;
            static_for<0, 16>([&](auto r_) __attribute__((always_inline)) {
                constexpr int r = decltype(r_)::value;
                constexpr int bi = kt * 16 + r;
                float d = st[kt][r] * isd + (float)bias8[bi >> 3][bi & 7];
                d = __builtin_amdgcn_fmed3f(d, -clampv, clampv);
                const float eu = __builtin_amdgcn_exp2f(d);
                e[kt][r] = eu;
                su += eu;
                sm += eu * (float)visp[qt][bi >> 1][bi & 1];
            });
        };
        auto finish = [&](auto qt_) __attribute__((always_inline)) {       // P, O^T = V^T P^T, O over Q
            constexpr int qt = decltype(qt_)::value;
            float16v oacc = zero16;
            static_for<0, 2>([&](auto kt_) __attribute__((always_inline)) {
                static_for<0, 2>([&](auto jb_) __attribute__((always_inline)) {
                    constexpr int kt = decltype(kt_)::value, jb = decltype(jb_)::value;
                    half8 pf;
                    static_for<0, 8>([&](auto u_) __attribute__((always_inline)) {
                        constexpr int u = decltype(u_)::value;
                        constexpr int r = 8 * jb + u, bi = kt * 16 + r;
                        const float vis = (float)visp[qt][bi >> 1][bi & 1];
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
            const uint32_t oaddr = (uint32_t)(uintptr_t)(qt ? Qp1 : Qp0);
            asm volatile("ds_write_b128 %0, %1" :: "v"(oaddr), "v"(ov) : "memory");
        };
        // the attention of group g in 7 chunks; PLAN bit k: a piece boundary in front of chunk k
        auto attend = [&](auto plan_, const int g) __attribute__((always_inline)) {
            constexpr int PLAN = decltype(plan_)::value;
            static_assert(PLAN & 1, "a period starts with a boundary");
#ifdef A2_NO_ATTN        // timing experiment: the attention waves only keep the piece cadence
            static_for<0, 7>([&](auto k_) __attribute__((always_inline)) { if constexpr ((PLAN >> decltype(k_)::value) & 1) bnd(); });
            return;
#endif
            // chunk 0: fragments of this unit from LDS (K, V shared by the halves; Q of both), bias of half 0, S^T of half 0
            bnd();
            {
                const char* Kb = smem + A2_K + ahl * 4096 + aboard * 64 * 32;
                kf0 = *reinterpret_cast<const half8*>(Kb + r31 * 32 + hsw);
                kf1 = *reinterpret_cast<const half8*>(Kb + (32 + r31) * 32 + hsw);
                Qp0 = smem + A2_QO + (g & 1) * 8192 + ahl * 4096 + (aboard * 64 + r31) * 32 + hsw;
                Qp1 = Qp0 + 32 * 32;
                qfr[0] = *reinterpret_cast<const half8*>(Qp0);
                qfr[1] = *reinterpret_cast<const half8*>(Qp1);
                const _Float16* vrow = reinterpret_cast<const _Float16*>(smem + A2_VT) + (au * 16 + l15) * A2_VROW;
                static_for<0, 2>([&](auto kt_) __attribute__((always_inline)) {
                    static_for<0, 2>([&](auto jb_) __attribute__((always_inline)) {
                        constexpr int kt = decltype(kt_)::value, jb = decltype(jb_)::value;
                        const half4v lo = *reinterpret_cast<const half4v*>(vrow + kt * 32 + 16 * jb + 4 * half);
                        const half4v hi = *reinterpret_cast<const half4v*>(vrow + kt * 32 + 16 * jb + 8 + 4 * half);
                        vf[kt][jb] = half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    });
                });
                bias_load(g, 0);
                st[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf0, qfr[0], zero16, 0, 0, 0);
                st[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf1, qfr[0], zero16, 0, 0, 0);
            }
            if constexpr (PLAN & 2) bnd();
            else asm volatile("s_waitcnt vmcnt(0)" : "+v"(bias8[0]), "+v"(bias8[1]), "+v"(bias8[2]), "+v"(bias8[3]) :: "memory");
            asm volatile("" : "+v"(bias8[0]), "+v"(bias8[1]), "+v"(bias8[2]), "+v"(bias8[3]));      // uses stay behind the wait
            su = 0.f; sm = 0.f;
            scores(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            if constexpr (PLAN & 4) bnd();
            scores(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
            su += __shfl_xor(su, 32); sm += __shfl_xor(sm, 32);
            cu = wu_ / su; cm = wm_ / sm;
            if constexpr (PLAN & 8) bnd();
            finish(std::integral_constant<int, 0>{});
            bias_load(g, 1);
            st[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf0, qfr[1], zero16, 0, 0, 0);
            st[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf1, qfr[1], zero16, 0, 0, 0);
            if constexpr (PLAN & 16) bnd();
            else asm volatile("s_waitcnt vmcnt(0)" : "+v"(bias8[0]), "+v"(bias8[1]), "+v"(bias8[2]), "+v"(bias8[3]) :: "memory");
            asm volatile("" : "+v"(bias8[0]), "+v"(bias8[1]), "+v"(bias8[2]), "+v"(bias8[3]));
            su = 0.f; sm = 0.f;
            scores(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
            if constexpr (PLAN & 32) bnd();
            scores(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
            su += __shfl_xor(su, 32); sm += __shfl_xor(sm, 32);
            cu = wu_ / su; cm = wm_ / sm;
            if constexpr (PLAN & 64) bnd();
            finish(std::integral_constant<int, 1>{});
        };

        // pieces 0-4: the G-waves compute qkv(0)
        for (int i = 0; i < 5; ++i) bnd();
        A2_ST(0, 1);
        attend(std::integral_constant<int, 0x57>{}, 0);                 // period 0: 5 pieces (qkv(1)): chunks {0}{1}{2,3}{4,5}{6}
        A2_ST(0, 2);
#pragma unroll 1
        for (int g = 1; g <= 8; ++g) {
            attend(std::integral_constant<int, 0x7f>{}, g);             // periods 1-8: 7 pieces (proj(g-1), qkv(g+1))
            if (g == 4) A2_ST(0, 3);
            if (g == 5) A2_ST(0, 4);
        }
        A2_ST(0, 5);
        attend(std::integral_constant<int, 0x03>{}, 9);                 // period 9: 2 pieces (proj(8)): chunks {0}{1..6}
        A2_ST(0, 6);
        bnd(); bnd();                                                   // proj(9)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        A2_BARRIER();                                                   // end of the main loop
        A2_ST(0, 7);
#ifdef A2_DRY
        if (lane == 0) reinterpret_cast<int*>(a.y)[blockIdx.x * 12 + w] = nbar;
        return;
#endif
        // the epilogue's barriers (the G-waves' __syncthreads below): 4 with a second output
        if (a.y2 != nullptr) {
            __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier();
        }
        return;
    }

    // =========================================================================================================================
    // GEMM waves
    // =========================================================================================================================
    const int gw = w - 4;
    const int gt = tid - 256;                                  // 0..511
    const int wm = gw >> 1, wn = gw & 1;
    const int xrow0 = 32 * wm + l15;
    const int xsw = (xrow0 >> 1) & 7;
    const int xe0 = ((lq ^ xsw) & 7) * 16, xe1 = (((4 + lq) ^ xsw) & 7) * 16;
    const uint32_t xa[2] = {lds0 + A2_X + xrow0 * 640 + xe0, lds0 + A2_X + xrow0 * 640 + xe1};      // rows xrow0 and (+16 * 640) xrow0 + 16
    const int wsw = (l15 >> 1) & 7;
    const int wq0 = (3 * wn * 16 + l15) * 128 + ((lq ^ wsw) & 7) * 16;            // qkv piece, kk = 0
    const int wq1 = (3 * wn * 16 + l15) * 128 + (((4 + lq) ^ wsw) & 7) * 16;      // kk = 1
    // proj piece: 64-byte rows; a ds_read_b128 is served in lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (one row
    // quad of lq = 0 / 2 next to two of lq = 1 / 3): chunk ^ (4 - quad) & 3 gives the 16 lanes of a group 16 different bank quads
    const int wpo = l15 * 64 + ((lq ^ (4 - (l15 >> 2))) & 3) * 16;
    const uint32_t ring_a = lds0 + A2_RING;
    // this wave's O rows (proj operand) in region 0; region 1 is 8192 further
    const uint32_t of_a0 = lds0 + A2_QO + (lq >> 1) * 4096 + (16 * gw + l15) * 32 + ((lq & 1) ^ (l15 >> 3)) * 16;

    const float4v zero4 = {0.f, 0.f, 0.f, 0.f};
    float4v oc[20];
    static_for<0, 20>([&](auto j_) __attribute__((always_inline)) { oc[decltype(j_)::value] = zero4; });
    half8 fs[2][5];
    half8 of;
    float4v qa[2][3];

    // One sequence = the pieces of one period: [proj piece 0, 1 of group gp,] [qkv piece 0..4 of the group being projected two
    // later], each piece in two halves (one k-step of the qkv GEMM / 80 output channels of the proj).  The fragments of half u+1
    // are read into the second register set before the MFMAs of half u are issued.  Per piece boundary: this wave's reads of
    // piece i are complete (lgkmcnt) and its part of piece i+1 has landed (vmcnt(1): one DMA instruction per wave and piece,
    // two pieces ahead) -> barrier -> read the first half of piece i+1 -> DMA piece i+3 into the slot of piece i -> MFMAs of the
    // last half of i.  `oreg` = region of the O rows the proj pieces multiply.
    auto sequence = [&](auto hp_, auto hq_, const int oreg) __attribute__((always_inline)) {
        constexpr bool HP = decltype(hp_)::value, HQ = decltype(hq_)::value;
        constexpr int NP = (HP ? 2 : 0) + (HQ ? 5 : 0), NU = 2 * NP;
        const int s0 = tslot;                                 // slot of this sequence's first piece
        const uint32_t of_a = of_a0 + (uint32_t)oreg * 8192u;
        auto load = [&](auto u_) __attribute__((always_inline)) {
            constexpr int u = decltype(u_)::value, S = u & 1, i = u >> 1, kk = u & 1;
            const uint32_t slot = ring_a + (uint32_t)(slot_plus(s0, i % A2_SLOTS) * A2_PIECE);
            if constexpr (HP && i < 2) {
                if constexpr (u == 0) a2_lds16<0>(of, of_a);
                const uint32_t pa = slot + wpo;
                static_for<0, 5>([&](auto jj_) __attribute__((always_inline)) {
                    constexpr int jj = decltype(jj_)::value;
                    a2_lds16<(5 * kk + jj) * 1024>(fs[S][jj], pa);
                });
            } else {
                constexpr int p = i - (HP ? 2 : 0);
                a2_lds16<128 * p>(fs[S][3], xa[kk]);
                a2_lds16<128 * p + 16 * 640>(fs[S][4], xa[kk]);
                const uint32_t wa = slot + (kk ? wq1 : wq0);
                static_for<0, 3>([&](auto j_) __attribute__((always_inline)) {
                    constexpr int j = decltype(j_)::value;
                    a2_lds16<j * 2048>(fs[S][j], wa);
                });
            }
        };
        auto arrived = [&](auto u_, auto n_) __attribute__((always_inline)) {
            constexpr int S = decltype(u_)::value & 1;
            a2_lds_arrived<decltype(n_)::value>(fs[S][0], fs[S][1], fs[S][2], fs[S][3], fs[S][4], of);
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
        auto boundary = [&]() __attribute__((always_inline)) {
            asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            A2_BARRIER();
            asm volatile("" ::: "memory");
        };
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's staging writes of the previous period are in LDS
#ifdef A2_NO_GEMM        // timing experiment: the GEMM waves only keep the piece cadence
        static_for<0, NP>([&](auto i_) __attribute__((always_inline)) {
            boundary();
            issue_at(ts + decltype(i_)::value + 2, slot_plus(tslot, (decltype(i_)::value + 2) % A2_SLOTS));
        });
        ts += NP; tslot = slot_plus(tslot, NP % A2_SLOTS);
        return;
#endif
        boundary();
        load(std::integral_constant<int, 0>{});
        issue_at(ts + 2, slot_plus(tslot, 2));
        static_for<0, NU>([&](auto u_) __attribute__((always_inline)) {
            constexpr int u = decltype(u_)::value;
            if constexpr (u + 1 < NU) {
                if constexpr (u & 1) {
                    arrived(u_, std::integral_constant<int, 0>{});
                    boundary();
                }
                load(std::integral_constant<int, u + 1>{});
                if constexpr (u & 1) issue_at(ts + (u >> 1) + 3, slot_plus(tslot, ((u >> 1) + 3) % A2_SLOTS));
                else arrived(u_, std::integral_constant<int, 5>{});
            } else {
                arrived(u_, std::integral_constant<int, 0>{});
            }
            mma(u_);
            __builtin_amdgcn_sched_barrier(0);                // the next half's wait stays behind these MFMAs
        });
        ts += NP; tslot = slot_plus(tslot, NP % A2_SLOTS);
    };
    auto zero_qa = [&]() __attribute__((always_inline)) {
        static_for<0, 2>([&](auto i_) __attribute__((always_inline)) {
            static_for<0, 3>([&](auto j_) __attribute__((always_inline)) { qa[decltype(i_)::value][decltype(j_)::value] = zero4; });
        });
    };
    // q, k (token-major) and v (transposed) of the group just computed, as fp16; Q into region `qreg`
    auto stage = [&](const int qreg) __attribute__((always_inline)) {
        static_for<0, 2>([&](auto i_) __attribute__((always_inline)) {
            static_for<0, 3>([&](auto j_) __attribute__((always_inline)) {
                constexpr int i = decltype(i_)::value, j = decltype(j_)::value;
                const int J = 3 * wn + j, type = J >> 1, hl = J & 1;          // wave-uniform
                const int token = 32 * wm + 16 * i + l15;
                const half4v h = {(_Float16)qa[i][j][0], (_Float16)qa[i][j][1], (_Float16)qa[i][j][2], (_Float16)qa[i][j][3]};
                if (type < 2) {
                    char* base = type == 0 ? smem + A2_QO + qreg * 8192 : smem + A2_K;
                    *reinterpret_cast<half4v*>(base + hl * 4096 + token * 32 + (((lq >> 1) ^ (l15 >> 3)) & 1) * 16 + (lq & 1) * 8) = h;
                } else {
                    const int unit = (token >> 6) * 2 + hl, sq = token & 63;
                    _Float16* vt = reinterpret_cast<_Float16*>(smem + A2_VT) + (unit * 16 + 4 * lq) * A2_VROW + sq;
                    vt[0] = h[0]; vt[A2_VROW] = h[1]; vt[2 * A2_VROW] = h[2]; vt[3 * A2_VROW] = h[3];
                }
            });
        });
    };

    zero_qa();
    sequence(std::false_type{}, std::true_type{}, 0);          // pieces 0-4: qkv(0)
    stage(0);
    A2_ST(1, 1);
    zero_qa();
    sequence(std::false_type{}, std::true_type{}, 0);          // period 0, pieces 5-9: qkv(1)
    stage(1);
    A2_ST(1, 2);
#pragma unroll 1
    for (int g = 1; g <= 8; ++g) {                             // period g: proj(g-1) from O(g-1), qkv(g+1)
        zero_qa();
        sequence(std::true_type{}, std::true_type{}, (g - 1) & 1);
        stage((g + 1) & 1);
        if (g == 4) A2_ST(1, 3);
        if (g == 5) A2_ST(1, 4);
    }
    A2_ST(1, 5);
    sequence(std::true_type{}, std::false_type{}, 0);          // period 9: proj(8)
    A2_ST(1, 6);
    sequence(std::true_type{}, std::false_type{}, 1);          // proj(9)
    // every wave's DMA (the two pad pieces included) has landed and every wave has left the ring before it is reused
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    A2_BARRIER();
    A2_ST(1, 7);
#ifdef A2_DRY
    if (lane == 0) reinterpret_cast<int*>(a.y)[blockIdx.x * 12 + w] = nbar;
    return;
#endif

    // ---- epilogue: residual + LayerNorm (per token: the wave holds all 320 channels of its 16 tokens)
    const int token = 16 * gw + l15;
    const int tsw = (token >> 1) & 7;
    char* xrow = smem + A2_X + token * 640 + (lq & 1) * 8;
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
    const float* par = reinterpret_cast<const float*>(smem + A2_PAR);
    float2* scr = reinterpret_cast<float2*>(smem + A2_RING);            // [8 waves][20][4] GroupNorm partials
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
        p1 = a2_row_sum(p1); p2 = a2_row_sum(p2);
        if (l15 == 0) scr[(gw * 20 + j) * 4 + lq] = make_float2(p1, p2);
    });
    // the wave's 16 rows are contiguous in the output: linear 16-byte reads of the LDS image, swizzle undone on the way
    auto flush = [&](_Float16* outp) __attribute__((always_inline)) {
        char* og = reinterpret_cast<char*>(outp) + (b0 * 64 + 16 * gw) * 640;
#pragma unroll
        for (int n = 0; n < 10; ++n) {
            const int q = n * 64 + lane;
            const int rl = q / 40, pos = q - rl * 40;
            const int grow = 16 * gw + rl;
            const int src = (pos & ~7) | ((pos ^ (grow >> 1)) & 7);
            const uint4 v = *reinterpret_cast<const uint4*>(smem + A2_X + grow * 640 + pos * 16);
            *reinterpret_cast<uint4*>(og + rl * 640 + src * 16) = v;
        }
    };
    flush(a.y);
    A2_ST(1, 8);
    if (a.y2 == nullptr) return;
    // ---- second output: act(GroupNorm16(y)) for the next residual block (statistics per board and 16-channel group)
    __syncthreads();
    if (gt < 40) {
        const int bd = gt / 20, j = gt - bd * 20;
        float s = 0.f, ss = 0.f;
        for (int ww = 0; ww < 4; ++ww)
            for (int q = 0; q < 4; ++q) { const float2 v = scr[((bd * 4 + ww) * 20 + j) * 4 + q]; s += v.x; ss += v.y; }
        const float mu = s * (1.f / 1024.f);
        float vr = ss * (1.f / 1024.f) - mu * mu;
        vr = vr > 0.f ? vr : 0.f;
        tot[gt] = make_float2(mu, rsqrtf(vr + 1e-5f));
    }
    __syncthreads();
    // per (board, channel) scale and shift over the gamma / beta slots (the second GroupNorm's parameters are dead after this)
    {
        float* parw = reinterpret_cast<float*>(smem + A2_PAR);
        float scv[2] = {0.f, 0.f}, shv[2] = {0.f, 0.f};
        if (gt < 320) {
            const float g2 = parw[640 + gt], b2 = parw[960 + gt];
#pragma unroll
            for (int bd = 0; bd < 2; ++bd) {
                const float2 mr = tot[bd * 20 + (gt >> 4)];
                scv[bd] = g2 * mr.y; shv[bd] = b2 - mr.x * scv[bd];
            }
        }
        __syncthreads();
        if (gt < 320) { parw[gt] = scv[0]; parw[320 + gt] = shv[0]; parw[640 + gt] = scv[1]; parw[960 + gt] = shv[1]; }
        __syncthreads();
    }
    static_for<0, 20>([&](auto j_) __attribute__((always_inline)) {
        constexpr int j = decltype(j_)::value;
        const float4 gm = *reinterpret_cast<const float4*>(par + (gw >> 2) * 640 + 16 * j + 4 * lq);
        const float4 bt = *reinterpret_cast<const float4*>(par + (gw >> 2) * 640 + 320 + 16 * j + 4 * lq);
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
    flush(a.y2);
    A2_ST(1, 9);
}

hipError_t launch_attn_block2(const AttnBlockArgs& a, hipStream_t st) {
    if (a.B <= 0 || a.B % 2 != 0 || a.ln_count <= 0 || a.ln_count > 320) return hipErrorInvalidValue;
    if (a.y2 != nullptr && a.act != ACT_SILU && a.act != ACT_RELU) return hipErrorInvalidValue;
    static DeviceOnce once;
    hipError_t e = once.run([] {
        hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_block2_kernel<ACT_SILU>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, A2_LDS);
        if (r != hipSuccess) return r;
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_block2_kernel<ACT_RELU>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, A2_LDS);
    });
    if (e != hipSuccess) return e;
    const dim3 grid((unsigned)(a.B / 2));
    if (a.act == ACT_RELU) hipLaunchKernelGGL(attn_block2_kernel<ACT_RELU>, grid, dim3(A2_THREADS), A2_LDS, st, a);
    else hipLaunchKernelGGL(attn_block2_kernel<ACT_SILU>, grid, dim3(A2_THREADS), A2_LDS, st, a);
    return hipGetLastError();
}

// 70 pieces in consumption order + 2 pad pieces (the last boundaries request two pieces past the end)
size_t attn_block2_pack_bytes() { return (size_t)(A2_NPIECES + 2) * A2_PIECE; }
// stream position of the packed piece (group g, piece pc of that group's seven: 0-4 qkv, 5-6 proj)
int attn_block2_stream_pos(int g, int pc) {
    if (pc < 5) return g == 0 ? pc : (g == 1 ? 5 + pc : 10 + 7 * (g - 2) + 2 + pc);     // qkv(g) rides in period g - 1
    const int hh = pc - 5;
    return g <= 7 ? 10 + 7 * g + hh : (g == 8 ? 66 + hh : 68 + hh);                     // proj(g) rides in period g + 1
}
