// attn_block3_kernel: one whole ChessAttention block of the tower (resnet.py:133-181: qkv 1x1 -> per-head scores / softmax / PV ->
// proj 1x1 -> residual add -> LayerNorm) plus the pre-activation GroupNorm of the residual block that follows, in ONE kernel for
// the 320-channel trunk -- round 4: 16 ROLE-SPECIALISED waves per board pair, four per SIMD.
//
// Round 3's kernel (attn_block.hip) ran every phase in all 8 waves at once -- qkv GEMM (matrix pipe + LDS reads), staging, softmax
// (VALU), proj GEMM -- and its ablations showed the parts to be additive: the matrix pipe idles during the softmax, the vector
// ALUs during the GEMMs.  A first role split with one attention wave per SIMD (tools/ubench/attn_block2.hip) was bit-identical but
// slower: a lone wave issues one vector instruction per 4 cycles and cannot hide its own latencies.  Here, per workgroup of
// 2 boards = 128 token rows:
//   waves 8-15 ("G", two per SIMD): the GEMMs.  A wave owns 16 token rows.  Their trunk values live in REGISTERS for the whole
//              kernel as the B fragments of the qkv GEMM (10 k-steps x 4 registers; the k order inside a k-step is permuted so
//              that the same registers are, element for element, the residual of the wave's accumulator tiles), which frees the
//              80 KB of LDS the trunk rows took.  Per period g: qkv GEMM of group g+1 ([16 x 320] x [320 x 96], 5 weight pieces of
//              64 k), Q, K (token-major) and V (transposed) to LDS as fp16.  After the last group: proj as ONE K = 320 GEMM from
//              the O buffer (20 pieces), accumulators initialised with the residual; LayerNorm; next block's GroupNorm + act.
//   waves 0-7  ("A", two per SIMD): the attention of group g = 2 heads x 2 boards, one (board, head, query half) per wave: S^T = K Q^T
//              and O^T = V^T P^T on MFMA 32x32x16, softmax arithmetic in between (relative-position bias in registers from a table
//              pre-arranged in accumulator order); O goes to the [128 tokens][320] O buffer (the LDS space the trunk rows vacated).
// So on every SIMD two waves' exp2 / clamp / mask arithmetic runs beside two waves' MFMA + fragment-read streams, at 128 registers
// per lane.  All waves meet at ONE barrier per weight piece (70 per board pair): the block's weights are one stream of 12 KB
// pieces in consumption order (qkv(0..9), then proj k-step 0..9 x 2 halves) through a 4-slot LDS ring filled three pieces ahead
// by global_load_lds from the G-waves (counted vmcnt: they have nothing else in flight, kernel_common.h); an A-wave's work of a
// period is cut into 5 chunks, one per piece of that period.
#include "../../matrix0_amd/csrc/kernel_common.h"
#include "../../matrix0_amd/csrc/conv_epilogue.h"

typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef float float4v __attribute__((ext_vector_type(4)));

namespace {
constexpr int A3_PIECE = 12288;
constexpr int A3_NPIECES = 70;
constexpr int A3_O = 0;                                   // [128][640 B], 16-byte chunk ^ (row>>1)&7 within 128 B; later y / y2
constexpr int A3_Q = 81920;                               // [2 heads][128 tokens][16] fp16; a token's two 16-byte halves at
                                                          // half ^ (token >> 3 & 1)
constexpr int A3_K = A3_Q + 8192;                         // same layout
constexpr int A3_VT = A3_K + 8192;                        // [4 units][16][68]
constexpr int A3_VROW = 68;
constexpr int A3_RING = A3_VT + 4 * 16 * A3_VROW * 2;     // 107008
constexpr int A3_PAR = A3_RING + 4 * A3_PIECE;            // 156160: LayerNorm gamma, beta, next GroupNorm gamma, beta [4][320] f32
constexpr int A3_LDS = A3_PAR + 4 * 320 * 4;              // 161280
constexpr int A3_THREADS = 1024;
}

__device__ __forceinline__ void a3_dma16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// 64 bytes per lane from global memory that the compiler does not track (the caller waits: vmcnt(0))
__device__ __forceinline__ void a3_load64(half8& b0, half8& b1, half8& b2, half8& b3, const half8* p) {
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\t"
                 "global_load_dwordx4 %2, %4, off offset:32\n\tglobal_load_dwordx4 %3, %4, off offset:48"
                 : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3) : "v"(p) : "memory");
}
// one 16-byte LDS read the compiler does not track (the caller waits: a3_arrived)
template <int OFF>
__device__ __forceinline__ void a3_lds16(half8& d, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
// ... and its wait: at most N younger LDS operations outstanding; the operand ties the first use to this point
template <int N>
__device__ __forceinline__ void a3_arrived(half8& f) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N) : "memory");
}
template <int CTRL>
__device__ __forceinline__ float a3_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float a3_row_sum(float v) {      // sum over the 16 lanes of a DPP row (every lane gets the total)
    v += a3_dpp<0xB1>(v);
    v += a3_dpp<0x4E>(v);
    v += a3_dpp<0x141>(v);
    v += a3_dpp<0x140>(v);
    return v;
}

#ifdef A3_STAMP
__device__ unsigned long long* g_a3_stamp;        // [blocks][2 roles][16] s_memtime stamps (tools/ubench/attn_block3_bench.hip)
#define A3_ST(role, k) do { if (lane == 0 && (role ? w == 8 : w == 0)) g_a3_stamp[((size_t)blockIdx.x * 2 + role) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define A3_ST(role, k) do {} while (0)
#endif
// A3_DRY: every barrier of the main loop becomes a counter (results are garbage): all 16 waves must report the same count before
// the real kernel is ever launched (a mismatch would hang the workgroup).
#ifdef A3_DRY
#define A3_BARRIER() do { ++nbar; } while (0)
#else
#define A3_BARRIER() __builtin_amdgcn_s_barrier()
#endif

template <int ACT>
__global__ __launch_bounds__(A3_THREADS) void attn_block3_kernel(AttnBlockArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);             // 0-7: attention waves, 8-15: GEMM waves
    const int l15 = lane & 15, lq = lane >> 4, r31 = lane & 31, half = lane >> 5;
    const size_t b0 = (size_t)blockIdx.x * 2;
    const char* xg = reinterpret_cast<const char*>(a.x) + b0 * 64 * 640;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    int nbar = 0;
    (void)nbar;
    A3_ST(0, 0); A3_ST(1, 0);

    if (w < 8) {
        // =====================================================================================================================
        // attention waves: unit (board, head-in-group) = w >> 1, query half = w & 1
        // =====================================================================================================================
        if (tid < 320) {
            float* par = reinterpret_cast<float*>(smem + A3_PAR);
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
        const char* const Kb = smem + A3_K + ahl * 4096 + aboard * 64 * 32;
        const char* const Qp = smem + A3_Q + ahl * 4096 + (aboard * 64 + aq) * 32 + hsw;
        const _Float16* const vrow = reinterpret_cast<const _Float16*>(smem + A3_VT) + (au * 16 + l15) * A3_VROW;
        const int orow = aboard * 64 + aq;
        const uint32_t obase = lds0 + A3_O + orow * 640;
        const int osw = (orow >> 1) & 7;

        // a piece boundary of these waves: their LDS reads / writes are done -> barrier (they move no weights)
        auto bnd = [&]() __attribute__((always_inline)) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            A3_BARRIER();
            asm volatile("" ::: "memory");
        };
        half8 bias8[4];
        float16v st[2];
        float e[2][16];
        // relative-position bias of (head, query half) in accumulator order: 64 B per lane, requested one period ahead (chunk 3 of
        // the period before: its registers are free once the scores are done) and waited for with vmcnt(0) -- these waves have no
        // other vector-memory operation in flight.  (inline asm: at the first use of an ordinary load's result hipcc waits
        // wherever that use lands)
        auto bias_request = [&](const int g) __attribute__((always_inline)) {
            const half8* bp = reinterpret_cast<const half8*>(a.bias) + ((size_t)((2 * g + ahl) * 2 + aqt) * 64 + lane) * 4;
            a3_load64(bias8[0], bias8[1], bias8[2], bias8[3], bp);
        };
        // the attention of group g in 5 chunks, a piece boundary in front of each
        auto attend = [&](const int g) __attribute__((always_inline)) {
#ifdef A3_NO_ATTN        // timing experiment: the attention waves only keep the piece cadence
            for (int i = 0; i < 5; ++i) bnd();
            return;
#endif
            bnd();
            {   // chunk 0: K and Q fragments, S^T = K Q^T
                const half8 kf0 = *reinterpret_cast<const half8*>(Kb + r31 * 32 + hsw);
                const half8 kf1 = *reinterpret_cast<const half8*>(Kb + (32 + r31) * 32 + hsw);
                const half8 qfr = *reinterpret_cast<const half8*>(Qp);
                st[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf0, qfr, zero16, 0, 0, 0);
                st[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf1, qfr, zero16, 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(bias8[0]), "+v"(bias8[1]), "+v"(bias8[2]), "+v"(bias8[3]) :: "memory");
            float su = 0.f, sm = 0.f;
            auto scores = [&](auto kt_, auto r0_, auto r1_) __attribute__((always_inline)) {
                constexpr int kt = decltype(kt_)::value;
                static_for<decltype(r0_)::value, decltype(r1_)::value>([&](auto r_) __attribute__((always_inline)) {
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
            using I8 = std::integral_constant<int, 8>; using I16 = std::integral_constant<int, 16>;
            scores(I0{}, I0{}, I8{});                                   // (still chunk 0)
            bnd();
            scores(I0{}, I8{}, I16{});                                  // chunk 1
            scores(I1{}, I0{}, I8{});
            bnd();
            scores(I1{}, I8{}, I16{});                                  // chunk 2
            su += __shfl_xor(su, 32);
            sm += __shfl_xor(sm, 32);
            const float cu = wu_ / su, cm = wm_ / sm;
            float16v oacc = zero16;
            half8 vf[2][2];
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
            bnd();
            // chunk 3.  ALL V^T fragments are read here: after the period's last boundary the G-waves write the next group's V^T
            static_for<0, 2>([&](auto kt_) __attribute__((always_inline)) {
                static_for<0, 2>([&](auto jb_) __attribute__((always_inline)) {
                    constexpr int kt = decltype(kt_)::value, jb = decltype(jb_)::value;
                    const half4v lo = *reinterpret_cast<const half4v*>(vrow + kt * 32 + 16 * jb + 4 * half);
                    const half4v hi = *reinterpret_cast<const half4v*>(vrow + kt * 32 + 16 * jb + 8 + 4 * half);
                    vf[kt][jb] = half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                });
            });
            if (g < 9) bias_request(g + 1);
            pv(I0{});
            bnd();
            pv(I1{});                                                   // chunk 4
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
        for (int i = 0; i < 5; ++i) bnd();                              // pieces 0-4: the G-waves compute qkv(0)
        A3_ST(0, 1);
#pragma unroll 1
        for (int g = 0; g < 10; ++g) {                                  // period g (pieces 5 g + 5 ...): attention of group g
            attend(g);
            if (g == 4) A3_ST(0, 2);
            if (g == 5) A3_ST(0, 3);
            if (g == 8) A3_ST(0, 4);
        }
        A3_ST(0, 5);
        for (int i = 0; i < 15; ++i) bnd();                             // the rest of the proj pieces
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        A3_BARRIER();                                                   // end of the main loop
        A3_ST(0, 6);
#ifdef A3_DRY
        if (lane == 0) reinterpret_cast<int*>(a.y)[blockIdx.x * 16 + w] = nbar;
        return;
#endif
        // the epilogue's barriers (the G-waves' __syncthreads below): 4 with a second output
        if (a.y2 != nullptr) {
            __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier();
        }
        return;
    }

    // =========================================================================================================================
    // GEMM waves: 16 token rows each
    // =========================================================================================================================
    const int gw = w - 8;
    const int gt = tid - 512;                                  // 0..511
    const int token = 16 * gw + l15;
    // the trunk rows of this wave as B fragments: k-step s, element e < 4: channel 32 s + 4 lq + e, e >= 4: 32 s + 16 + 4 lq + (e - 4)
    // (the weights are packed in the same k order) = the residual of accumulator tiles 2 s (e < 4) and 2 s + 1 (e >= 4)
    half8 Xf[10];
    {
        const char* xr = xg + (size_t)token * 640 + lq * 8;
        static_for<0, 10>([&](auto s_) __attribute__((always_inline)) {
            constexpr int s = decltype(s_)::value;
            const half4v lo = *reinterpret_cast<const half4v*>(xr + s * 64);
            const half4v hi = *reinterpret_cast<const half4v*>(xr + s * 64 + 32);
            Xf[s] = half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        });
    }
    // a piece is 12 x 1 KB: every G-wave issues one full 16-byte DMA and one with its upper 32 lanes masked off (1.5 KB per
    // wave), so the count of outstanding vector-memory operations is the same in all 8 waves
    const char* wsrc = reinterpret_cast<const char*>(a.wpack) + gw * 1536 + lane * 16;
    char* const ring_w = smem + A3_RING + gw * 1536;
    auto issue = [&](int t) __attribute__((always_inline)) {
        const char* s = wsrc + (size_t)t * A3_PIECE;
        char* d = ring_w + (t & 3) * A3_PIECE;
        a3_dma16(s, d);
        if (lane < 32) a3_dma16(s + 1024, d + 1024);
    };
    // the trunk rows must be in their registers before any COUNTED wait (kernel_common.h: register loads and LDS-DMA do not
    // retire in one order): wait for them here, then start the weight stream
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(Xf[0]), "+v"(Xf[1]), "+v"(Xf[2]), "+v"(Xf[3]), "+v"(Xf[4]), "+v"(Xf[5]), "+v"(Xf[6]),
                 "+v"(Xf[7]), "+v"(Xf[8]), "+v"(Xf[9]) :: "memory");
    issue(0); issue(1); issue(2);
    int ts = 0;                                                // next piece
    const uint32_t ring_a = lds0 + A3_RING;
    // AB_WAIT(4) = all but this wave's two youngest pieces have landed (two DMA instructions per piece)
    auto boundary = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(4)" ::: "memory");
        A3_BARRIER();
        asm volatile("" ::: "memory");
        issue(ts + 3);
    };

    const float4v zero4 = {0.f, 0.f, 0.f, 0.f};
    float4v qa[6];
    const int wsw = (l15 >> 1) & 7;
    const uint32_t wq0 = (uint32_t)(l15 * 128 + ((lq ^ wsw) & 7) * 16);            // qkv piece, k-step 0 of the piece
    const uint32_t wq1 = (uint32_t)(l15 * 128 + (((4 + lq) ^ wsw) & 7) * 16);      // k-step 1
    // qkv GEMM of one head group: 5 pieces x 2 k-steps x 6 channel tiles (q0 q1 k0 k1 v0 v1 of the two heads)
    // The fragments of k-step u+1 are read into the second register set before the MFMAs of k-step u are issued, also across a
    // piece boundary: boundary of piece i+1 (this wave's reads of piece i are complete, its part of piece i+1 has landed) -> read the
    // first k-step of piece i+1 -> DMA piece i+4 into the slot of piece i -> MFMAs of the last k-step of piece i.
    auto qkv_group = [&]() __attribute__((always_inline)) {
        static_for<0, 6>([&](auto j_) __attribute__((always_inline)) { qa[decltype(j_)::value] = zero4; });
#ifdef A3_NO_GEMM        // timing experiment: the GEMM waves only keep the piece cadence
        for (int i = 0; i < 5; ++i) { boundary(); ++ts; }
        return;
#endif
        half8 wf[2][6];
        uint32_t slot = 0;
        auto load = [&](auto u_) __attribute__((always_inline)) {
            constexpr int u = decltype(u_)::value, S = u & 1;
            const uint32_t wa = slot + ((u & 1) ? wq1 : wq0);
            static_for<0, 6>([&](auto j_) __attribute__((always_inline)) { constexpr int j = decltype(j_)::value; a3_lds16<j * 2048>(wf[S][j], wa); });
        };
        auto next_piece = [&]() __attribute__((always_inline)) {
            boundary();
            slot = ring_a + (uint32_t)((ts & 3) * A3_PIECE);
            ++ts;
        };
        next_piece();
        load(std::integral_constant<int, 0>{});
        static_for<0, 10>([&](auto u_) __attribute__((always_inline)) {
            constexpr int u = decltype(u_)::value, S = u & 1;
            if constexpr (u + 1 < 10) {
                if constexpr (u & 1) next_piece();            // (its lgkmcnt(0): the fragments of k-step u are in their registers)
                load(std::integral_constant<int, u + 1>{});
                if constexpr (!(u & 1)) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(wf[S][0]), "+v"(wf[S][1]), "+v"(wf[S][2]), "+v"(wf[S][3]), "+v"(wf[S][4]), "+v"(wf[S][5]) :: "memory");
                else asm volatile("" : "+v"(wf[S][0]), "+v"(wf[S][1]), "+v"(wf[S][2]), "+v"(wf[S][3]), "+v"(wf[S][4]), "+v"(wf[S][5]));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[S][0]), "+v"(wf[S][1]), "+v"(wf[S][2]), "+v"(wf[S][3]), "+v"(wf[S][4]), "+v"(wf[S][5]) :: "memory");
            }
            static_for<0, 6>([&](auto j_) __attribute__((always_inline)) {
                constexpr int j = decltype(j_)::value;
                qa[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[S][j], Xf[u], qa[j], 0, 0, 0);
            });
            __builtin_amdgcn_sched_barrier(0);                // the next k-step's wait stays behind these MFMAs
        });
    };
    // q, k (token-major) and v (transposed) of the group just computed, as fp16
    auto stage = [&]() __attribute__((always_inline)) {
        static_for<0, 6>([&](auto j_) __attribute__((always_inline)) {
            constexpr int J = decltype(j_)::value, type = J >> 1, hl = J & 1;
            const half4v h = {(_Float16)qa[J][0], (_Float16)qa[J][1], (_Float16)qa[J][2], (_Float16)qa[J][3]};
            if constexpr (type < 2) {
                *reinterpret_cast<half4v*>(smem + (type == 0 ? A3_Q : A3_K) + hl * 4096 + token * 32 + (((lq >> 1) ^ (l15 >> 3)) & 1) * 16 + (lq & 1) * 8) = h;
            } else {
                const int unit = (token >> 6) * 2 + hl, sq = token & 63;
                _Float16* vt = reinterpret_cast<_Float16*>(smem + A3_VT) + (unit * 16 + 4 * lq) * A3_VROW + sq;
                vt[0] = h[0]; vt[A3_VROW] = h[1]; vt[2 * A3_VROW] = h[2]; vt[3 * A3_VROW] = h[3];
            }
        });
    };

    qkv_group(); stage();                                      // pieces 0-4: qkv(0)
    A3_ST(1, 1);
#pragma unroll 1
    for (int g = 0; g < 9; ++g) {                              // period g: qkv(g + 1) while the A-waves attend to group g
        qkv_group(); stage();
        if (g == 4) A3_ST(1, 2);
        if (g == 5) A3_ST(1, 3);
    }
    A3_ST(1, 4);

    // ---- proj: out[16 tokens x 320] = x + O[16 x 320] Wproj^T, k-step g = the two heads of group g (the A-waves finish group 9
    // during the first pieces), accumulators start from the residual
    float4v oc[20];
    static_for<0, 20>([&](auto j_) __attribute__((always_inline)) {
        constexpr int j = decltype(j_)::value;
        static_for<0, 4>([&](auto r_) __attribute__((always_inline)) {
            constexpr int r = decltype(r_)::value;
            oc[j][r] = (float)Xf[j >> 1][(j & 1) * 4 + r];
        });
    });
    // proj piece: [160 channels][32 k] in 64-byte rows; a ds_read_b128 is served in lane groups {0-3, 12-15, 20-27}, {4-11, 16-19,
    // 28-31}, ...: chunk ^ (4 - quad) & 3 gives the 16 lanes of a group 16 different bank quads
    const uint32_t wpo = (uint32_t)(l15 * 64 + ((lq ^ (4 - (l15 >> 2))) & 3) * 16);
    const int tsw = (token >> 1) & 7;
    const uint32_t orow_a = lds0 + A3_O + token * 640;
#pragma unroll 1
    for (int g = 0; g < 10; ++g) {
        half8 of;
        static_for<0, 2>([&](auto hh_) __attribute__((always_inline)) {
            constexpr int hh = decltype(hh_)::value;
            boundary();
            const uint32_t pa = ring_a + (uint32_t)((ts & 3) * A3_PIECE) + wpo;
            ++ts;
            half8 pw[5];
            if constexpr (hh == 0) {
                const int c = 4 * g + lq;
                a3_lds16<0>(of, orow_a + (uint32_t)(((c & ~7) | ((c ^ tsw) & 7)) * 16));
            }
            // two rounds of 5 channel tiles through one fragment set (80 accumulators leave no room for a second)
            static_for<0, 2>([&](auto kk_) __attribute__((always_inline)) {
                constexpr int kk = decltype(kk_)::value;
                static_for<0, 5>([&](auto jj_) __attribute__((always_inline)) { constexpr int jj = decltype(jj_)::value; a3_lds16<(5 * kk + jj) * 1024>(pw[jj], pa); });
                static_for<0, 5>([&](auto jj_) __attribute__((always_inline)) {
                    constexpr int jj = decltype(jj_)::value;
                    constexpr int c = 10 * hh + 5 * kk + jj;
                    a3_arrived<4 - jj>(pw[jj]);
                    if constexpr (hh == 0 && kk == 0 && jj == 0) asm volatile("" : "+v"(of));      // read before pw[0]: there by now
                    oc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pw[jj], of, oc[c], 0, 0, 0);
                });
            });
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    A3_ST(1, 5);
    // every wave's DMA (the three pad pieces included) has landed and every wave has left the ring and the O buffer
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    A3_BARRIER();
    A3_ST(1, 6);
#ifdef A3_DRY
    if (lane == 0) reinterpret_cast<int*>(a.y)[blockIdx.x * 16 + w] = nbar;
    return;
#endif

    // ---- epilogue: LayerNorm (per token: the wave holds all 320 channels of its 16 tokens; the residual is already in)
    char* xrow = smem + A3_O + token * 640 + (lq & 1) * 8;
    float s1 = 0.f, s2 = 0.f;
    static_for<0, 20>([&](auto j_) __attribute__((always_inline)) {
        constexpr int j = decltype(j_)::value;
        static_for<0, 4>([&](auto r_) __attribute__((always_inline)) {
            constexpr int r = decltype(r_)::value;
            const float v = oc[j][r];
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
    const float* par = reinterpret_cast<const float*>(smem + A3_PAR);
    float2* scr = reinterpret_cast<float2*>(smem + A3_RING);            // [8 waves][20][4] GroupNorm partials
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
        *reinterpret_cast<half4v*>(xrow + pos * 16) = h;                // the O buffer is free: y image, same layout
        p1 = a3_row_sum(p1); p2 = a3_row_sum(p2);
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
            const uint4 v = *reinterpret_cast<const uint4*>(smem + A3_O + grow * 640 + pos * 16);
            *reinterpret_cast<uint4*>(og + rl * 640 + src * 16) = v;
        }
    };
    flush(a.y);
    A3_ST(1, 7);
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
        float* parw = reinterpret_cast<float*>(smem + A3_PAR);
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
    A3_ST(1, 8);
}

hipError_t launch_attn_block3(const AttnBlockArgs& a, hipStream_t st) {
    if (a.B <= 0 || a.B % 2 != 0 || a.ln_count <= 0 || a.ln_count > 320) return hipErrorInvalidValue;
    if (a.y2 != nullptr && a.act != ACT_SILU && a.act != ACT_RELU) return hipErrorInvalidValue;
    static DeviceOnce once;
    hipError_t e = once.run([] {
        hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_block3_kernel<ACT_SILU>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, A3_LDS);
        if (r != hipSuccess) return r;
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_block3_kernel<ACT_RELU>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, A3_LDS);
    });
    if (e != hipSuccess) return e;
    const dim3 grid((unsigned)(a.B / 2));
    if (a.act == ACT_RELU) hipLaunchKernelGGL(attn_block3_kernel<ACT_RELU>, grid, dim3(A3_THREADS), A3_LDS, st, a);
    else hipLaunchKernelGGL(attn_block3_kernel<ACT_SILU>, grid, dim3(A3_THREADS), A3_LDS, st, a);
    return hipGetLastError();
}

// 70 pieces in consumption order + 3 pad pieces (the last boundaries request three pieces past the end)
size_t attn_block3_pack_bytes() { return (size_t)(A3_NPIECES + 3) * A3_PIECE; }
// stream position of piece pc (0-4: qkv k-chunks of 64, 5-6: proj halves of 160 channels) of head group g
int attn_block3_stream_pos(int g, int pc) { return pc < 5 ? 5 * g + pc : 50 + 2 * g + (pc - 5); }
// channel that sits at k-slot kl (0..31) of a qkv k-step: slot (lq = kl >> 3, e = kl & 7) holds channel 4 lq + e (e < 4) or
// 16 + 4 lq + (e - 4) of the k-step's 32, matching the register-resident trunk fragments
int attn_block3_qkv_kperm(int kl) { const int lq = kl >> 3, e = kl & 7; return e < 4 ? 4 * lq + e : 16 + 4 * lq + (e - 4); }
