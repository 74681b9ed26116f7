// EXPERIMENT (round 4, measured slower than conv_zs_kernel, not in the library: profiles/r04_exp_conv_square_tiles.log).
// conv_zd_kernel: conv1 of a residual block (3x3, 320 -> 320, GroupNorm + activation epilogue) with ALL of the zero padding skipped.
//
// conv_zs_kernel (read its header first: same MFMA, same LDS-DMA ring, same ping-pong groups and barrier protocol) makes an M-tile
// out of one board row of two boards, so the tiles that only see the padding row above / below the board are not issued: 6 of the
// 72 (tap, tile) pairs.  The padding COLUMNS (one lane in eight for the six taps with dx != 0) are still multiplied.  The main loop
// is power-bound: matrix work that is not issued is the only thing that has ever turned into time there.
//
// Here an M-tile is ONE SQUARE of SIXTEEN boards.  For tap (dy, dx) the tile of square (y, x) multiplies the padding iff
// (y + dy, x + dx) is off the board -- then the whole tile is skipped: 92 of the 576 (square, tap) pairs = 16 % of the matrix work
// instead of 8.3 %, and no lane ever reads a zero region.  Workgroup = 16 boards x 64 squares x 80 output channels (one channel
// quarter; 1024 x 80 outputs = the 256 x 320 of conv_zs, same 8 x 5 accumulator tiles per wave).  Wave w owns the wrapped
// diagonal {(y, (y + w) & 7) : y = 0..7}: one square of every board row and of every board column, so for every tap all eight
// waves skip the same number of tiles (one for the four edge taps, two for the corner taps, less one where both fall on a corner
// square): M-tile t = board row y = t, so the dy skips are compile-time as in conv_zs, and the dx skip is ONE wave-uniform tile
// index per wave (a scalar branch around that tile's five MFMAs).
//
// The price: a workgroup sees only 80 of the 320 output channels, so the four channel quarters of a 16-board tile each read
// the tile's activations (4x the L2 -> LDS activation bytes, 1/4 of the weight bytes: 1.1 MB per workgroup instead of 2.0 MB; the
// four quarters get consecutive slots of one XCD's dispatch order, so HBM is read once), and anything that needs all 320
// channels of a board -- the squeeze-excite gate of the fused tail -- cannot live in this kernel: conv2 + tail stays on
// conv_zs_kernel.  GroupNorm groups (16 channels) are inside a quarter; their 64 squares are spread over the eight waves, so
// the statistics take one exchange through LDS.
//
// LDS: weight ring [4][80 ch x 32 k] (64-byte rows, 5 KB per half-tile of this quarter) | activations [2][64 squares][16 boards]
// [32 channels] (64-byte rows, 64 KB per 32-channel chunk, double-buffered).  Both row kinds are read with ds_read_b128 by lane
// (c15 = row, q = 16-byte chunk) at chunk q ^ ((4 - (c15 >> 2)) & 3): the lane groups a b128 read is served in ({0-3, 12-15,
// 20-27}, ...: MI355X_MICROARCH.md) then touch 16 different bank quads.  K order: 32-channel chunk, tap (conv_zs: 64-channel
// chunk, tap, half): results differ from conv_zs_kernel's in the last bits, are deterministic and do not depend on the
// other boards of a launch.  Weights: conv_zs's half-tile layout (GemmArgs::w_pp), rows [80 qr, 80 qr + 80) of every half-tile.
#include "../../matrix0_amd/csrc/kernel_common.h"
#include "../../matrix0_amd/csrc/conv_epilogue.h"

namespace {
constexpr int ZD_WH_BYTES = 80 * 64;                    // this quarter's rows of a half-tile
constexpr int ZD_FULL_HALF = 320 * 64;                  // a packed half-tile (all 320 channels)
constexpr int ZD_OFF_A = 4 * ZD_WH_BYTES;               // 20,480: in front of the activations (the tap offsets reach 9 KB back)
constexpr int ZD_A_BYTES = 64 * 1024;
constexpr int ZD_OFF_S = ZD_OFF_A + 2 * ZD_A_BYTES;     // 151,552: [8 waves][16 boards][5 groups] float2, epilogue only
constexpr int ZD_LDS = 160 * 1024;                      // the epilogue stages the whole tile: 8 x 20 KB
typedef float zd_float4v __attribute__((ext_vector_type(4)));
typedef _Float16 zd_half4 __attribute__((ext_vector_type(4)));
}

__device__ __forceinline__ void zd_glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
#define ZD_FENCE() asm volatile("" ::: "memory")

// The five MFMAs of tile T (weights as operand A) unless T is the wave's tile on the padding column of this tap (wave-uniform `tsk`)
template <int T>
__device__ __forceinline__ void zd_mma5(zd_float4v (&c)[5], const half8 (&b)[5], const half8& a, int tsk) {
#ifdef ZD_NOSKIP    // timing experiment: no branch, the padding column's tile multiplied too (results wrong)
#pragma unroll
    for (int i = 0; i < 5; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[i], a, c[i], 0, 0, 0);
    return;
#endif
    asm volatile(
        "s_cmp_eq_u32 %11, %12\n\t"
        "s_cbranch_scc1 .Lzd_skip%=\n\t"
        "v_mfma_f32_16x16x32_f16 %0, %5, %10, %0\n\t"
        "v_mfma_f32_16x16x32_f16 %1, %6, %10, %1\n\t"
        "v_mfma_f32_16x16x32_f16 %2, %7, %10, %2\n\t"
        "v_mfma_f32_16x16x32_f16 %3, %8, %10, %3\n\t"
        "v_mfma_f32_16x16x32_f16 %4, %9, %10, %4\n"
        ".Lzd_skip%=:"
        : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4])
        : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(a), "s"(tsk), "n"(T)
        : "scc");
}

#ifdef SW_STAMP
__device__ unsigned long long* g_zd_stamp;      // [blocks][8]: entry, loop start, loop end, exit (s_memrealtime), loop cycles, hw id, xcc
#endif

template <int ACT>
__global__ __launch_bounds__(512) void conv_zd_kernel(GemmArgs a) {
    constexpr int NG = 5, MT = 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* W_lds = smem;
    char* A_lds = smem + ZD_OFF_A;

#ifdef SW_STAMP
    const unsigned long long st_entry = __builtin_amdgcn_s_memrealtime();
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wave >> 2;                     // ping-pong group
    // workgroup ids go to the XCDs round-robin: the four quarters of a tile are ids 8 apart = neighbours in ONE XCD's order
    const int jx = blockIdx.x >> 3;
    const int qr = jx & 3;                        // channel quarter
    const int tile = (jx >> 2) * 8 + (blockIdx.x & 7);
    const int boards = a.Mvalid >> 6;
    if (tile * 16 >= boards) return;
    const int b0 = tile * 16;
    const int Cin = a.Cin;
    const int nch = Cin >> 5;                     // 32-channel chunks
    const int c15 = lane & 15;
    const int q = lane >> 4;

    // ---- DMA constants.  Activation piece s = square s of the 16 boards (1 KB): lane i -> board i >> 2, slot i & 3 holds channel
    // chunk (i & 3) ^ key(board).  Boards beyond the batch re-read the last valid board (their results are never stored).
    const char* in_bytes = reinterpret_cast<const char*>(a.in);
    const int dboard = lane >> 2;
    int bsrc = b0 + dboard;
    bsrc = bsrc < boards ? bsrc : boards - 1;
    const uint32_t a_glane = (uint32_t)bsrc * 64u * (uint32_t)Cin * 2u + 16u * (uint32_t)((lane & 3) ^ ((4 - (dboard >> 2)) & 3));
    const uint32_t row_bytes = (uint32_t)Cin * 2u;
    // weights: 640 bytes of the quarter's 5 KB per wave (lanes 0-39)
    const char* w_q = reinterpret_cast<const char*>(a.w) + (size_t)qr * ZD_WH_BYTES + wave * 640;
    const uint32_t w_lane = (uint32_t)lane * 16u;
    auto w_off = [&](int c32, int tap) -> size_t {      // half-tile of (32-channel chunk, tap) in conv_zs's order (64-chunk, tap, half)
        return (size_t)(2 * ((c32 >> 1) * 9 + tap) + (c32 & 1)) * ZD_FULL_HALF;
    };
    auto issue_w = [&](int c32, int tap, int slot) __attribute__((always_inline)) {
        if (lane < 40) zd_glds16(w_q + w_off(c32, tap) + w_lane, W_lds + slot * ZD_WH_BYTES + wave * 640);
    };
    auto issue_a = [&](int c32, int s, int buf) __attribute__((always_inline)) {
        const char* src = in_bytes + (size_t)c32 * 64 + (size_t)s * row_bytes;
        zd_glds16(src + a_glane, A_lds + buf * ZD_A_BYTES + s * 1024);
    };

    zd_float4v acc[MT][NG];
    static_for<0, MT>([&](auto mi) __attribute__((always_inline)) {
        static_for<0, NG>([&](auto ni) __attribute__((always_inline)) {
            acc[decltype(mi)::value][decltype(ni)::value] = zd_float4v{0.f, 0.f, 0.f, 0.f};
        });
    });

    // ---- per-lane constants of the fragment reads
    const int key = (4 - (c15 >> 2)) & 3;
    const int frag_lane = c15 * 64 + 16 * (q ^ key);                    // same row shape for weights and activations
    // tile t = square (t, (t + wave) & 7); the tap offsets (dy * 8 + dx) KB are added as immediates on top of a base 9 KB back
    int tsq[MT];
    static_for<0, MT>([&](auto t_) __attribute__((always_inline)) {
        constexpr int t = decltype(t_)::value;
        tsq[t] = (t * 8 + ((t + wave) & 7) - 9) * 1024;
    });
    const int tsk_m = (8 - wave) & 7;             // the tile on column 0: skipped for dx = -1
    const int tsk_p = (7 - wave) & 7;             // the tile on column 7: skipped for dx = +1

    // ---- prologue: chunk 0 (8 pieces per wave), half-tiles 0..2
#pragma unroll
    for (int i = 0; i < 8; ++i) issue_a(0, 8 * i + wave, 0);
    issue_w(0, 0, 0);
    issue_w(0, 1, 1);
    issue_w(0, 2, 2);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");    // the activations and half-tile 0; 1 and 2 are retired by the loop's counted waits
    __builtin_amdgcn_s_barrier();
    if (wp == 1) __builtin_amdgcn_s_barrier();          // group 1 runs one barrier behind group 0

    int y = 0;                                    // step = (chunk, tap); ring slot y & 3
    int wc = 0, wt = 3;                           // (chunk, tap) of the next half-tile to request (three ahead)
#ifdef SW_STAMP
    const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll 1
    for (int c = 0; c < nch; ++c) {
        const char* Ab = A_lds + (c & 1) * ZD_A_BYTES + frag_lane;
        const int cn = c + 1 < nch ? c + 1 : nch - 1;     // the last chunk re-requests itself into the idle buffer (same wait counts)
        const int nbuf = (c + 1) & 1;
        static_for<0, 3>([&](auto t3_) __attribute__((always_inline)) {
            constexpr int dy = decltype(t3_)::value - 1;
            constexpr int LO = dy == -1 ? 1 : 0;          // first / one-past-last M-tile with an operand on the board
            constexpr int HI = dy == 1 ? 7 : 8;
#pragma unroll 1
            for (int dxi = 0; dxi < 3; ++dxi) {           // tap = 3 (dy + 1) + dxi
                const char* Ad = Ab + (8 + dy * 8 + dxi) * 1024;                // base 9 KB back + (9 + 8 dy + dx) KB
                const int tsk = dxi == 0 ? tsk_m : (dxi == 2 ? tsk_p : 8);      // the tile on the padding column of this tap (8: none)
                half8 fa[MT], fb[NG];
                const char* Wb = W_lds + (y & 3) * ZD_WH_BYTES + frag_lane;
                static_for<LO, HI>([&](auto t_) __attribute__((always_inline)) {
                    constexpr int t = decltype(t_)::value;
                    // (the tile on the padding column is read too -- from the neighbouring row's other end, valid LDS -- and not used)
                    fa[t] = *reinterpret_cast<const half8*>(Ad + tsq[t]);
                });
                static_for<0, NG>([&](auto ni_) __attribute__((always_inline)) {
                    constexpr int ni = decltype(ni_)::value;
                    fb[ni] = *reinterpret_cast<const half8*>(Wb + ni * 1024);
                });
                // this wave's DMA of the step goes out in the load section (conv_zs.hip: an LDS-DMA instruction costs 40-60 cycles of
                // a wave's issue stream, here they fall under the partner group's MFMAs): one activation piece of the next chunk
                // (taps 0-7: square 8 tap + wave) and its 640 bytes of the half-tile three steps ahead
                ZD_FENCE();
                const bool last_tap = dy == 1 && dxi == 2;
#ifndef ZD_NO_ADMA
                if (!last_tap) issue_a(cn, 8 * (3 * (dy + 1) + dxi) + wave, nbuf);
#endif
#ifndef ZD_NO_WDMA
                issue_w(wc < nch ? wc : nch - 1, wc < nch ? wt : 8, (y + 3) & 3);
#endif
                wt += 1;
                if (wt == 9) { wt = 0; wc += 1; }
                ZD_FENCE();
                // Retired in order (LDS-DMA only, kernel_common.h).  The NEXT step reads its fragments before its own wait, so this
                // wait covers what step y + 1 needs: half-tile y + 1 and, when y + 1 opens a chunk, every activation piece of that
                // chunk (the last one went out at tap 7).  What may stay in flight = what was requested after those:
                if (dy == 1 && dxi == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");        // after A(tap 7): W(y+2) | W(y+3)
                else if (dy == -1 && dxi == 0) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");   // after W(y+1): W(y+2) | A, W(y+3)
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                             // after W(y+1): A, W(y+2) | A, W(y+3)
                ZD_FENCE();
                __builtin_amdgcn_s_barrier();
                ZD_FENCE();
                __builtin_amdgcn_s_setprio(1);
                static_for<LO, HI>([&](auto t_) __attribute__((always_inline)) {
                    constexpr int t = decltype(t_)::value;
                    // weights as operand A: a lane gets 4 consecutive channels of (board c15, square of tile t).  The five MFMAs of a
                    // tile and the wave-uniform branch that skips them on the padding column are ONE asm block: as C++ control
                    // flow around the accumulators hipcc spilled 150 registers.  (The accumulators of the five are different
                    // registers, so no MFMA of a block waits for another; the compiler's own waits for the fragments precede it.)
                    zd_mma5<t>(acc[t], fb, fa[t], tsk);
                });
                __builtin_amdgcn_s_setprio(0);
                ZD_FENCE();
                __builtin_amdgcn_s_barrier();
                ZD_FENCE();
                y += 1;
            }
        });
    }
#ifdef SW_STAMP
    const unsigned long long st_c1 = __builtin_amdgcn_s_memtime(), st_r1 = __builtin_amdgcn_s_memrealtime();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the re-requested pieces have landed
    if (wp == 0) __builtin_amdgcn_s_barrier();          // matches group 1's extra barrier
    __builtin_amdgcn_s_barrier();                       // every wave's DMA has landed before anyone stages output

#ifdef PP_NO_EPILOGUE
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NG; ++ni) asm volatile("" :: "v"(acc[mi][ni]));
#else
    // ---- epilogue: GroupNorm(16 channels x 64 squares of a board) + activation, fp16 out.
    // lane (c15 = board, q) holds, in register r of tile (t, ni): square of tile t, channel 80 qr + 16 ni + 4 q + r.
    {
        float s[NG], ss[NG];
        static_for<0, NG>([&](auto ni_) __attribute__((always_inline)) {
            constexpr int ni = decltype(ni_)::value;
            float s_ = 0.f, ss_ = 0.f;
            static_for<0, MT>([&](auto t_) __attribute__((always_inline)) {
                const zd_float4v av = acc[decltype(t_)::value][ni];
                static_for<0, 4>([&](auto r_) __attribute__((always_inline)) { const float v = av[decltype(r_)::value]; s_ += v; ss_ = __builtin_fmaf(v, v, ss_); });
            });
            s_ += __shfl_xor(s_, 16); ss_ += __shfl_xor(ss_, 16);         // the group's four channel quads
            s_ += __shfl_xor(s_, 32); ss_ += __shfl_xor(ss_, 32);
            s[ni] = s_; ss[ni] = ss_;
        });
        float2* S = reinterpret_cast<float2*>(smem + ZD_OFF_S);
        if (q == 0) {
            static_for<0, NG>([&](auto ni_) __attribute__((always_inline)) {
                constexpr int ni = decltype(ni_)::value;
                S[(wave * 16 + c15) * NG + ni] = make_float2(s[ni], ss[ni]);
            });
        }
        __syncthreads();
        static_for<0, NG>([&](auto ni_) __attribute__((always_inline)) {
            constexpr int ni = decltype(ni_)::value;
            float s_ = 0.f, ss_ = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) {                               // fixed order: the same sums in every wave
                const float2 p = S[(w8 * 16 + c15) * NG + ni];
                s_ += p.x; ss_ += p.y;
            }
            s[ni] = s_; ss[ni] = ss_;
        });
        __syncthreads();                                                   // S lies inside wave 7's image
        char* img = smem + wave * 20480;                                   // [8 tiles][16 boards][80 channels] fp16
        const int colbase = qr * 80 + 4 * q;
        float g[NG][4], sh[NG][4];
        static_for<0, NG>([&](auto ni_) __attribute__((always_inline)) {
            constexpr int ni = decltype(ni_)::value;
            const float mean = s[ni] * (1.f / 1024.f);
            float var = ss[ni] * (1.f / 1024.f) - mean * mean;
            var = var > 0.f ? var : 0.f;
            const float rstd = rsqrtf(var + 1e-5f);
            const float4 gm = *reinterpret_cast<const float4*>(a.gn_gamma + colbase + ni * 16);
            const float4 bt = *reinterpret_cast<const float4*>(a.gn_beta + colbase + ni * 16);
            g[ni][0] = rstd * gm.x; g[ni][1] = rstd * gm.y; g[ni][2] = rstd * gm.z; g[ni][3] = rstd * gm.w;
            sh[ni][0] = bt.x - mean * g[ni][0]; sh[ni][1] = bt.y - mean * g[ni][1];
            sh[ni][2] = bt.z - mean * g[ni][2]; sh[ni][3] = bt.w - mean * g[ni][3];
        });
        // tile t of the image = 16 boards x 160 B = 160 16-byte units: lanes take units lane, 64 + lane and (lanes < 32) 128 + lane;
        // unit u = (board u / 10, chunk u % 10).  Tile by tile: the stores of tile t are in flight under the arithmetic of t + 1.
        char* wbase = img + c15 * 160 + q * 8;
        char* out = reinterpret_cast<char*>(a.out) + ((size_t)b0 * 64 * a.ldo + qr * 80) * 2;
        const uint32_t ldo2 = (uint32_t)a.ldo * 2u;
        int fb_[3], fc_[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { const int u = 64 * k + lane; fb_[k] = u / 10; fc_[k] = u - fb_[k] * 10; }
        const int bvalid = boards - b0;
        static_for<0, MT>([&](auto t_) __attribute__((always_inline)) {
            constexpr int t = decltype(t_)::value;
            static_for<0, NG>([&](auto ni_) __attribute__((always_inline)) {
                constexpr int ni = decltype(ni_)::value;
                zd_half4 h;
                static_for<0, 4>([&](auto r_) __attribute__((always_inline)) {
                    constexpr int r = decltype(r_)::value;
                    h[r] = (_Float16)act_fast<ACT>(acc[t][ni][r] * g[ni][r] + sh[ni][r]);
                });
                *reinterpret_cast<zd_half4*>(wbase + t * 2560 + ni * 32) = h;
            });
            const uint32_t sq = (uint32_t)(t * 8 + ((t + wave) & 7));
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (k < 2 || lane < 32) {
                    const uint4 v = *reinterpret_cast<const uint4*>(img + (t * 160 + 64 * k + lane) * 16);
                    if (fb_[k] < bvalid)
                        *reinterpret_cast<uint4*>(out + (((uint32_t)fb_[k] * 64u + sq) * ldo2 + (uint32_t)fc_[k] * 16u)) = v;
                }
            }
        });
    }
#endif
#ifdef SW_STAMP
    if (tid == 0) {
        unsigned long long* o = g_zd_stamp + (size_t)blockIdx.x * 8;
        unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        o[0] = st_entry; o[1] = st_r0; o[2] = st_r1; o[3] = __builtin_amdgcn_s_memrealtime(); o[4] = st_c1 - st_c0; o[5] = hw; o[6] = xcc;
    }
#endif
}

// true: launch_conv_zd takes these arguments (conv1 of a residual block: GroupNorm + activation epilogue, nothing else fused)
bool conv_zd_supports(const GemmArgs& a) {
    if (a.gn_gamma == nullptr || a.gn_beta == nullptr || a.res != nullptr || a.mul != nullptr || a.bias != nullptr) return false;
    if (a.out_stats != nullptr || a.out_f32 != 0 || a.out_scale != 1.f || a.posenc != nullptr || a.out2 != nullptr) return false;
    if (!a.w_pp || a.N != 320 || a.Npad != 320 || a.Cin % 64 != 0) return false;
    if (a.Mvalid <= 0 || a.Mvalid % 64 != 0 || a.Mvalid > a.Mrows) return false;
    if (a.epi_act != ACT_SILU && a.epi_act != ACT_RELU) return false;
    if ((size_t)a.Mrows * (size_t)(a.ldo > a.Cin ? a.ldo : a.Cin) * 2 >= ((size_t)1 << 32)) return false;     // 32-bit lane offsets
    return true;
}

template <int ACT>
static hipError_t launch_conv_zd_e(const GemmArgs& a, hipStream_t st) {
    static DeviceOnce once;
    hipError_t e = once.run([] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_zd_kernel<ACT>), hipFuncAttributeMaxDynamicSharedMemorySize, ZD_LDS);
    });
    if (e != hipSuccess) return e;
    const int tiles = (a.Mvalid / 64 + 15) / 16;
    const dim3 grid((unsigned)(((tiles + 7) / 8) * 32));
    hipLaunchKernelGGL((conv_zd_kernel<ACT>), grid, dim3(512), ZD_LDS, st, a);
    return hipGetLastError();
}

hipError_t launch_conv_zd(const GemmArgs& a, hipStream_t st) {
    if (!conv_zd_supports(a)) return hipErrorInvalidValue;
    return a.epi_act == ACT_SILU ? launch_conv_zd_e<ACT_SILU>(a, st) : launch_conv_zd_e<ACT_RELU>(a, st);
}
