// conv_big_kernel: the dominant kernel of the network forward (3x3 / 1x1 implicit-GEMM conv on MFMA with
// global_load_lds staging and fused block epilogues).  See net_kernels.hip's header for the design.
#include "kernel_common.h"
#include "conv_epilogue.h"
#include "conv_tail.h"

// ---------------------------------------------------------------------------
// conv_big: the hot kernel (see file header)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    // LDS destination = wave-uniform base + lane*16 (hardware); the global source is per lane
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// EPI: 0 plain (bias/act/mul/scale + column sums), 1 fused GroupNorm16+act.  A template parameter so that each
// epilogue gets its own register allocation.  (Fusing the SE gate + residual, or residual + LayerNorm, into this
// epilogue was built and measured: correct but ~3x slower than conv + ew_board -- the per-board SE MLP is a
// latency-bound GEMV that a 1-workgroup-per-CU kernel cannot hide -- so those passes stay in ew_board_kernel.)
// WNW: waves along N.  2: 8 waves (2 per SIMD), wave tile 64x160, <=256 VGPRs.  1: 4 waves (one per SIMD, the whole
// 512-entry register file each), wave tile 64x320: 12 LDS fragment reads per 20 MFMAs instead of 7 per 10.
template <int TAPS, int EPI, int WNW, int ACT = ACT_NONE>
__global__ __launch_bounds__(256 * WNW) void conv_big_kernel(GemmArgs a) {
    // split K (1x1 only): workgroup row blockIdx.y takes 1/ksplit of the 64-channel chunks and writes an fp32 partial tile
    const int ksp = (TAPS == 1 && a.ksplit > 1) ? a.ksplit : 1;
    const int kz = (TAPS == 1 && a.ksplit > 1) ? blockIdx.y : 0;
    if (ksp > 1) a.out = reinterpret_cast<float*>(a.out) + (size_t)kz * a.Mrows * a.ldo;
    constexpr int NT = 10 / WNW;
    constexpr int NWAVES = 4 * WNW;
    constexpr int WP = 40 / NWAVES;      // weight DMA pieces per wave and stage
    constexpr int AP = 32 / NWAVES;      // activation DMA pieces per wave and chunk
    constexpr int A_BYTES = 256 * 128;    // 4 boards x 64 squares x 64 channels fp16
    constexpr int W_BYTES = 320 * 128;    // 320 output channels x 64 k fp16
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* A_lds = smem;                   // [2][A_BYTES]
    char* W_lds = smem + 2 * A_BYTES;     // [2][W_BYTES]
    char* Z_lds = W_lds + 2 * W_BYTES;    // one all-zero square (128 B)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3;              // board within the tile
    const int wn = wave >> 2;             // N part (32*NT channels = 2*NT whole GroupNorm groups)
    // 1-D grid, XCD-aware: workgroup ids go round-robin over the 8 XCDs (each with its own L2), so the N blocks of one
    // row block are given ids that are consecutive ON ONE XCD -- they run back to back there and the row block's
    // activations are fetched from HBM once instead of once per N block (qkv: N = 960 = 3 blocks).
    const int nblk = a.Npad / 320, rblk = a.Mrows >> 8;
    int rb, nb;
    if ((rblk & 7) == 0) {
        const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
        rb = (q / nblk) * 8 + xcd;
        nb = q % nblk;
    } else {
        rb = blockIdx.x % rblk;
        nb = blockIdx.x / rblk;
    }
    const int m0 = rb * 256;
    const int n0 = nb * 320;
    const int Cin = a.Cin;
    const int nchunk = Cin >> 6;                 // all chunks (the weight layout's stride)
    const int cchunks = nchunk / ksp;            // this workgroup's chunks, starting at c0
    const int c0 = kz * cchunks;
    const int nsteps = cchunks * TAPS;
    const int half = lane >> 5;

    if (tid < 8) reinterpret_cast<uint4*>(Z_lds)[tid] = make_uint4(0, 0, 0, 0);
    static_assert(WP >= 1 && AP >= 1 && NT >= WP, "piece distribution");

    const char* in_bytes = reinterpret_cast<const char*>(a.in);
    const char* w_bytes = reinterpret_cast<const char*>(a.w);

    auto issue_A = [&](int chunk, int buf) {
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int q = wave * AP + i;                // 1-KiB piece: squares 8q..8q+7 of the 256-row tile
            const int p = 8 * q + (lane >> 3);
            const int cl = lane & 7;                    // LDS 16-byte chunk this lane fills
            const char* src = in_bytes + ((size_t)(m0 + p) * Cin + (size_t)(c0 + chunk) * 64) * 2 + 16 * (cl ^ ((p >> 1) & 7));
            glds16(src, A_lds + buf * A_BYTES + q * 1024);
        }
    };
    auto issue_W = [&](int step, int buf) {
        const int chunk = step / TAPS, tap = step - chunk * TAPS;
        const char* src = w_bytes + ((size_t)(tap * nchunk + c0 + chunk) * a.Npad + n0) * 128;
#pragma unroll
        for (int i = 0; i < WP; ++i) {
            const int q = wave * WP + i;
            glds16(src + q * 1024 + lane * 16, W_lds + buf * W_BYTES + q * 1024);
        }
    };
    // single 1-KiB pieces, so the DMA issue (~100 cycles of the wave's issue slot each) can be spread between
    // MFMA groups instead of stalling all eight waves right after the barrier
    auto issue_W_piece = [&](const char* wsrc, int buf, int i) {
        const int q = wave * WP + i;
        glds16(wsrc + q * 1024 + lane * 16, W_lds + buf * W_BYTES + q * 1024);
    };
    auto issue_A_piece = [&](int chunk, int buf, int i) {
        const int q = wave * AP + i;
        const int p = 8 * q + (lane >> 3);
        const int cl = lane & 7;
        const char* src = in_bytes + ((size_t)(m0 + p) * Cin + (size_t)(c0 + chunk) * 64) * 2 + 16 * (cl ^ ((p >> 1) & 7));
        glds16(src, A_lds + buf * A_BYTES + q * 1024);
    };

    float16v acc[2][NT];
    static_for<0, 2>([&](auto mi) __attribute__((always_inline)) {
        static_for<0, NT>([&](auto ni) __attribute__((always_inline)) {
            acc[decltype(mi)::value][decltype(ni)::value] = float16v{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f,
                                                                      0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        });
    });

    // per-lane constants of the fragment reads
    const int r31 = lane & 31;
    const int wfx = ((r31 >> 1) & 7) ^ half;                           // weight rows: swizzle key ^ k-half
    const int wrow_off = (wn * NT * 32 + r31) * 128;
    int prow[2], py[2], px[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        prow[mi] = wm * 64 + mi * 32 + r31;
        py[mi] = (prow[mi] >> 3) & 7;
        px[mi] = prow[mi] & 7;
    }

    // ---- main loop ----
    // Stage s = (chunk, tap) weights in W buffer s&1, activations of chunk c in A buffer c&1.  Software pipeline:
    //   * fragment reads run one k-step ahead of the MFMAs, ACROSS the step boundary (two register sets);
    //   * one barrier per step, placed after the step's last LDS reads (before its 4th k-step): by then every wave
    //     has waited for its DMA pieces of stage s+1 (issued one step earlier) and has finished reading stage s, so
    //     after the barrier stage s+1 is visible and stage s's buffers are free;
    //   * the DMA pieces of stage s+2 are issued right after that barrier, between the MFMA pairs of the 4th k-step.
    half8 fa0[2], fa1[2], fb[2][NT];
    const char* abase[2];
    int afx[2];
    const char* Wb;
    auto set_addr = [&](int st_) __attribute__((always_inline)) {
        const int ch = st_ / TAPS, tp = st_ - ch * TAPS;
        const int dy = (TAPS == 9) ? (tp / 3 - 1) : 0;
        const int dx = (TAPS == 9) ? (tp - (tp / 3) * 3 - 1) : 0;
        const char* Ab = A_lds + (ch & 1) * A_BYTES;
        Wb = W_lds + (st_ & 1) * W_BYTES + wrow_off;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int yy = py[mi] + dy, xx = px[mi] + dx;
            const bool ok = (TAPS == 1) || ((unsigned)yy < 8u && (unsigned)xx < 8u);
            const int pp = prow[mi] + dy * 8 + dx;
            abase[mi] = ok ? Ab + pp * 128 : Z_lds;
            afx[mi] = ok ? (((pp >> 1) & 7) ^ half) : 0;
        }
    };
    auto load_frags = [&](auto kk_, auto buf_) __attribute__((always_inline)) {
        constexpr int kk = decltype(kk_)::value;
        constexpr int bf = decltype(buf_)::value;
        fa0[bf] = *reinterpret_cast<const half8*>(abase[0] + 16 * (afx[0] ^ (kk << 1)));
        fa1[bf] = *reinterpret_cast<const half8*>(abase[1] + 16 * (afx[1] ^ (kk << 1)));
        const int woff = 16 * (wfx ^ (kk << 1));
        static_for<0, NT>([&](auto ni_) __attribute__((always_inline)) {
            constexpr int ni = decltype(ni_)::value;
            fb[bf][ni] = *reinterpret_cast<const half8*>(Wb + ni * 4096 + woff);
        });
    };
    auto mfma_set = [&](auto buf_) __attribute__((always_inline)) {
        constexpr int cur = decltype(buf_)::value;
        static_for<0, NT>([&](auto ni_) __attribute__((always_inline)) {
            constexpr int ni = decltype(ni_)::value;
            acc[0][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa0[cur], fb[cur][ni], acc[0][ni], 0, 0, 0);
            acc[1][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa1[cur], fb[cur][ni], acc[1][ni], 0, 0, 0);
        });
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;

    issue_A(0, 0);
    issue_W(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (nsteps > 1) {                       // stage 1 (its buffers have never been read)
        issue_W(1, 1);
        if (TAPS == 1) issue_A(1, 1);
    }
    set_addr(0);
    load_frags(I0{}, I0{});
    for (int s = 0; s < nsteps; ++s) {
        load_frags(I1{}, I1{}); mfma_set(I0{});
        load_frags(I2{}, I0{}); mfma_set(I1{});
        load_frags(I3{}, I1{}); mfma_set(I0{});
        // step boundary: stage s+1 landed + stage s no longer read
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int s2 = s + 2;
        const bool more2 = s2 < nsteps;
        const int ch2 = s2 / TAPS, tp2 = s2 - ch2 * TAPS;
        const bool newA = more2 && tp2 == 0;
        const char* wsrc = w_bytes + ((size_t)(tp2 * nchunk + c0 + ch2) * a.Npad + n0) * 128;
        if (s + 1 < nsteps) set_addr(s + 1);
        // 4th k-step: MFMAs of set 1, next step's first fragments into set 0, stage s+2 DMA pieces in between
        if (s + 1 < nsteps) load_frags(I0{}, I0{});
        static_for<0, NT>([&](auto ni_) __attribute__((always_inline)) {
            constexpr int ni = decltype(ni_)::value;
            acc[0][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa0[1], fb[1][ni], acc[0][ni], 0, 0, 0);
            acc[1][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa1[1], fb[1][ni], acc[1][ni], 0, 0, 0);
            if constexpr (ni < WP) { if (more2) issue_W_piece(wsrc, s2 & 1, ni); }
            if constexpr (ni < AP) { if (newA) issue_A_piece(ch2, ch2 & 1, ni); }
        });
    }

    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();                      // the epilogue stages the output tile over the A/W buffers
    // EPI 5: out = res + act(GroupNorm16(conv)) (+ the next GroupNorm's second output), conv_tail.h's PRE form
    if constexpr (EPI == 5) conv_tail_epilogue<ACT, true>(acc, a, smem, m0, wm, wn, wave, lane);
    else conv_tile_epilogue<EPI, ACT_NONE, NT>(acc, a, smem + wave * (NT * 64 * 64), m0, n0, wm, wn, lane);
}

template <int TAPS, int EPI, int WNW, int ACT = ACT_NONE>
static hipError_t launch_conv_big_e(const GemmArgs& a, hipStream_t st) {
    const size_t lds = 160 * 1024;
    static DeviceOnce once;
    hipError_t e = once.run([] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_big_kernel<TAPS, EPI, WNW, ACT>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    if (e != hipSuccess) return e;
    dim3 grid((a.Mrows / 256) * (a.Npad / 320), a.ksplit > 1 ? a.ksplit : 1);
    hipLaunchKernelGGL((conv_big_kernel<TAPS, EPI, WNW, ACT>), grid, dim3(256 * WNW), lds, st, a);
    return hipGetLastError();
}

hipError_t launch_conv_big(const GemmArgs& a, int taps, hipStream_t st) {
    // 1x1 convs whose N is a multiple of 320 (qkv, proj).  3x3: conv_zs_kernel / conv_pp16_kernel.
    // WNW = 2 (8 waves).  The 4-wave / 512-register form (WNW = 1) was built and measured: numerically identical,
    // 2x slower with hipcc's schedule (LDS latency exposed with one wave per SIMD, spills) -- not instantiated.
    if (taps != 1) return hipErrorInvalidValue;
    if (a.res != nullptr) {   // x + act(GroupNorm16(conv1x1(x))) in the epilogue (piece-square-table conv of the chess features)
        if (a.pre_gamma == nullptr || a.se_w1 != nullptr || a.N != 320 || a.Npad != 320 || a.ldo != 320 || a.bias != nullptr ||
            a.out_stats != nullptr || a.mul != nullptr || a.out_f32 != 0 || a.ksplit > 1 || (a.y2 != nullptr && a.gn_gamma == nullptr))
            return hipErrorInvalidValue;
        if ((size_t)a.Mrows * a.ldo * 2 >= ((size_t)1 << 32)) return hipErrorInvalidValue;
        if (a.epi_act == ACT_SILU) return launch_conv_big_e<1, 5, 2, ACT_SILU>(a, st);
        if (a.epi_act == ACT_RELU) return launch_conv_big_e<1, 5, 2, ACT_RELU>(a, st);
        return hipErrorInvalidValue;
    }
    if (a.gn_gamma != nullptr) return hipErrorInvalidValue;
    if (a.ksplit > 1 && ((a.Cin >> 6) % a.ksplit != 0 || !a.out_f32 || a.bias || a.mul || a.epi_act != ACT_NONE || a.out_stats))
        return hipErrorInvalidValue;
    if ((size_t)a.Mrows * a.ldo * 2 >= ((size_t)1 << 32)) return hipErrorInvalidValue;   // 32-bit store offsets
    const bool general = a.mul != nullptr || a.out_f32 != 0 || a.epi_act != ACT_NONE;   // per-element epilogue
    return general ? launch_conv_big_e<1, 2, 2>(a, st) : launch_conv_big_e<1, 0, 2>(a, st);
}

// Second pass of a split-K GEMM: fixed-order sum of the partial tiles, bias, activation, fp16.
__global__ void splitk_reduce_kernel(const float* __restrict__ part, int splits, size_t MN, int N, const float* __restrict__ bias,
                                     int act, _Float16* __restrict__ out) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= MN) return;
    float4 s = *reinterpret_cast<const float4*>(part + i);
    for (int z = 1; z < splits; ++z) {
        const float4 v = *reinterpret_cast<const float4*>(part + (size_t)z * MN + i);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const int n = (int)(i % (size_t)N);
    if (bias != nullptr) { s.x += bias[n]; s.y += bias[n + 1]; s.z += bias[n + 2]; s.w += bias[n + 3]; }
    typedef _Float16 half4r __attribute__((ext_vector_type(4)));
    *reinterpret_cast<half4r*>(out + i) = half4r{(_Float16)act_apply(s.x, act), (_Float16)act_apply(s.y, act),
                                                 (_Float16)act_apply(s.z, act), (_Float16)act_apply(s.w, act)};
}

hipError_t launch_splitk_reduce(const float* part, int splits, int M, int N, const float* bias, int act, _Float16* out,
                                hipStream_t st) {
    if (splits < 1 || N % 4 != 0) return hipErrorInvalidValue;
    const size_t MN = (size_t)M * N;
    const unsigned blocks = (unsigned)((MN / 4 + 255) / 256);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, part, splits, MN, N, bias, act, out);
    return hipGetLastError();
}
