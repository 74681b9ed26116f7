// MI355X (gfx950 / CDNA4) kernels for the R24-320 policy/value network forward.
//
// Replaces the reference's torch forward (azchess/model/resnet.py:656-760) on the
// self-play hot path.  Data layout in HBM: every activation is "NHWC"
// [board][64 squares][C] fp16, square n = row*8+col of the reference tensor
// (row 0 = rank 8).  A board's 64 squares are the 64 rows of one wave's MFMA
// tile, so every per-board reduction (GroupNorm statistics, SE pooling) is
// wave-local.
//
// Kernels:
//   conv_big_kernel<TAPS>  the hot kernel: implicit-GEMM 3x3 / 1x1 conv (and wide FC layers) on
//       v_mfma_f32_32x32x16_f16, fp32 accumulate.  WG = 8 waves, tile = 256 rows (4 boards) x 320
//       output channels, K stepped as (64-channel chunk, tap).  Both operands go global -> LDS by
//       global_load_lds (16 B/lane, no VGPR staging): the 64-channel slice of the 4 boards once per
//       chunk (double-buffered, reused by all 9 taps through shifted reads; out-of-board taps read a
//       shared zero pixel) and one 320x64 weight stage per step (double-buffered, pre-swizzled on the
//       host).  128-byte LDS rows, 16-byte chunk index XOR (row>>1)&7 -> conflict-free ds_read_b128.
//       Epilogue: bias/act/mul/scale + per-(board,channel) sums, or a fused GroupNorm16+activation
//       over the wave's own board (a wave holds all 64 rows of one board x 10 whole groups).
//   conv_gemm_kernel<TAPS,1,1,32>  generic small-tile variant (any N%32==0, Cin%32==0): stem, heads, FCs.
//   ew_board_kernel   per-board elementwise glue: GN+act, SE gate, residual add,
//       positional encoding, LayerNorm over C, output statistics.
//   attn_core_kernel  ChessAttention scores/softmax/PV for one (board, head).
//   planes_to_nhwc_kernel  f32 [B,19,8,8] -> fp16 [B,64,32].
#include "kernel_common.h"
#include "conv_epilogue.h"

// ---------------------------------------------------------------------------
// conv_gemm
// ---------------------------------------------------------------------------
// EPI 0: bias / runtime activation / gate multiply / scale, fp16 or f32 stored per element, per-(board, channel) sums.
// EPI 1 (round 4): act<ACT>(GroupNorm16(conv)) [+ positional encoding] applied to the accumulators (a wave holds all 64 squares of
//        its board for its 32 NT channels = 2 NT whole groups), fp16 through the wave's LDS image in 16-byte stores -- the stem
//        and the head convs no longer write a raw tensor + statistics for an ew_board pass to read back.
template <int TAPS, int WN, int NT, int KC, int EPI = 0, int ACT = 0>
__global__ __launch_bounds__(256 * WN) void conv_gemm_kernel(GemmArgs a) {
    constexpr int NB = WN * NT * 32;      // output channels per workgroup
    constexpr int NTHR = 256 * WN;
    constexpr int AST = KC + 8;           // LDS row stride in halfs (pad: conflict-free b128 reads)
    constexpr int APIX = (TAPS == 9) ? 100 : 64;
    constexpr int A_ELEMS = 4 * APIX * AST;
    constexpr int W_ELEMS = NB * AST;
    constexpr int A_PIECES = 4 * 64 * KC / 8;             // 16-byte pieces per A chunk
    constexpr int W_PIECES = NB * KC / 8;
    constexpr int A_PER = (A_PIECES + NTHR - 1) / NTHR;
    constexpr int W_PER = (W_PIECES + NTHR - 1) / NTHR;
    constexpr int K8 = KC / 8;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* A_lds = reinterpret_cast<_Float16*>(smem);
    _Float16* W_lds = A_lds + A_ELEMS;                    // 2 buffers

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave % 4;              // board within the tile
    const int wn = wave / 4;              // N half
    const int m0 = blockIdx.x * 256;      // first row
    const int n0 = blockIdx.y * NB;
    const int Cin = a.Cin;
    const int nchunk = Cin / KC;
    const int Npad = a.Npad;

    // zero the halo image once (borders stay zero for the whole kernel)
    if (TAPS == 9) {
        for (int i = tid; i < A_ELEMS / 8; i += NTHR)
            reinterpret_cast<uint4*>(A_lds)[i] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();

    float16v acc[2][NT];
    static_for<0, 2>([&](auto mi) __attribute__((always_inline)) {
        static_for<0, NT>([&](auto ni) __attribute__((always_inline)) {
            acc[decltype(mi)::value][decltype(ni)::value] = float16v{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f,
                                                                      0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        });
    });

    // per-lane LDS base of its two A rows (squares lane&31 and 32+(lane&31) of board wm)
    int apix[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        int px = mi * 32 + (lane & 31);
        if (TAPS == 9) apix[mi] = wm * 100 + ((px >> 3) + 1) * 10 + (px & 7) + 1;
        else apix[mi] = wm * 64 + px;
    }
    const int khalf = 8 * (lane >> 5);

    uint4 areg[A_PER];
    uint4 wreg[W_PER];

    auto load_A = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            int p = tid + i * NTHR;
            if (A_PIECES % NTHR == 0 || p < A_PIECES) {
                int row = p / K8, c8 = p % K8;
                areg[i] = *reinterpret_cast<const uint4*>(a.in + (size_t)(m0 + row) * Cin + chunk * KC + c8 * 8);
            }
        }
    };
    auto store_A = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            int p = tid + i * NTHR;
            if (A_PIECES % NTHR == 0 || p < A_PIECES) {
                int row = p / K8, c8 = p % K8;
                int b = row >> 6, px = row & 63;
                uint4 v = areg[i];
                int pix = (TAPS == 9) ? (b * 100 + ((px >> 3) + 1) * 10 + (px & 7) + 1) : row;
                *reinterpret_cast<uint4*>(A_lds + pix * AST + c8 * 8) = v;
            }
        }
    };
    auto load_W = [&](int step) {
        // step = chunk*TAPS + tap ; packed layout [tap][chunk][Npad][KC]
        int chunk = step / TAPS, tap = step % TAPS;
        const _Float16* src = a.w + ((size_t)(tap * nchunk + chunk) * Npad + n0) * KC;
#pragma unroll
        for (int i = 0; i < W_PER; ++i) {
            int p = tid + i * NTHR;
            if (W_PIECES % NTHR == 0 || p < W_PIECES)
                wreg[i] = *reinterpret_cast<const uint4*>(src + (size_t)p * 8);
        }
    };
    auto store_W = [&](int buf) {
#pragma unroll
        for (int i = 0; i < W_PER; ++i) {
            int p = tid + i * NTHR;
            if (W_PIECES % NTHR == 0 || p < W_PIECES) {
                int n = p / K8, k8 = p % K8;
                *reinterpret_cast<uint4*>(W_lds + buf * W_ELEMS + n * AST + k8 * 8) = wreg[i];
            }
        }
    };

    const int nsteps = nchunk * TAPS;
    load_A(0);
    load_W(0);
    for (int s = 0; s < nsteps; ++s) {
        const int chunk = s / TAPS, tap = s % TAPS;
        if (tap == 0) {
            if (s > 0) __syncthreads();   // previous chunk's reads of A_lds are done
            store_A(chunk);
        }
        store_W(s & 1);
        __syncthreads();
        if (s + 1 < nsteps) {
            load_W(s + 1);
            if ((s + 1) % TAPS == 0) load_A((s + 1) / TAPS);
        }
        const int tapoff = (TAPS == 9) ? ((tap / 3 - 1) * 10 + (tap % 3 - 1)) : 0;
        const _Float16* Wb = W_lds + (s & 1) * W_ELEMS + (wn * NT * 32 + (lane & 31)) * AST + khalf;
        static_for<0, KC / 16>([&](auto kk_) __attribute__((always_inline)) {
            constexpr int kk = decltype(kk_)::value;
            half8 af0 = *reinterpret_cast<const half8*>(A_lds + (apix[0] + tapoff) * AST + kk * 16 + khalf);
            half8 af1 = *reinterpret_cast<const half8*>(A_lds + (apix[1] + tapoff) * AST + kk * 16 + khalf);
            static_for<0, NT>([&](auto ni_) __attribute__((always_inline)) {
                constexpr int ni = decltype(ni_)::value;
                half8 bf = *reinterpret_cast<const half8*>(Wb + ni * 32 * AST + kk * 16);
                acc[0][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af0, bf, acc[0][ni], 0, 0, 0);
                acc[1][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af1, bf, acc[1][ni], 0, 0, 0);
            });
        });
    }

    // ---------------- epilogue ----------------
    if constexpr (EPI == 1) {
        __syncthreads();                                  // every wave has left the operand tiles: the LDS becomes the staging images
        char* lds_wave = smem + wave * (64 * 64 * NT);
        const int cb = n0 + wn * NT * 32;                 // first column of this wave's tile
        const int r31 = lane & 31, half = lane >> 5;
        GemmArgs o = a;                                   // destination of this wave's columns (conv_stage_flush reads out, ldo, Mvalid)
        int lc = cb;
        if (a.out2 != nullptr && cb >= a.nsplit) { o.out = a.out2; o.ldo = a.ldo2; lc = cb - a.nsplit; }
        o.out = reinterpret_cast<_Float16*>(o.out) + lc;
        char* wbase = conv_stage_base<NT>(lds_wave, lane);
        static_for<0, NT>([&](auto ni_) __attribute__((always_inline)) {
            constexpr int ni = decltype(ni_)::value;
            const int col = cb + ni * 32 + r31;
            float s = 0.f, ss = 0.f;
            static_for<0, 2>([&](auto mi_) __attribute__((always_inline)) {
                const float16v av = acc[decltype(mi_)::value][ni];
                static_for<0, 16>([&](auto r_) __attribute__((always_inline)) { const float v = av[decltype(r_)::value]; s += v; ss += v * v; });
            });
#pragma unroll
            for (int o2 = 1; o2 <= 8; o2 <<= 1) { s += __shfl_xor(s, o2); ss += __shfl_xor(ss, o2); }
            s += __shfl_xor(s, 32); ss += __shfl_xor(ss, 32);
            const float mean = s * (1.f / 1024.f);
            float var = ss * (1.f / 1024.f) - mean * mean;
            var = var > 0.f ? var : 0.f;
            const float g = rsqrtf(var + 1e-5f) * a.gn_gamma[col];
            const float sh = a.gn_beta[col] - mean * g;
            static_for<0, 2>([&](auto mi_) __attribute__((always_inline)) {
                constexpr int mi = decltype(mi_)::value;
                float v[16];
                static_for<0, 16>([&](auto r_) __attribute__((always_inline)) {
                    constexpr int r = decltype(r_)::value;
                    v[r] = act_fast<ACT>(acc[mi][ni][r] * g + sh);
                });
                if (a.posenc != nullptr) {
                    static_for<0, 16>([&](auto r_) __attribute__((always_inline)) {
                        constexpr int r = decltype(r_)::value;
                        const int sq = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        v[r] += a.posenc[(size_t)sq * a.N + col];
                    });
                }
                conv_stage_tile<NT, mi, ni>(v, wbase, lane);
            });
        });
        conv_stage_flush<NT>(o, lds_wave, m0, 0, wm, 0, lane);
        return;
    }
    // (everything indexed with compile-time constants: a runtime-indexed accumulator goes to scratch)
    const int ldo = a.ldo;
    const int rowbase = m0 + wm * 64 + 4 * (lane >> 5);
    const int colbase = n0 + wn * NT * 32 + (lane & 31);
    const int epi_act = a.epi_act;
    const float oscale = a.out_scale;
    const bool has_mul = a.mul != nullptr;
    const bool f32out = a.out_f32 != 0;
    const bool want_stats = a.out_stats != nullptr;
    static_for<0, NT>([&](auto ni_) __attribute__((always_inline)) {
        constexpr int ni = decltype(ni_)::value;
        const int col = colbase + ni * 32;
        const float bias = a.bias != nullptr ? a.bias[col] : 0.f;
        float s = 0.f, ss = 0.f;
        static_for<0, 2>([&](auto mi_) __attribute__((always_inline)) {
            constexpr int mi = decltype(mi_)::value;
            const float16v av = acc[mi][ni];
            static_for<0, 16>([&](auto r_) __attribute__((always_inline)) {
                constexpr int r = decltype(r_)::value;
                const int row = rowbase + mi * 32 + (r & 3) + 8 * (r >> 2);
                float v = av[r] + bias;
                if (epi_act != ACT_NONE) v = act_apply(v, epi_act);
                if (has_mul) v *= (float)a.mul[(size_t)row * ldo + col];
                v *= oscale;
                s += v; ss += v * v;
                if (row < a.Mvalid) {
                    if (f32out) reinterpret_cast<float*>(a.out)[(size_t)row * ldo + col] = v;
                    else reinterpret_cast<_Float16*>(a.out)[(size_t)row * ldo + col] = (_Float16)v;
                }
            });
        });
        if (want_stats) {
            s += __shfl_xor(s, 32);
            ss += __shfl_xor(ss, 32);
            if (lane < 32) {
                float* st = a.out_stats + ((size_t)(m0 / 64 + wm) * a.N + col) * 2;
                st[0] = s; st[1] = ss;
            }
        }
    });
}


template <int TAPS, int WN, int NT, int KC>
static size_t conv_gemm_lds(int Cin) {
    constexpr int NB = WN * NT * 32;
    constexpr int AST = KC + 8;
    constexpr int APIX = (TAPS == 9) ? 100 : 64;
    (void)Cin;
    const size_t main_loop = (size_t)(4 * APIX * AST + 2 * NB * AST) * 2 + 64;
    const size_t staging = (size_t)WN * 4 * 64 * 64 * NT;          // EPI 1: one [64 rows][32 NT] fp16 image per wave
    return main_loop > staging ? main_loop : staging;
}

template <int TAPS, int WN, int NT, int KC, int EPI = 0, int ACT = 0>
static hipError_t launch_conv_gemm_t(const GemmArgs& a, hipStream_t st) {
    constexpr int NB = WN * NT * 32;
    size_t lds = conv_gemm_lds<TAPS, WN, NT, KC>(a.Cin);
    static DeviceOnce once;
    hipError_t e = once.run([] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_kernel<TAPS, WN, NT, KC, EPI, ACT>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    if (e != hipSuccess) return e;
    dim3 grid(a.Mrows / 256, a.Npad / NB);
    hipLaunchKernelGGL((conv_gemm_kernel<TAPS, WN, NT, KC, EPI, ACT>), grid, dim3(256 * WN), lds, st, a);
    return hipGetLastError();
}
// small tile with the fused GroupNorm epilogue: the stem (3x3, 64 channels per workgroup), one head conv (64 or 128 channels:
// the whole N in one workgroup, the input rows read once), or the policy-head and value-head convs together (64 + 128 channels,
// three wave groups, two outputs)
template <int ACT>
static hipError_t launch_conv_gemm_gn(const GemmArgs& a, int taps, hipStream_t st) {
    if (a.bias != nullptr || a.mul != nullptr || a.out_stats != nullptr || a.out_f32 || a.out_scale != 1.f || a.Npad != a.N)
        return hipErrorInvalidValue;
    // (stem: 64 channels per workgroup; a 160-channel tile -- two N blocks instead of five -- measured 1351 us against 711: 160
    // accumulator registers at one wave per SIMD)
    if (taps == 9) return a.Npad % 64 == 0 && a.out2 == nullptr ? launch_conv_gemm_t<9, 1, 2, 32, 1, ACT>(a, st) : hipErrorInvalidValue;
    if (a.posenc != nullptr) return hipErrorInvalidValue;
    if (a.out2 != nullptr) return a.Npad == 192 && a.nsplit == 64 ? launch_conv_gemm_t<1, 3, 2, 32, 1, ACT>(a, st) : hipErrorInvalidValue;
    if (a.Npad == 160) return launch_conv_gemm_t<1, 1, 5, 32, 1, ACT>(a, st);      // SSL head convs of the 320-wide trunk
    if (a.Npad == 128) return launch_conv_gemm_t<1, 2, 2, 32, 1, ACT>(a, st);
    if (a.Npad == 64) return launch_conv_gemm_t<1, 1, 2, 32, 1, ACT>(a, st);
    if (a.Npad == 32) return launch_conv_gemm_t<1, 1, 1, 32, 1, ACT>(a, st);
    return hipErrorInvalidValue;
}

hipError_t launch_conv_big(const GemmArgs& a, int taps, hipStream_t st);   // conv_big.hip
hipError_t launch_conv_pp16(const GemmArgs& a, hipStream_t st);            // conv_pp16.hip
hipError_t launch_conv_zs(const GemmArgs& a, hipStream_t st);              // conv_zs.hip
bool conv_zs_supports(const GemmArgs& a);

int conv_gemm_tile_n(int Cin, int Npad) {
    return (Npad % 320 == 0 && Cin % 64 == 0) ? 320 : 32;
}
int conv_gemm_kc(int Cin, int Npad) {
    return (Npad % 320 == 0 && Cin % 64 == 0) ? 64 : 32;
}

hipError_t launch_conv_gemm(const GemmArgs& a, int taps, hipStream_t st) {
    if (a.Mrows % 256 != 0 || a.Cin % 32 != 0 || a.Npad % 32 != 0) return hipErrorInvalidValue;
    const bool big = conv_gemm_tile_n(a.Cin, a.Npad) == 320;
    if (a.gn_gamma != nullptr && !big) {                              // small tile with the fused GroupNorm epilogue
        if (a.epi_act == ACT_SILU) return launch_conv_gemm_gn<ACT_SILU>(a, taps, st);
        if (a.epi_act == ACT_RELU) return launch_conv_gemm_gn<ACT_RELU>(a, taps, st);
        return hipErrorInvalidValue;
    }
    if (taps == 9) {
        if (big) {
            if (!a.w_pp) return hipErrorInvalidValue;
            // conv_zs_kernel: the v_mfma_f32_16x16x32_f16 loop with the wave tile laid out so that the M-tiles that only see the
            // zero padding above / below the board are skipped (8.3 % of the MFMAs).  conv_pp16_kernel (the same loop without the
            // skipping) takes the shapes conv_zs does not (squeeze-excite wider than 96 hidden units) and A/B runs
            // (GemmArgs::no_zs, set from M0_CONV_ZS=0 when the network is created).
            if (!a.no_zs && conv_zs_supports(a)) return launch_conv_zs(a, st);
            return launch_conv_pp16(a, st);
        }
        return launch_conv_gemm_t<9, 1, 1, 32>(a, st);
    } else if (taps == 1) {
        if (big) return a.w_pp ? hipErrorInvalidValue : launch_conv_big(a, 1, st);
        // Head convs over the whole trunk (M = 64 x boards, N = 64 / 128): a workgroup per 32 output channels re-reads its 256
        // trunk rows once per N block (168 MB x 2..4 at 4096 boards); 64 channels per workgroup halve that.
        // Same arithmetic per output element (bit-identical).  The FCs (M = boards) keep the narrow tile: they need the workgroups.
        // (64 channels per workgroup; 128 -- the whole value-head conv in one workgroup -- measured slower: 200 registers)
        if (a.Mrows / 256 >= 256 && a.Npad % 64 == 0) return launch_conv_gemm_t<1, 1, 2, 32>(a, st);
        return launch_conv_gemm_t<1, 1, 1, 32>(a, st);
    }
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------
// ew_board: one workgroup per board, tensor [64][C] fp16.
//   v = t
//   if gn_gamma: v = act(GroupNorm16(v))             (statistics from t_stats)
//   elif gate:   v *= gate[b][c]                     (squeeze-excite, se_gate_kernel)
//   if res:      v += res
//   if posenc:   v += posenc[n][c]
//   if ln_g:     v = LayerNorm_C(v)
//   y = v                                             (the raw residual stream)
//   out_stats = per-channel (sum, sumsq) of y over the 64 squares
//   if y2: y2 = act(GroupNorm16(y; gn2_gamma, gn2_beta))   -- the pre-activation input of the next block's conv1
//
// Memory-bound (4 x 40 KB per board at C = 320) and, per board, a chain of dependent steps (statistics -> gate MLP ->
// values -> statistics -> second output), so what decides the speed is how many bytes a CU keeps in flight:
// 2C threads, thread = (8-channel chunk, group of 4 squares); every thread issues ALL its tensor loads (4 squares x
// {t, res} x 16 B) before anything else, the gate / GroupNorm parameters are computed while they fly, the board's
// values stay packed in registers for the second output (no re-read), every lane is live, every reduction has a fixed
// order (bit-reproducible).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(768) void ew_board_kernel(EwArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int C = a.C, NC = C >> 3;
    float* sc = reinterpret_cast<float*>(smem);        // [C] scale (GN) or gate (SE); later GN2 scale
    float* sh = sc + C;                                // [C] shift / pooled mean
    float* tot = sh + C;                               // [C][2]
    float* rowbuf = tot + 2 * C;                       // [64][2] LayerNorm (mean, rstd) per square
    float* red = rowbuf + 128;                            // [16][C][2] stats partials; SE partials; LN partials [64][NC][2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nthr = blockDim.x, nw = nthr >> 6;       // 16 * NC threads
    const int chunk = tid % NC, sg = tid / NC;         // squares 4 sg .. 4 sg + 3
    const int c0 = chunk * 8;
    const int b = blockIdx.x;
    const _Float16* t = a.t + (size_t)b * 64 * C + (size_t)(sg * 4) * C + c0;
    const _Float16* res = a.res ? a.res + (size_t)b * 64 * C + (size_t)(sg * 4) * C + c0 : nullptr;
    const float* tst = a.t_stats ? a.t_stats + (size_t)b * C * 2 : nullptr;

    // 1. all tensor loads first
    const bool gn = a.gn_gamma != nullptr;
    const bool se = (!gn) && a.gate != nullptr;
    float gatev[8];
    {
        const float4 g0 = se ? *reinterpret_cast<const float4*>(a.gate + (size_t)b * C + c0) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 g1 = se ? *reinterpret_cast<const float4*>(a.gate + (size_t)b * C + c0 + 4) : make_float4(1.f, 1.f, 1.f, 1.f);
        gatev[0] = g0.x; gatev[1] = g0.y; gatev[2] = g0.z; gatev[3] = g0.w;
        gatev[4] = g1.x; gatev[5] = g1.y; gatev[6] = g1.z; gatev[7] = g1.w;
    }
    // (the second output's GroupNorm parameters too: a late load is one more exposed latency per board)
    float g2w = 0.f, g2b = 0.f;
    if (a.y2 != nullptr && tid < C) { g2w = a.gn2_gamma[tid]; g2b = a.gn2_beta[tid]; }
    half8 tv[4], rv[4];
    static_for<0, 4>([&](auto k_) __attribute__((always_inline)) {
        constexpr int k = decltype(k_)::value;
        tv[k] = *reinterpret_cast<const half8*>(t + k * C);
        rv[k] = res ? *reinterpret_cast<const half8*>(res + k * C) : half8{0, 0, 0, 0, 0, 0, 0, 0};
    });

    // 2. per-channel scale / shift
    if (gn) {
        for (int c = tid; c < C; c += nthr) {
            const int g0 = (c >> 4) << 4;
            float s = 0.f, ss = 0.f;
            for (int j = 0; j < 16; ++j) { s += tst[2 * (g0 + j)]; ss += tst[2 * (g0 + j) + 1]; }
            const float mean = s * (1.f / 1024.f);
            float var = ss * (1.f / 1024.f) - mean * mean;
            var = var > 0.f ? var : 0.f;
            const float g = a.gn_gamma[c] * rsqrtf(var + 1e-5f);
            sc[c] = g;
            sh[c] = a.gn_beta[c] - mean * g;
        }
    }
    __syncthreads();

    // 3. values
    float scl[8], shl[8];
    static_for<0, 8>([&](auto i_) __attribute__((always_inline)) {
        constexpr int i = decltype(i_)::value;
        scl[i] = gn ? sc[c0 + i] : gatev[i];
        shl[i] = gn ? sh[c0 + i] : 0.f;
    });
    float x[4][8];
    float pe[4][8];                                      // positional encoding [64][C] f32 (stem only): two 16-byte loads per square
    if (a.posenc) {
        static_for<0, 4>([&](auto k_) __attribute__((always_inline)) {
            constexpr int k = decltype(k_)::value;
            const float4 p0 = *reinterpret_cast<const float4*>(a.posenc + (sg * 4 + k) * C + c0);
            const float4 p1 = *reinterpret_cast<const float4*>(a.posenc + (sg * 4 + k) * C + c0 + 4);
            pe[k][0] = p0.x; pe[k][1] = p0.y; pe[k][2] = p0.z; pe[k][3] = p0.w;
            pe[k][4] = p1.x; pe[k][5] = p1.y; pe[k][6] = p1.z; pe[k][7] = p1.w;
        });
    }
    static_for<0, 4>([&](auto k_) __attribute__((always_inline)) {
        constexpr int k = decltype(k_)::value;
        static_for<0, 8>([&](auto i_) __attribute__((always_inline)) {
            constexpr int i = decltype(i_)::value;
            float v = (float)tv[k][i];
            if (gn) v = act_apply(v * scl[i] + shl[i], a.act);
            else if (se) v *= scl[i];
            v += (float)rv[k][i];
            if (a.posenc) v += pe[k][i];
            x[k][i] = v;
        });
    });
    if (a.ln_g != nullptr) {
        // LayerNorm over C per square: partial sums per (square, chunk) -> 8 threads per square -> (mean, rstd)
        float2* part = reinterpret_cast<float2*>(red);          // [64][NC]
        static_for<0, 4>([&](auto k_) __attribute__((always_inline)) {
            constexpr int k = decltype(k_)::value;
            float rs = 0.f, rss = 0.f;
            static_for<0, 8>([&](auto i_) __attribute__((always_inline)) { const float v = x[k][decltype(i_)::value]; rs += v; rss += v * v; });
            part[(sg * 4 + k) * NC + chunk] = make_float2(rs, rss);
        });
        __syncthreads();
        float2* rowp = reinterpret_cast<float2*>(rowbuf);        // [64] (mean, rstd)
        for (int u = tid; u < 512; u += nthr) {                  // 8 consecutive lanes per square
            const int sq = u >> 3, p8 = u & 7;
            float rs = 0.f, rss = 0.f;
            for (int ch = p8; ch < NC; ch += 8) { const float2 v = part[sq * NC + ch]; rs += v.x; rss += v.y; }
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) { rs += __shfl_xor(rs, o); rss += __shfl_xor(rss, o); }
            if (p8 == 0) {
                const float cnt = (float)(a.ln_count > 0 ? a.ln_count : C);
                const float mean = rs / cnt;
                float var = rss / cnt - mean * mean;
                var = var > 0.f ? var : 0.f;
                rowp[sq] = make_float2(mean, rsqrtf(var + 1e-5f));
            }
        }
        __syncthreads();
        float lg[8], lb[8];
        static_for<0, 8>([&](auto i_) __attribute__((always_inline)) {
            constexpr int i = decltype(i_)::value;
            lg[i] = a.ln_g[c0 + i]; lb[i] = a.ln_b[c0 + i];
        });
        static_for<0, 4>([&](auto k_) __attribute__((always_inline)) {
            constexpr int k = decltype(k_)::value;
            const float2 mr = rowp[sg * 4 + k];
            static_for<0, 8>([&](auto i_) __attribute__((always_inline)) {
                constexpr int i = decltype(i_)::value;
                x[k][i] = (x[k][i] - mr.x) * mr.y * lg[i] + lb[i];
            });
        });
        __syncthreads();                                         // part (red) is reused below
    }

    // 4. first output + per-channel statistics
    _Float16* y = a.y + (size_t)b * 64 * C + (size_t)(sg * 4) * C + c0;
    half8 yv[4];
    float csum[8], csq[8];
    static_for<0, 8>([&](auto i_) __attribute__((always_inline)) { csum[decltype(i_)::value] = 0.f; csq[decltype(i_)::value] = 0.f; });
    static_for<0, 4>([&](auto k_) __attribute__((always_inline)) {
        constexpr int k = decltype(k_)::value;
        static_for<0, 8>([&](auto i_) __attribute__((always_inline)) {
            constexpr int i = decltype(i_)::value;
            yv[k][i] = (_Float16)x[k][i];
            csum[i] += x[k][i]; csq[i] += x[k][i] * x[k][i];
        });
        *reinterpret_cast<half8*>(y + k * C) = yv[k];
    });
    if (a.out_stats == nullptr && a.y2 == nullptr) return;
    static_for<0, 8>([&](auto i_) __attribute__((always_inline)) {
        constexpr int i = decltype(i_)::value;
        reinterpret_cast<float2*>(red)[sg * C + c0 + i] = make_float2(csum[i], csq[i]);
    });
    __syncthreads();
    for (int c = tid; c < C; c += nthr) {
        float s = 0.f, ss = 0.f;
        for (int g = 0; g < 16; ++g) { const float2 v = reinterpret_cast<const float2*>(red)[g * C + c]; s += v.x; ss += v.y; }
        tot[2 * c] = s; tot[2 * c + 1] = ss;
        if (a.out_stats != nullptr) {
            a.out_stats[((size_t)b * C + c) * 2] = s;
            a.out_stats[((size_t)b * C + c) * 2 + 1] = ss;
        }
    }
    if (a.y2 == nullptr) return;
    __syncthreads();
    // 5. second output from the register copy of y
    if (tid < C) {                                       // nthr == 2 C
        const int c = tid, g0 = (c >> 4) << 4;
        float s = 0.f, ss = 0.f;
        for (int j = 0; j < 16; ++j) { s += tot[2 * (g0 + j)]; ss += tot[2 * (g0 + j) + 1]; }
        const float mean = s * (1.f / 1024.f);
        float var = ss * (1.f / 1024.f) - mean * mean;
        var = var > 0.f ? var : 0.f;
        const float g = g2w * rsqrtf(var + 1e-5f);
        sc[c] = g;
        sh[c] = g2b - mean * g;
    }
    __syncthreads();
    _Float16* y2 = a.y2 + (size_t)b * 64 * C + (size_t)(sg * 4) * C + c0;
    static_for<0, 8>([&](auto i_) __attribute__((always_inline)) { constexpr int i = decltype(i_)::value; scl[i] = sc[c0 + i]; shl[i] = sh[c0 + i]; });
    static_for<0, 4>([&](auto k_) __attribute__((always_inline)) {
        constexpr int k = decltype(k_)::value;
        half8 ov;
        static_for<0, 8>([&](auto i_) __attribute__((always_inline)) {
            constexpr int i = decltype(i_)::value;
            ov[i] = (_Float16)act_apply((float)yv[k][i] * scl[i] + shl[i], a.act);
        });
        *reinterpret_cast<half8*>(y2 + k * C) = ov;
    });
}

hipError_t launch_ew_board(const EwArgs& a, int boards, hipStream_t st) {
    if (a.C > 384 || a.C % 32 != 0) return hipErrorInvalidValue;
    const int nthr = 2 * a.C;                           // (C/8 chunks) x 16 square groups
    const size_t lds = (size_t)(4 * a.C + 128 + 32 * a.C) * 4;
    hipLaunchKernelGGL(ew_board_kernel, dim3(boards), dim3(nthr), lds, st, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// se_gate: squeeze-excite gate of every board (resnet.py:59-68), gate = sigmoid(W2 act(W1 pool + b1) + b2) with
// pool = per-channel mean over the 64 squares, taken from the conv epilogue's sums.  One workgroup = 8 boards, C
// threads.  The gate is a chain of dependent global-memory latencies, so it runs here, once, with every weight
// fetched in batches of 16 independent loads and used for 8 boards, instead of inside each board's ew_board
// workgroup (measured there: +100 us per call).
// The multiply-adds are explicit fused operations.  Written as `s[q] += w * x` over the 8 boards of a thread, hipcc fused some
// of the eight and compiled the others as a packed multiply followed by a packed add (two roundings): the gate of a board then
// depended, in the last bit, on its position in the batch modulo 8 -- found in round 4 as root values that differed by 1e-6
// from run to run whenever two games raced for batch rows (tools/race_screen_small.py), and as a self-play test that failed
// once in a few dozen runs.  (The 320-wide path has its own squeeze-excite in conv_zs_tail.h and was never affected.)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(384) void se_gate_kernel(SeGateArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int C = a.C, Hd = a.hidden;
    float* pool = reinterpret_cast<float*>(smem);       // [8][C]
    float* part = pool + 8 * C;                         // [parts][8][Hd]
    float* hid = part + 8 * C;                          // [8][Hd]   (parts * Hd <= C)
    const int tid = threadIdx.x, nthr = blockDim.x;     // == C
    const int b0 = blockIdx.x * 8;
    const int nb = a.B - b0 < 8 ? a.B - b0 : 8;
    for (int q = 0; q < 8; ++q)
        pool[q * C + tid] = q < nb ? a.t_stats[((size_t)(b0 + q) * C + tid) * 2] * (1.f / 64.f) : 0.f;
    __syncthreads();
    const int parts = nthr / Hd;                        // >= 1 (launcher)
    {
        const int j = tid % Hd, p = tid / Hd;
        if (p < parts) {
            const int cpp = (C + parts - 1) / parts;
            const int cbeg = p * cpp, cend = cbeg + cpp < C ? cbeg + cpp : C;
            float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int cb = cbeg; cb < cend; cb += 16) {
                float w[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) w[u] = cb + u < cend ? a.w1[(size_t)(cb + u) * Hd + j] : 0.f;
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int c = cb + u < cend ? cb + u : cbeg;      // w[u] == 0 past the end
#pragma unroll
                    for (int q = 0; q < 8; ++q) s[q] = __builtin_fmaf(w[u], pool[q * C + c], s[q]);   // explicit: see the header
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) part[(p * 8 + q) * Hd + j] = s[q];
        }
    }
    __syncthreads();
    for (int i = tid; i < 8 * Hd; i += nthr) {
        const int q = i / Hd, j = i - q * Hd;
        float s = a.b1[j];
        for (int p = 0; p < parts; ++p) s += part[(p * 8 + q) * Hd + j];
        hid[q * Hd + j] = act_apply(s, a.act);
    }
    __syncthreads();
    {
        const int c = tid;
        const float bias = a.b2[c];
        float s[8] = {bias, bias, bias, bias, bias, bias, bias, bias};
        for (int jb = 0; jb < Hd; jb += 16) {
            float w[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) w[u] = jb + u < Hd ? a.w2[(size_t)(jb + u) * C + c] : 0.f;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int j = jb + u < Hd ? jb + u : 0;
#pragma unroll
                for (int q = 0; q < 8; ++q) s[q] = __builtin_fmaf(w[u], hid[q * Hd + j], s[q]);
            }
        }
        for (int q = 0; q < nb; ++q) a.gate[(size_t)(b0 + q) * C + c] = 1.f / (1.f + __expf(-s[q]));
    }
}

hipError_t launch_se_gate(const SeGateArgs& a, hipStream_t st) {
    if (a.C > 384 || a.C % 32 != 0 || a.hidden < 1 || a.hidden > a.C) return hipErrorInvalidValue;
    const size_t lds = (size_t)(8 * a.C + 8 * a.C + 8 * a.hidden) * 4;
    hipLaunchKernelGGL(se_gate_kernel, dim3((a.B + 7) / 8), dim3(a.C), lds, st, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// attn_core: ChessAttention.forward lines 142-179 for one (board, head) per wave.
// qkv [B][64][3C] fp16, channel = (t*H + h)*D + d, D == 16.  Lane = query square.
//   S = QK^T/sqrt(D) (+rel_bias[h]) clamp +-50 ; masked branch fill -1e4 ;   (rel_bias arrives times log2 e)
//   out = (1-mix)*softmax(S_masked)V + mix*softmax(S)V     (mix in (0,1))
//   mix >= 1: masked only ; mix <= 0: unmasked only.
// o [B][64][C] fp16 with channel h*D+d.
// ---------------------------------------------------------------------------
#define ATT_BOARDS 16
__global__ __launch_bounds__(256) void attn_core_kernel(AttnArgs a) {
    // One wave per (board, head), both matrix products on MFMA 32x32x16:
    //   S^T = K Q^T  (A = K rows = keys, B = Q^T cols = queries; K = head_dim = 16: one MFMA per 32x32 tile, operands
    //                 straight from global memory) -> a lane owns one query column and 16 keys per tile in registers,
    //                 so the softmax sums need a single cross-half shuffle;
    //   O^T = V^T P^T (A = V^T rows = head dims (16 of the 32 used), B = P^T cols = queries): the B operand IS the
    //                 lane's register block of probabilities -- the contraction index may be enumerated in any order as
    //                 long as A agrees, so A is gathered from a transposed LDS copy of V in the accumulator's key order.
    // The first version did PV on the VALU (1024 FMAs + 256 LDS reads per lane and job): 520 us per call, VALU-bound;
    // the second read rel_bias from global memory per job (32 KB per job, 2.7 GB per call through L2): 311 us.
    // Workgroup = 4 waves = 4 consecutive heads (they share the 128-byte lines of a qkv row), looping over
    // ATT_BOARDS boards (3 workgroups per CU at 154 VGPRs); the 4 heads' relative-position bias sits in LDS (fp16, pre-scaled) for all of them.
    constexpr int VROW = 68;                           // halfs per LDS row (64 keys + pad: 136 B, conflict-free b64 reads)
    __shared__ __attribute__((aligned(16))) _Float16 Vt[4][16][VROW];
    __shared__ __attribute__((aligned(16))) _Float16 Bs[4][64][VROW];
    __shared__ __attribute__((aligned(16))) _Float16 Ms[64][VROW];       // visibility mask as 0/1 (same for every head and board)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = a.H, C = a.C;
    const int hgroups = (H + 3) >> 2;
    const int hg = blockIdx.x % hgroups, bgroup = blockIdx.x / hgroups;
    const int h = hg * 4 + wave;
    const bool hlive = h < H;
    const int r31 = lane & 31, half = lane >> 5;
    if (a.rel_bias != nullptr) {
        for (int i = tid; i < 4 * 64 * 16; i += 256) {             // 4 keys per item
            const int hh = i >> 10, q = (i >> 4) & 63, k4 = (i & 15) * 4;
            if (hg * 4 + hh < H) {
                const float4 v = *reinterpret_cast<const float4*>(a.rel_bias + ((size_t)(hg * 4 + hh) * 64 + q) * 64 + k4);
                typedef _Float16 half4s __attribute__((ext_vector_type(4)));
                *reinterpret_cast<half4s*>(&Bs[hh][q][k4]) = half4s{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
            }
        }
    }
    for (int i = tid; i < 64 * 16; i += 256) {                     // 4 keys per item
        const int q = i >> 4, k4 = (i & 15) * 4;
        const uint64_t m = a.mask[q] >> k4;
        typedef _Float16 half4s __attribute__((ext_vector_type(4)));
        *reinterpret_cast<half4s*>(&Ms[q][k4]) = half4s{(_Float16)(float)(m & 1), (_Float16)(float)((m >> 1) & 1),
                                                        (_Float16)(float)((m >> 2) & 1), (_Float16)(float)((m >> 3) & 1)};
    }
    __syncthreads();
    for (int it = 0; it < ATT_BOARDS; ++it) {
    const int b = bgroup * ATT_BOARDS + it;
    const bool live = hlive && b < a.B;
    if (!live) continue;                               // wave-uniform; no barriers below
    const _Float16* base = a.qkv + (size_t)b * 64 * 3 * C;
    {   // V row `lane` (key) -> Vt[d][key]
        const _Float16* vp = base + (size_t)lane * 3 * C + (2 * H + h) * 16;
        const half8 v0 = *reinterpret_cast<const half8*>(vp), v1 = *reinterpret_cast<const half8*>(vp + 8);
#pragma unroll
        for (int d = 0; d < 8; ++d) { Vt[wave][d][lane] = v0[d]; Vt[wave][8 + d][lane] = v1[d]; }
    }
    // MFMA operands: A = K (rows = keys), B = Q^T (cols = queries); lane holds 8 consecutive head dims
    half8 kf[2], qf[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        kf[t] = *reinterpret_cast<const half8*>(base + (size_t)(t * 32 + r31) * 3 * C + (1 * H + h) * 16 + 8 * half);
        qf[t] = *reinterpret_cast<const half8*>(base + (size_t)(t * 32 + r31) * 3 * C + (0 * H + h) * 16 + 8 * half);
    }
    const float16v zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // Vt[wave] is private to this wave: its LDS writes and reads execute in program order, no barrier
    // A operand of the PV product for S^T tile kt, register block jb (regs 8jb..8jb+7): V[key][d], d = lane & 15,
    // keys kt*32 + 16 jb + 4 half + {0..3} and + 8 more (the accumulator's row order)
    typedef _Float16 half4v __attribute__((ext_vector_type(4)));
    half8 vf[2][2];
    {
        const _Float16* vrow = &Vt[wave][lane & 15][0];
        static_for<0, 2>([&](auto kt_) __attribute__((always_inline)) {
            static_for<0, 2>([&](auto jb_) __attribute__((always_inline)) {
                constexpr int kt = decltype(kt_)::value, jb = decltype(jb_)::value;
                const half4v lo = *reinterpret_cast<const half4v*>(vrow + kt * 32 + 16 * jb + 4 * half);
                const half4v hi = *reinterpret_cast<const half4v*>(vrow + kt * 32 + 16 * jb + 8 + 4 * half);
                vf[kt][jb] = half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            });
        });
    }
    float wm_, wu_;   // output weights of the masked / unmasked branch (resnet.py:154-174)
    if (a.mix > 0.f && a.mix < 1.f) { wm_ = 1.f - a.mix; wu_ = 1.f - (1.f - a.mix); }
    else if (a.mix >= 1.f) { wm_ = 1.f; wu_ = 0.f; }
    else { wm_ = 0.f; wu_ = 1.f; }
    const float isd = a.inv_sqrt_d * 1.44269504088896f;            // scores in log2 units: exp(x) = exp2(x log2 e)
    const float clampv = 50.f * 1.44269504088896f;
    static_for<0, 2>([&](auto qt_) __attribute__((always_inline)) {
        constexpr int qt = decltype(qt_)::value;
        const int q = qt * 32 + r31;
        const half8 qfr = qf[qt];
        // S^T tiles for this query tile: row (key) = (r&3)+8*(r>>2)+4*half (+32 for the 2nd), col (query) = lane&31
        float16v st[2];
        st[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0], qfr, zero, 0, 0, 0);
        st[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[1], qfr, zero, 0, 0, 0);
        const _Float16* rb = a.rel_bias ? &Bs[wave][q][0] : nullptr;
        const _Float16* vm = &Ms[q][0];          // 1 where the mask lets query q see the key, as an fp16 multiplicand
        // scores clamped to [-50,50]: exp needs no max subtraction; masked fill -1e4 underflows to exactly 0
        float e[2][16];
        float su = 0.f, sm = 0.f;
        static_for<0, 2>([&](auto kt_) __attribute__((always_inline)) {
            constexpr int kt = decltype(kt_)::value;
            static_for<0, 4>([&](auto g_) __attribute__((always_inline)) {
                constexpr int g = decltype(g_)::value;
                const int key0 = kt * 32 + 8 * g + 4 * half;
                half4v bias = {0, 0, 0, 0};
                if (rb) bias = *reinterpret_cast<const half4v*>(rb + key0);
                const half4v vis = *reinterpret_cast<const half4v*>(vm + key0);
                static_for<0, 4>([&](auto j_) __attribute__((always_inline)) {
                    constexpr int j = decltype(j_)::value;
                    float d = st[kt][4 * g + j] * isd + (float)bias[j];
                    d = __builtin_amdgcn_fmed3f(d, -clampv, clampv);
                    const float eu = __builtin_amdgcn_exp2f(d);
                    e[kt][4 * g + j] = eu;
                    su += eu;
                    sm += eu * (float)vis[j];
                });
            });
        });
        su += __shfl_xor(su, 32);
        sm += __shfl_xor(sm, 32);
        const float cu = wu_ / su, cm = wm_ / sm;
        // probabilities (both branches folded into one weight) as the B operand, O^T accumulated over the 4 key blocks
        float16v oacc = zero;
        static_for<0, 2>([&](auto kt_) __attribute__((always_inline)) {
            constexpr int kt = decltype(kt_)::value;
            static_for<0, 2>([&](auto jb_) __attribute__((always_inline)) {
                constexpr int jb = decltype(jb_)::value;
                half8 pf;
                // regs 8jb..8jb+3 = keys kt*32 + 16 jb + 4 half + {0..3}, regs +4..+7 = the same + 8
                const half4v v0 = *reinterpret_cast<const half4v*>(vm + kt * 32 + 16 * jb + 4 * half);
                const half4v v1 = *reinterpret_cast<const half4v*>(vm + kt * 32 + 16 * jb + 8 + 4 * half);
                static_for<0, 8>([&](auto u_) __attribute__((always_inline)) {
                    constexpr int u = decltype(u_)::value;
                    constexpr int r = 8 * jb + u;                          // key = kt*32 + (r&3) + 8*(r>>2) + 4*half
                    const float vis = (float)(u < 4 ? v0[u & 3] : v1[u & 3]);
                    pf[u] = (_Float16)(e[kt][r] * (vis * cm + cu));
                });
                oacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[kt][jb], pf, oacc, 0, 0, 0);
            });
        });
        // O^T: lane = query, regs 0..7 = head dims (r&3) + 8*(r>>2) + 4*half -> 16 contiguous bytes per lane after one
        // exchange with the other half
        typedef _Float16 half2v_ __attribute__((ext_vector_type(2)));
        union { half2v_ h2[2]; uint32_t u[2]; } lo4, hi4, rcv;
        lo4.h2[0] = half2v_{(_Float16)oacc[0], (_Float16)oacc[1]}; lo4.h2[1] = half2v_{(_Float16)oacc[2], (_Float16)oacc[3]};
        hi4.h2[0] = half2v_{(_Float16)oacc[4], (_Float16)oacc[5]}; hi4.h2[1] = half2v_{(_Float16)oacc[6], (_Float16)oacc[7]};
        rcv.u[0] = __shfl_xor(half ? lo4.u[0] : hi4.u[0], 32);
        rcv.u[1] = __shfl_xor(half ? lo4.u[1] : hi4.u[1], 32);
        uint4 ov;
        if (half == 0) ov = make_uint4(lo4.u[0], lo4.u[1], rcv.u[0], rcv.u[1]);    // d 0..3 own, 4..7 from the partner
        else ov = make_uint4(rcv.u[0], rcv.u[1], hi4.u[0], hi4.u[1]);              // d 8..11 from the partner, 12..15 own
        *reinterpret_cast<uint4*>(a.o + ((size_t)b * 64 + q) * C + h * 16 + 8 * half) = ov;
    });
    }
}

hipError_t launch_attn_core(const AttnArgs& a, hipStream_t st) {
    const int hgroups = (a.H + 3) / 4, bgroups = (a.B + ATT_BOARDS - 1) / ATT_BOARDS;
    hipLaunchKernelGGL(attn_core_kernel, dim3((unsigned)(hgroups * bgroups)), dim3(256), 0, st, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// planes f32 [B][P][64] -> fp16 [B][64][32] (channels >= P zero)
// ---------------------------------------------------------------------------
__global__ void planes_to_nhwc_kernel(const float* __restrict__ x, _Float16* __restrict__ y, int B, int P) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)B * 64 * 32;
    if (i >= total) return;
    int c = (int)(i % 32);
    long bn = i / 32;
    int n = (int)(bn % 64);
    long b = bn / 64;
    float v = (c < P) ? x[(b * P + c) * 64 + n] : 0.f;
    y[i] = (_Float16)v;
}

hipError_t launch_planes_to_nhwc(const float* x, void* y, int B, int P, hipStream_t st) {
    long total = (long)B * 64 * 32;
    hipLaunchKernelGGL(planes_to_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x,
                       reinterpret_cast<_Float16*>(y), B, P);
    return hipGetLastError();
}

// fp16 [B][64][ld] (first n channels) -> f32 [B][ctot][64] at channel offset coff
// (SSL head outputs, NCHW for the boundary)
__global__ void nhwc_to_nchw_f32_kernel(const _Float16* __restrict__ x, float* __restrict__ y, int B, int ld, int n,
                                        int ctot, int coff) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)B * n * 64;
    if (i >= total) return;
    int sq = (int)(i % 64);
    long bc = i / 64;
    int c = (int)(bc % n);
    long b = bc / n;
    y[(b * ctot + coff + c) * 64 + sq] = (float)x[(b * 64 + sq) * ld + c];
}

hipError_t launch_nhwc_to_nchw_f32(const void* x, float* y, int B, int ld, int n, int ctot, int coff, hipStream_t st) {
    long total = (long)B * n * 64;
    hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
                       reinterpret_cast<const _Float16*>(x), y, B, ld, n, ctot, coff);
    return hipGetLastError();
}
