// Residual-block tail fused into conv2's epilogue, for conv_zs_kernel's accumulator layout (conv_zs_epilogue.h; the
// arithmetic is conv_tail.h's):
//   t      = conv2 output (the workgroup holds 4 whole boards x all 320 channels in its accumulators)
//   gate   = sigmoid(W2 act(W1 mean_squares(t) + b1) + b2)            squeeze-excite, resnet.py:59-68 (optional)
//   y      = x + gate * t                                             the residual stream          -> a.out
//   y2     = act(GroupNorm16(y; next block's bn1))                    the next conv1's operand     -> a.y2 (optional)
// A wave = (board pair wp, channel quarter wn): 2 boards x 80 channels, accumulators in the activation-as-operand-A layout
// (conv_zs_epilogue.h, zsa_* helpers): lane (c15, q) holds board q >> 1, square 8 mi + 4 (q & 1) + r, channel 16 ni + c15.
//
// Squeeze-excite on the matrix cores.  The two small FCs used to be scalar loops over LDS (5 barriers, ~7 us per tile, most
// of it the 100 KB of fp32->fp16 weights arriving from L2).  Here:
//   A  channel means of t from the accumulators (in-lane sums + one shuffle), written as fp16 hi + lo parts (the pair
//      carries ~22 mantissa bits) in A-fragment order: poolA[hi|lo][board][320]
//   B  FC1 as 16x16x32 MFMAs: rows = boards, K = 320 channels, N = hidden units; wave nt computes hidden units 16 nt..16 nt+15
//      (B fragments = host-packed pieces of W1, GemmArgs::se_wf), adds b1, activates, writes hidden as hi + lo fp16
//   C  FC2 per wave for its own 80 channels: rows = boards ordered so that accumulator row 4q + r is board q >> 1 of the
//      wave's pair -- the gate of (board, channel) comes out in exactly the lane that scales that accumulator column.
// Three workgroup barriers.  The weight pieces are one stream of 1-KiB DMA pieces (se_wf): 10 x ceil(Hd/16) of W1, then
// 20 x ceil(Hd/32) of W2.
//
// D  gate * t staged as fp16 in the wave's private image [64 squares][2 boards][80 ch]; the loads of x are issued between
//    the tile columns into the registers the staged accumulators free
// E  lane = (16-byte channel chunk of a board, square mod 3): add x, store y, write y back to the image, per-channel sums ->
//    GroupNorm statistics by shuffles;  F  second pass over the image: y2 = act(y * scale + shift), 16-byte stores.
// D-F touch only the wave's own image: no workgroup barrier after C.
#pragma once
#include "conv_zs_epilogue.h"

constexpr int ZS_SE_HMAX = 96;            // squeeze-excite hidden units the fused tail takes (LDS: 60 + 60 pieces)
constexpr int ZS_SE_WOFF = 8192;          // LDS offset of the weight pieces during the squeeze-excite phase

// number of 1-KiB pieces of GemmArgs::se_wf for Hd hidden units (net.hip packs them, pack_se_fragments)
__host__ __device__ inline int zs_se_pieces_w1(int Hd) { return 10 * ((Hd + 15) >> 4); }
__host__ __device__ inline int zs_se_pieces_w2(int Hd) { return 20 * ((Hd + 31) >> 5); }

__device__ __forceinline__ void zs_tail_glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

#ifdef SW_STAMP
__device__ unsigned long long* g_zs_tail_stamp;    // [blocks][8] realtime stamps of the tail's phases (tools/ubench)
#define ZS_TAIL_STAMP(k) do { if (tid == 0) g_zs_tail_stamp[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ZS_TAIL_STAMP(k) do {} while (0)
#endif

// PRE: t is first normalised and activated, t <- act(GroupNorm16(t; a.pre_gamma, a.pre_beta)) -- the chess-feature
// convs (resnet.py:229-244: x += act(norm(conv(x)))); no squeeze-excite in that case.
template <int ACT, bool PRE = false>
__device__ __forceinline__ void zs_tail_epilogue(float4v (&acc)[8][5], const GemmArgs& a, char* smem, int m0, int wp, int wn,
                                                 int wave, int lane) {
    constexpr int NG = 5, MT = 8;
    const int tid = wave * 64 + lane;
    const int c15 = lane & 15, q = lane >> 4;
    const bool se = a.se_w1 != nullptr;
    // phase E/F lane mapping and the second output's GroupNorm parameters (fetched now: a late load is an exposed
    // global-memory latency in a kernel with one workgroup per CU)
    constexpr int NCH = 20;                                      // 16-byte chunks per image row: 2 boards x 10
    constexpr int NIT = 22;                                      // ceil(64 / 3)
    const int c20 = lane % NCH, rsub = lane / NCH;
    const bool lane_on = rsub < 3;
    const int bd = c20 >= 10 ? 1 : 0, ch = c20 - 10 * bd;        // board of the pair, chunk of its 80 channels
    float gg[8], bb[8];
    if (a.y2 != nullptr) {
        const int c0 = wn * 80 + ch * 8;
        const float4 g0 = *reinterpret_cast<const float4*>(a.gn_gamma + c0), g1 = *reinterpret_cast<const float4*>(a.gn_gamma + c0 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(a.gn_beta + c0), b1 = *reinterpret_cast<const float4*>(a.gn_beta + c0 + 4);
        gg[0] = g0.x; gg[1] = g0.y; gg[2] = g0.z; gg[3] = g0.w; gg[4] = g1.x; gg[5] = g1.y; gg[6] = g1.z; gg[7] = g1.w;
        bb[0] = b0.x; bb[1] = b0.y; bb[2] = b0.z; bb[3] = b0.w; bb[4] = b1.x; bb[5] = b1.y; bb[6] = b1.z; bb[7] = b1.w;
    }
    float gv[NG];
    ZS_TAIL_STAMP(0);
#ifdef TAIL_NO_SE
    if (false) {
#else
    if (se && !PRE) {
#endif
        const int Hd = a.se_hidden;
        const int NT1 = (Hd + 15) >> 4, KS2 = (Hd + 31) >> 5;
        const int n1 = 10 * NT1, npieces = n1 + 20 * KS2;
        _Float16* poolA = reinterpret_cast<_Float16*>(smem);           // [2][4][320]  5120 B
        _Float16* hidA = reinterpret_cast<_Float16*>(smem + 5120);     // [2][4][128]  2048 B, zero beyond the hidden units
        const char* wf = smem + ZS_SE_WOFF;
        for (int p = wave; p < npieces; p += 8)
            zs_tail_glds16(reinterpret_cast<const char*>(a.se_wf) + (size_t)p * 1024 + lane * 16, smem + ZS_SE_WOFF + p * 1024);
        if (tid < 128) reinterpret_cast<uint4*>(hidA)[tid] = make_uint4(0, 0, 0, 0);
        const int j1 = 16 * wave + c15;
        const float b1v = (wave < NT1 && j1 < Hd) ? a.se_b1[j1] : 0.f;
        float b2v[NG];
        static_for<0, NG>([&](auto ni_) __attribute__((always_inline)) {
            b2v[decltype(ni_)::value] = a.se_b2[wn * 80 + decltype(ni_)::value * 16 + c15];
        });
        // A: channel means of the lane's (board, 5 channels)
        static_for<0, NG>([&](auto ni_) __attribute__((always_inline)) {
            constexpr int ni = decltype(ni_)::value;
            float s = 0.f;
            static_for<0, MT>([&](auto mi_) __attribute__((always_inline)) {
                static_for<0, 4>([&](auto r_) __attribute__((always_inline)) { s += acc[decltype(mi_)::value][ni][decltype(r_)::value]; });
            });
            s += __shfl_xor(s, 16);
            const float mean = s * (1.f / 64.f);
            const _Float16 hi = (_Float16)mean;
            const _Float16 lo = (_Float16)(mean - (float)hi);
            if ((q & 1) == 0) {
                const int o = (2 * wp + (q >> 1)) * 320 + wn * 80 + ni * 16 + c15;
                poolA[o] = hi; poolA[4 * 320 + o] = lo;
            }
        });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // B: hidden units 16 wave .. 16 wave + 15 of the 4 boards (A rows: board = row & 3)
        if (wave < NT1) {
            float4v h0 = {0.f, 0.f, 0.f, 0.f}, h1 = {0.f, 0.f, 0.f, 0.f};
            const _Float16* pa = poolA + (c15 & 3) * 320 + 8 * q;
            const char* wb = wf + (size_t)(wave * 10) * 1024 + lane * 16;
#pragma unroll
            for (int ks = 0; ks < 10; ++ks) {
                const half8 ah = *reinterpret_cast<const half8*>(pa + 32 * ks);
                const half8 al = *reinterpret_cast<const half8*>(pa + 4 * 320 + 32 * ks);
                const half8 bw = *reinterpret_cast<const half8*>(wb + ks * 1024);
                h0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bw, h0, 0, 0, 0);
                h1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bw, h1, 0, 0, 0);
            }
            if (q == 0) {                                            // rows 0..3 = boards 0..3, column c15 = hidden unit j1
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float hv = act_fast<ACT>(h0[r] + h1[r] + b1v);
                    const _Float16 hi = (_Float16)hv;
                    hidA[r * 128 + j1] = hi;
                    hidA[4 * 128 + r * 128 + j1] = (_Float16)(hv - (float)hi);
                }
            }
        }
        __syncthreads();
        // C: gate of this wave's 2 boards x 80 channels (A rows 0..7: board a of the pair, 8..15: board b)
        {
            float4v g0[NG], g1[NG];
            static_for<0, NG>([&](auto ni_) __attribute__((always_inline)) {
                g0[decltype(ni_)::value] = float4v{0.f, 0.f, 0.f, 0.f}; g1[decltype(ni_)::value] = float4v{0.f, 0.f, 0.f, 0.f};
            });
            const _Float16* ha = hidA + (2 * wp + (c15 >> 3)) * 128 + 8 * q;
            const char* wb = wf + (size_t)n1 * 1024 + (size_t)(5 * wn * KS2) * 1024 + lane * 16;
            for (int ks = 0; ks < KS2; ++ks) {
                const half8 ah = *reinterpret_cast<const half8*>(ha + 32 * ks);
                const half8 al = *reinterpret_cast<const half8*>(ha + 4 * 128 + 32 * ks);
                static_for<0, NG>([&](auto ni_) __attribute__((always_inline)) {
                    constexpr int ni = decltype(ni_)::value;
                    const half8 bw = *reinterpret_cast<const half8*>(wb + (size_t)(ni * KS2 + ks) * 1024);
                    g0[ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bw, g0[ni], 0, 0, 0);
                    g1[ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bw, g1[ni], 0, 0, 0);
                });
            }
            static_for<0, NG>([&](auto ni_) __attribute__((always_inline)) {
                constexpr int ni = decltype(ni_)::value;
                gv[ni] = __builtin_amdgcn_rcpf(1.f + __expf(-(g0[ni][0] + g1[ni][0] + b2v[ni])));
            });
        }
        __syncthreads();                                          // the images below overwrite the pools and the weights
    } else {
        static_for<0, NG>([&](auto ni_) __attribute__((always_inline)) { gv[decltype(ni_)::value] = 1.f; });
    }
    float pv[NG];                                                 // PRE: per-column shift (gv = scale)
    if constexpr (PRE) {
        // GroupNorm(16 channels x 64 squares) of t on the accumulators: a board's group is 32 lanes of this wave
        static_for<0, NG>([&](auto ni_) __attribute__((always_inline)) {
            constexpr int ni = decltype(ni_)::value;
            const int col = wn * 80 + ni * 16 + c15;
            float s = 0.f, ss = 0.f;
            static_for<0, MT>([&](auto mi_) __attribute__((always_inline)) {
                const float4v av = acc[decltype(mi_)::value][ni];
                static_for<0, 4>([&](auto r_) __attribute__((always_inline)) { const float v = av[decltype(r_)::value]; s += v; ss += v * v; });
            });
            s = zsa_sum_board(s); ss = zsa_sum_board(ss);
            const float mean = s * (1.f / 1024.f);
            float var = ss * (1.f / 1024.f) - mean * mean;
            var = var > 0.f ? var : 0.f;
            gv[ni] = rsqrtf(var + 1e-5f) * a.pre_gamma[col];
            pv[ni] = a.pre_beta[col] - mean * gv[ni];
        });
    }

    ZS_TAIL_STAMP(1);
    // D: gate * t -> the wave's fp16 image; the loads of x are issued between the tile columns, into the registers
    // the staged accumulators free (x is 2-3 us away and nothing else runs on this CU)
    char* img = smem + wave * 20480;
    char* wbase = zsa_stage_base(img, lane);
    const uint32_t ldo2 = (uint32_t)a.ldo * 2u;
    const size_t tile_off = ((size_t)(m0 + wp * 128) * a.ldo + wn * 80) * 2;      // wave-uniform
    const char* xin = reinterpret_cast<const char*>(a.res) + tile_off;
    char* yout = reinterpret_cast<char*>(a.out) + tile_off;
    const int rows_valid = a.Mvalid - (m0 + wp * 128 + bd * 64);                  // of this lane's board
    const uint32_t lane_goff = (uint32_t)(bd * 64 + rsub) * ldo2 + (uint32_t)ch * 16u;
    const uint32_t lane_loff = (uint32_t)(rsub * NCH + c20) * 16u;
    half8 xv[NIT];
    static_for<0, NG>([&](auto ni_) __attribute__((always_inline)) {
        constexpr int ni = decltype(ni_)::value;
        static_for<0, MT>([&](auto mi_) __attribute__((always_inline)) {
            constexpr int mi = decltype(mi_)::value;
            float v[4];
            static_for<0, 4>([&](auto r_) __attribute__((always_inline)) {
                constexpr int r = decltype(r_)::value;
                if constexpr (PRE) v[r] = act_fast<ACT>(acc[mi][ni][r] * gv[ni] + pv[ni]);
                else v[r] = acc[mi][ni][r] * gv[ni];
            });
            zsa_stage_tile<mi, ni>(v, wbase, lane);
        });
        __builtin_amdgcn_sched_barrier(0);
        // the 22 loads of x spread over the 5 channel tiles (5 after each of the first two, 4 after the others)
        static_for<(ni < 2 ? ni * 5 : 10 + (ni - 2) * 4), (ni < 2 ? ni * 5 + 5 : 10 + (ni - 1) * 4)>([&](auto it_) __attribute__((always_inline)) {
            constexpr int it = decltype(it_)::value;
            const int row = rsub + 3 * it;
#ifdef TAIL_NO_XLOAD
            xv[it] = half8{0, 0, 0, 0, 0, 0, 0, 0};
#else
            xv[it] = (lane_on && row < 64) ? *reinterpret_cast<const half8*>(xin + (lane_goff + (uint32_t)(3 * it) * ldo2))
                                           : half8{0, 0, 0, 0, 0, 0, 0, 0};
#endif
        });
        __builtin_amdgcn_sched_barrier(0);
    });

    ZS_TAIL_STAMP(2);
    // E: y = x + image; lane = (chunk of 8 channels of a board, square mod 3), squares rsub, rsub+3, ...; lanes 60..63 idle.
    // The sum of two fp16 numbers rounded to fp16 is what the fp32 add + conversion gives, so y is computed with packed
    // fp16 adds (4 instructions per 8 channels); the GroupNorm sums (this lane's 8 channels x its squares) use the
    // 2-element fp16 dot product with fp32 accumulation, on the rounded y (the tensor that is actually stored).
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 ones = {(_Float16)1.f, (_Float16)1.f};
    float gs = 0.f, gss = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int row = rsub + 3 * it;
        if (lane_on && row < 64) {
            half8* ip = reinterpret_cast<half8*>(img + lane_loff + (uint32_t)(3 * it * NCH) * 16u);
            const half8 yv = *ip + xv[it];
            static_for<0, 4>([&](auto i_) __attribute__((always_inline)) {
                constexpr int i = decltype(i_)::value;
                const h2 p = {yv[2 * i], yv[2 * i + 1]};
                gs = __builtin_amdgcn_fdot2(p, ones, gs, false);
                gss = __builtin_amdgcn_fdot2(p, p, gss, false);
            });
            *ip = yv;
            if (row < rows_valid) *reinterpret_cast<half8*>(yout + (lane_goff + (uint32_t)(3 * it) * ldo2)) = yv;
        }
    }
    ZS_TAIL_STAMP(3);
#ifdef TAIL_NO_Y2
    return;
#endif
    if (a.y2 == nullptr) return;

    // GroupNorm statistics of y: over the 3 square classes (lanes c20, c20+20, c20+40), then over the group's 16
    // channels = this lane's 8 + the neighbour chunk's 8 (c20 ^ 1: same board, since a board has an even number of chunks)
    {
        const float s1 = __shfl(gs, c20 + NCH), s2 = __shfl(gs, c20 + 2 * NCH);
        const float q1 = __shfl(gss, c20 + NCH), q2 = __shfl(gss, c20 + 2 * NCH);
        gs = __shfl(gs, c20) + s1 + s2;                          // every lane: totals of its chunk (same order everywhere)
        gss = __shfl(gss, c20) + q1 + q2;
        const float so = __shfl_xor(gs, 1), qo = __shfl_xor(gss, 1);      // partner chunk (c20 ^ 1 is lane ^ 1 for lanes < 60)
        const float lo_s = (c20 & 1) ? so : gs, hi_s = (c20 & 1) ? gs : so;
        const float lo_q = (c20 & 1) ? qo : gss, hi_q = (c20 & 1) ? gss : qo;
        gs = lo_s + hi_s; gss = lo_q + hi_q;
    }
    const float mean = gs * (1.f / 1024.f);
    float var = gss * (1.f / 1024.f) - mean * mean;
    var = var > 0.f ? var : 0.f;
    const float rstd = rsqrtf(var + 1e-5f);
    float scl[8], shl[8];
    static_for<0, 8>([&](auto i_) __attribute__((always_inline)) {
        constexpr int i = decltype(i_)::value;
        scl[i] = gg[i] * rstd; shl[i] = bb[i] - mean * scl[i];
    });
    // F: y2 = act(GroupNorm(y)) from the image
    char* y2out = reinterpret_cast<char*>(a.y2) + tile_off;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int row = rsub + 3 * it;
        if (lane_on && row < 64 && row < rows_valid) {
            const half8 yv = *reinterpret_cast<const half8*>(img + lane_loff + (uint32_t)(3 * it * NCH) * 16u);
            half8 ov;
            static_for<0, 8>([&](auto i_) __attribute__((always_inline)) {
                constexpr int i = decltype(i_)::value;
                ov[i] = (_Float16)act_fast<ACT>((float)yv[i] * scl[i] + shl[i]);
            });
            *reinterpret_cast<half8*>(y2out + (lane_goff + (uint32_t)(3 * it) * ldo2)) = ov;
        }
    }
    ZS_TAIL_STAMP(4);
}
