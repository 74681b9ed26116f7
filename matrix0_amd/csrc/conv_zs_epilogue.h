// Output epilogues of conv_zs_kernel (conv_zs.hip): a wave holds TWO boards x 80 channels as 8 x 5 accumulator tiles of
// v_mfma_f32_16x16x32_f16.  M-tile mi = board row y = mi of both boards (8 squares of board a, 8 of board b).  The MFMAs are
// issued with the WEIGHT fragment as operand A and the activation fragment as operand B, so lane l = (c15 = l & 15, q = l >> 4)
// holds, in register r of tile (mi, ni),
//     board  c15 >> 3  (of the wave's pair),   square  8 mi + (c15 & 7),   channel  16 ni + 4 q + r  (of the wave's quarter):
// four CONSECUTIVE channels of one square -- one 8-byte LDS write per tile and no cross-lane exchange (with the activations as
// operand A a lane held four squares of one channel and neighbouring lanes had to swap values through DPP before every
// ds_write_b32: ~3.5 vector instructions per value in epilogues that are bound by vector issue).
// Finished values go as fp16 into the wave's private LDS image [64 squares][2 boards][80 channels] (320-byte rows, the two
// boards side by side) and leave it as 160-byte row runs; a GroupNorm group (16 channels x 64 squares of one board) is one
// channel tile ni of the 32 lanes with the same c15 >> 3.
#pragma once
#include "conv_epilogue.h"

typedef float float4v __attribute__((ext_vector_type(4)));
typedef _Float16 half4zs __attribute__((ext_vector_type(4)));

__device__ __forceinline__ char* zs_stage_base(char* img, int lane) {
    const int c15 = lane & 15, q = lane >> 4;
    return img + (c15 & 7) * 320 + (c15 >> 3) * 160 + q * 8;
}
template <int MI, int NI>
__device__ __forceinline__ void zs_stage_tile(const float (&v)[4], char* wbase, int lane) {
    (void)lane;
    const half4zs h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
    *reinterpret_cast<half4zs*>(wbase + MI * 8 * 320 + NI * 32) = h;
}

template <int CTRL>
__device__ __forceinline__ float zs_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over the 8 lanes that hold the squares of one board row (same channels; lane bits 0-2): every lane gets the total.
// DPP adds (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror), not LDS-crossbar shuffles.
__device__ __forceinline__ float zs_sum_row(float v) {
    v += zs_dpp<0xB1>(v);
    v += zs_dpp<0x4E>(v);
    v += zs_dpp<0x141>(v);
    return v;
}
// sum over the 32 lanes that hold one board (same c15 >> 3): the 8 squares of the row and the 4 channel quads (lane bits 4-5)
__device__ __forceinline__ float zs_sum_board(float v) {
    v = zs_sum_row(v);
    v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
    return v;
}

// ---- the same helpers for the ACTIVATION-as-operand-A layout (conv2 with the fused tail keeps it: there the squeeze-excite
// pooling and gate want one channel per lane, and the tail measured 0.4 % slower in the channel-quad layout): lane (c15, q) holds
// board q >> 1, square 8 mi + 4 (q & 1) + r, channel 16 ni + c15.
// Lanes l and l^1 (adjacent channels) exchange two values so that the even lane owns squares +0 / +2 and the odd lane squares
// +1 / +3 of a channel PAIR: two ds_write_b32 per tile.
__device__ __forceinline__ char* zsa_stage_base(char* img, int lane) {
    const int q = lane >> 4;
    return img + ((q & 1) * 4 + (lane & 1)) * 320 + (q >> 1) * 160 + ((lane & 15) >> 1) * 4;
}
template <int MI, int NI>
__device__ __forceinline__ void zsa_stage_tile(const float (&v)[4], char* wbase, int lane) {
    const bool odd = (lane & 1) != 0;
    static_for<0, 2>([&](auto p_) __attribute__((always_inline)) {
        constexpr int p = decltype(p_)::value;                     // square pair: 2p (even lane) and 2p + 1 (odd lane)
        const float mine = odd ? v[2 * p + 1] : v[2 * p];
        const float send = odd ? v[2 * p] : v[2 * p + 1];
        const float recv = __builtin_bit_cast(
            float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, true));   // lane ^ 1
        const half2v h = {(_Float16)(odd ? recv : mine), (_Float16)(odd ? mine : recv)};
        *reinterpret_cast<half2v*>(wbase + (MI * 8 + 2 * p) * 320 + NI * 32) = h;
    });
}

// sum over the 32 lanes that hold one board (same q >> 1)
__device__ __forceinline__ float zsa_sum_board(float v) {
#pragma unroll
    for (int o = 1; o <= 16; o <<= 1) v += __shfl_xor(v, o);
    return v;
}

// image -> global: unit u = it * 64 + lane = (square, board, 16-byte chunk of the 80 channels); 160-byte row runs.
// `out` = first row of the wave's board pair, first channel of its quarter.
__device__ __forceinline__ void zs_stage_flush(char* out, uint32_t ldo2, int rows_valid, const char* img, int lane) {
    int sq = lane / 20, c20 = lane - sq * 20;
#pragma unroll
    for (int it = 0; it < 20; ++it) {
        const uint4 v = *reinterpret_cast<const uint4*>(img + (it * 64 + lane) * 16);
        const int bd = c20 >= 10 ? 1 : 0;
        const int row = bd * 64 + sq;
        if (row < rows_valid) *reinterpret_cast<uint4*>(out + ((uint32_t)row * ldo2 + (uint32_t)(c20 - 10 * bd) * 16u)) = v;
        c20 += 4; sq += 3;                                          // 64 = 3 * 20 + 4
        if (c20 >= 20) { c20 -= 20; sq += 1; }
    }
}

// EPI 0: bias / activation ACT / scale, fp16 store, per-(board, channel) sum and sum of squares.
// EPI 1: GroupNorm(16 channels x 64 squares) + activation ACT in registers.
// wp = board pair of the 4-board tile, wn = channel quarter.
template <int EPI, int ACT>
__device__ __forceinline__ void zs_tile_epilogue(float4v (&acc)[8][5], const GemmArgs& a, char* img, int m0, int n0, int wp,
                                                 int wn, int lane) {
    const int c15 = lane & 15, q = lane >> 4;
    const int colbase = n0 + wn * 80 + 4 * q;                       // + 16 ni + r: this lane's four channels of tile ni
    char* wbase = zs_stage_base(img, lane);
    char* out = reinterpret_cast<char*>(a.out) + ((size_t)(m0 + wp * 128) * a.ldo + n0 + wn * 80) * 2;
    const int rows_valid = a.Mvalid - (m0 + wp * 128);
    if constexpr (EPI == 1) {
        // pass 1: statistics -> per-channel scale / shift.  Pass 2 goes board row by board row (M-tile mi = 8 squares of both
        // boards = 8 complete image rows): normalise + activate + stage the row's 5 tiles, then store that row -- the global
        // stores of row mi are in flight while the VALU works on row mi + 1.
        float g[5][4], sh[5][4];
        static_for<0, 5>([&](auto ni_) __attribute__((always_inline)) {
            constexpr int ni = decltype(ni_)::value;
            float s = 0.f, ss = 0.f;
            static_for<0, 8>([&](auto mi_) __attribute__((always_inline)) {
                const float4v av = acc[decltype(mi_)::value][ni];
                static_for<0, 4>([&](auto r_) __attribute__((always_inline)) { const float v = av[decltype(r_)::value]; s += v; ss += v * v; });
            });
            s = zs_sum_board(s); ss = zs_sum_board(ss);
            const float mean = s * (1.f / 1024.f);
            float var = ss * (1.f / 1024.f) - mean * mean;
            var = var > 0.f ? var : 0.f;
            const float rstd = rsqrtf(var + 1e-5f);
            const float4 gm = *reinterpret_cast<const float4*>(a.gn_gamma + colbase + ni * 16);
            const float4 bt = *reinterpret_cast<const float4*>(a.gn_beta + colbase + ni * 16);
            g[ni][0] = rstd * gm.x; g[ni][1] = rstd * gm.y; g[ni][2] = rstd * gm.z; g[ni][3] = rstd * gm.w;
            sh[ni][0] = bt.x - mean * g[ni][0]; sh[ni][1] = bt.y - mean * g[ni][1];
            sh[ni][2] = bt.z - mean * g[ni][2]; sh[ni][3] = bt.w - mean * g[ni][3];
        });
        const uint32_t ldo2 = (uint32_t)a.ldo * 2u;
        // row mi of the image = 8 squares x 20 chunks = 160 16-byte units: lanes 0..63 take units lane, 64 + lane and (lanes < 32)
        // 128 + lane; unit u = (square u / 20, board-chunk u % 20)
        int fsq[3], fc20[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { const int u = 64 * k + lane; fsq[k] = u / 20; fc20[k] = u - fsq[k] * 20; }
        static_for<0, 8>([&](auto mi_) __attribute__((always_inline)) {
            constexpr int mi = decltype(mi_)::value;
            static_for<0, 5>([&](auto ni_) __attribute__((always_inline)) {
                constexpr int ni = decltype(ni_)::value;
                float v[4];
                static_for<0, 4>([&](auto r_) __attribute__((always_inline)) {
                    constexpr int r = decltype(r_)::value;
                    v[r] = act_fast<ACT>(acc[mi][ni][r] * g[ni][r] + sh[ni][r]);
                });
                zs_stage_tile<mi, ni>(v, wbase, lane);
            });
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (k < 2 || lane < 32) {
                    const uint4 v = *reinterpret_cast<const uint4*>(img + (mi * 160 + 64 * k + lane) * 16);
                    const int bd = fc20[k] >= 10 ? 1 : 0;
                    const int row = bd * 64 + mi * 8 + fsq[k];
                    if (row < rows_valid)
                        *reinterpret_cast<uint4*>(out + ((uint32_t)row * ldo2 + (uint32_t)(fc20[k] - 10 * bd) * 16u)) = v;
                }
            }
        });
    }
    if constexpr (EPI == 0) {
        const float oscale = a.out_scale;
        const bool want_stats = a.out_stats != nullptr;
        float* stats = a.out_stats + ((size_t)(m0 / 64 + 2 * wp + (c15 >> 3)) * a.N + colbase) * 2;
        static_for<0, 5>([&](auto ni_) __attribute__((always_inline)) {
            constexpr int ni = decltype(ni_)::value;
            float bias[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.bias != nullptr) {
                const float4 b4 = *reinterpret_cast<const float4*>(a.bias + colbase + ni * 16);
                bias[0] = b4.x; bias[1] = b4.y; bias[2] = b4.z; bias[3] = b4.w;
            }
            float s[4] = {0.f, 0.f, 0.f, 0.f}, ss[4] = {0.f, 0.f, 0.f, 0.f};
            static_for<0, 8>([&](auto mi_) __attribute__((always_inline)) {
                constexpr int mi = decltype(mi_)::value;
                float v[4];
                static_for<0, 4>([&](auto r_) __attribute__((always_inline)) {
                    constexpr int r = decltype(r_)::value;
                    v[r] = act_fast<ACT>(acc[mi][ni][r] + bias[r]) * oscale;
                    s[r] += v[r]; ss[r] += v[r] * v[r];
                });
                zs_stage_tile<mi, ni>(v, wbase, lane);
            });
            if (want_stats) {
                static_for<0, 4>([&](auto r_) __attribute__((always_inline)) {
                    constexpr int r = decltype(r_)::value;
                    const float st = zs_sum_row(s[r]), sst = zs_sum_row(ss[r]);
                    if ((c15 & 7) == 0) { stats[(ni * 16 + r) * 2] = st; stats[(ni * 16 + r) * 2 + 1] = sst; }
                });
            }
        });
        zs_stage_flush(out, (uint32_t)a.ldo * 2u, rows_valid, img, lane);
    }
}
