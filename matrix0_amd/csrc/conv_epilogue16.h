// Output epilogues of conv_pp16_kernel: the wave's 64 squares x 160 channels as 4 x 10 accumulator tiles of
// v_mfma_f32_16x16x32_f16.  Lane l = (c15 = l & 15, q = l >> 4) holds, in register r of tile (mi, ni), the value of
//     square 16 mi + 4 q + r,   channel 16 ni + c15          (of the wave's board / channel half)
// Staging, flush, statistics and the residual-block tail do what conv_epilogue.h / conv_tail.h do for the 32x32 layout:
// finished values go as fp16 into the wave's private LDS image [64 rows][160 ch] and leave it in full 320-byte row runs.
#pragma once
#include "conv_epilogue.h"

typedef float float4v __attribute__((ext_vector_type(4)));

// Lanes l and l^1 (adjacent channels) exchange two values so that the even lane owns rows 4q+0 / 4q+2 and the odd lane
// rows 4q+1 / 4q+3 of a channel PAIR: two ds_write_b32 per tile, rows of different parity -> different banks.
__device__ __forceinline__ char* conv_stage_base16(char* lds_wave, int lane) {
    return lds_wave + (4 * (lane >> 4) + (lane & 1)) * 320 + ((lane & 15) >> 1) * 4;
}
template <int MI, int NI>
__device__ __forceinline__ void conv_stage_tile16(const float (&v)[4], char* wbase, int lane) {
    const bool odd = (lane & 1) != 0;
    static_for<0, 2>([&](auto p_) __attribute__((always_inline)) {
        constexpr int p = decltype(p_)::value;                     // row pair: rows 2p (even lane) and 2p + 1 (odd lane)
        const float mine = odd ? v[2 * p + 1] : v[2 * p];
        const float send = odd ? v[2 * p] : v[2 * p + 1];
        const float recv = __builtin_bit_cast(
            float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, true));   // lane ^ 1
        const half2v h = {(_Float16)(odd ? recv : mine), (_Float16)(odd ? mine : recv)};
        *reinterpret_cast<half2v*>(wbase + (MI * 16 + 2 * p) * 320 + NI * 32) = h;
    });
}

// sum over the 64 lanes
__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
    for (int o = 1; o <= 32; o <<= 1) v += __shfl_xor(v, o);
    return v;
}

// EPI 0: bias / activation ACT / scale, fp16 store, per-(board, channel) sum and sum of squares.
// EPI 1: GroupNorm(16 channels x 64 squares) + activation ACT in registers: group = channel tile ni, whole in the wave.
template <int EPI, int ACT>
__device__ __forceinline__ void conv_tile_epilogue16(float4v (&acc)[4][10], const GemmArgs& a, char* lds_wave, int m0,
                                                     int n0, int wm, int wn, int lane) {
    const int c15 = lane & 15;
    const int colbase = n0 + wn * 160 + c15;
    char* wbase = conv_stage_base16(lds_wave, lane);
    if constexpr (EPI == 1) {
        static_for<0, 10>([&](auto ni_) __attribute__((always_inline)) {
            constexpr int ni = decltype(ni_)::value;
            const int col = colbase + ni * 16;
            float s = 0.f, ss = 0.f;
            static_for<0, 4>([&](auto mi_) __attribute__((always_inline)) {
                const float4v av = acc[decltype(mi_)::value][ni];
                static_for<0, 4>([&](auto r_) __attribute__((always_inline)) { const float v = av[decltype(r_)::value]; s += v; ss += v * v; });
            });
            s = wave_sum64(s); ss = wave_sum64(ss);
            const float mean = s * (1.f / 1024.f);
            float var = ss * (1.f / 1024.f) - mean * mean;
            var = var > 0.f ? var : 0.f;
            const float g = rsqrtf(var + 1e-5f) * a.gn_gamma[col];
            const float sh = a.gn_beta[col] - mean * g;
            static_for<0, 4>([&](auto mi_) __attribute__((always_inline)) {
                constexpr int mi = decltype(mi_)::value;
                float v[4];
                static_for<0, 4>([&](auto r_) __attribute__((always_inline)) {
                    constexpr int r = decltype(r_)::value;
                    v[r] = act_fast<ACT>(acc[mi][ni][r] * g + sh);
                });
                conv_stage_tile16<mi, ni>(v, wbase, lane);
            });
        });
        conv_stage_flush<5>(a, lds_wave, m0, n0, wm, wn, lane);
    }
    if constexpr (EPI == 0) {
        const float oscale = a.out_scale;
        const bool want_stats = a.out_stats != nullptr;
        float* stats = a.out_stats + ((size_t)(m0 / 64 + wm) * a.N + colbase) * 2;
        static_for<0, 10>([&](auto ni_) __attribute__((always_inline)) {
            constexpr int ni = decltype(ni_)::value;
            const float bias = a.bias != nullptr ? a.bias[colbase + ni * 16] : 0.f;
            float s = 0.f, ss = 0.f;
            static_for<0, 4>([&](auto mi_) __attribute__((always_inline)) {
                constexpr int mi = decltype(mi_)::value;
                float v[4];
                static_for<0, 4>([&](auto r_) __attribute__((always_inline)) {
                    constexpr int r = decltype(r_)::value;
                    v[r] = act_fast<ACT>(acc[mi][ni][r] + bias) * oscale;
                    s += v[r]; ss += v[r] * v[r];
                });
                conv_stage_tile16<mi, ni>(v, wbase, lane);
            });
            if (want_stats) {
                s += __shfl_xor(s, 16); ss += __shfl_xor(ss, 16);
                s += __shfl_xor(s, 32); ss += __shfl_xor(ss, 32);
                if (lane < 16) { stats[ni * 32] = s; stats[ni * 32 + 1] = ss; }
            }
        });
        conv_stage_flush<5>(a, lds_wave, m0, n0, wm, wn, lane);
    }
}
