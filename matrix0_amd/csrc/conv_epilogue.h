// Output epilogues of the 256x320 conv tile (conv_big_kernel; the 32x32x16 3x3 experiment tools/ubench/conv_pp.hip uses them too).
#pragma once
#include "kernel_common.h"

typedef _Float16 half2v __attribute__((ext_vector_type(2)));

// Activation with a compile-time kind (a kernel template parameter: a runtime switch per element, or a switch
// around the whole epilogue, wrecks the register allocation of the 160 accumulators) and the hardware reciprocal (1 ulp) instead of an IEEE division: the result is rounded to fp16 right after.
template <int ACT>
__device__ __forceinline__ float act_fast(float v) {
    if constexpr (ACT == ACT_RELU) return v > 0.f ? v : 0.f;
    else if constexpr (ACT == ACT_SILU) return v * __builtin_amdgcn_rcpf(1.f + __expf(-v));
    else if constexpr (ACT == ACT_LEAKY) return v > 0.f ? v : 0.05f * v;
    else if constexpr (ACT == ACT_SIGMOID) return __builtin_amdgcn_rcpf(1.f + __expf(-v));
    else if constexpr (ACT == ACT_TANH) return tanhf(v);
    else return v;
}
// fp16 store of a wave's 64 x (32*NT) tile through LDS.  The MFMA accumulator layout gives a lane one column and 16
// rows, i.e. 2-byte global stores in 64-byte runs (measured: 116 us of a 557 us launch).  Instead: neighbouring lanes
// swap one value (DPP) so that every lane owns a column PAIR of one row, the pairs go to this wave's private LDS
// image [64 rows][32*NT] with ds_write_b32 (even lanes row r, odd lanes row r+1: different banks), and the image is
// read back linearly, 16 bytes per lane, and stored with global_store_dwordx4 (full 64*NT-byte row runs).
// `lds_wave` = this wave's 64*64*NT-byte region; the caller has made sure no wave still reads the tile buffers.
template <int NT>
__device__ __forceinline__ char* conv_stage_base(char* lds_wave, int lane) {
    return lds_wave + (4 * (lane >> 5) + (lane & 1)) * (NT * 64) + ((lane & 31) >> 1) * 4;
}

// one 32x32 accumulator tile (mi, ni), finished values v[r] in MFMA register order
template <int NT, int MI, int NI>
__device__ __forceinline__ void conv_stage_tile(const float (&v)[16], char* wbase, int lane) {
    constexpr int ROWB = NT * 64;                  // bytes per staged row
    const bool odd = (lane & 1) != 0;
    static_for<0, 8>([&](auto rp_) __attribute__((always_inline)) {
        constexpr int r0 = 2 * decltype(rp_)::value;
        const float v0 = v[r0], v1 = v[r0 + 1];                              // rows R and R+1 of this lane's column
        const float send = odd ? v0 : v1;
        const float recv = __builtin_bit_cast(
            float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, true));   // lane ^ 1
        const half2v h = {(_Float16)(odd ? recv : v0), (_Float16)(odd ? v1 : recv)};
        *reinterpret_cast<half2v*>(wbase + (MI * 32 + (r0 & 3) + 8 * (r0 >> 2)) * ROWB + NI * 64) = h;
    });
}

// after all tiles are staged: LDS operations of one wave execute in order, so the reads see the writes.
// Addresses: uniform 64-bit base + 32-bit lane offset (the launchers reject outputs of 4 GiB or more).
// FENCE_EVERY > 0: at most that many 16-byte pieces in flight (callers that still hold live accumulators)
template <int NT, int FENCE_EVERY = 0>
__device__ __forceinline__ void conv_stage_flush(const GemmArgs& a, const char* lds_wave, int m0, int n0, int wm, int wn,
                                                 int lane) {
    constexpr int NCH = NT * 4;                    // 16-byte pieces per staged row
    const uint32_t ldo2 = (uint32_t)a.ldo * 2u;
    char* out = reinterpret_cast<char*>(a.out) + ((size_t)(m0 + wm * 64) * a.ldo + n0 + wn * NT * 32) * 2;   // uniform
    const int rows_valid = a.Mvalid - (m0 + wm * 64);
    // image unit i = it*64 + lane = row*NCH + ch, advanced incrementally (64 = (64/NCH)*NCH + 64%NCH)
    int row = lane / NCH, ch = lane - row * NCH;
#pragma unroll
    for (int it = 0; it < NCH; ++it) {
        if (FENCE_EVERY > 0 && it % (FENCE_EVERY > 0 ? FENCE_EVERY : 1) == 0) asm volatile("" ::: "memory");
        const uint4 v = *reinterpret_cast<const uint4*>(lds_wave + (it * 64 + lane) * 16);
        if (row < rows_valid) *reinterpret_cast<uint4*>(out + ((uint32_t)row * ldo2 + (uint32_t)ch * 16u)) = v;
        ch += 64 % NCH; row += 64 / NCH;
        if (ch >= NCH) { ch -= NCH; row += 1; }
    }
}

// acc[mi][ni]: 32x32 MFMA accumulators of the wave's 64 x (32*NT) tile (board wm of the 4-board tile, N part wn).
// EPI 0: bias / activation ACT / scale, fp16 store, per-(board,channel) sum and sum of squares.
// EPI 1: GroupNorm(16 channels x 64 squares) + activation ACT in registers -- the wave owns whole groups.
// EPI 2: bias / runtime activation a.epi_act / gate multiply / scale, fp16 or f32 output stored per element
//        (small head GEMMs only; ACT ignored).
// EPI 0/1 finish each 32x32 tile's values, stage the tile in LDS at once (frees its accumulators), then flush.
template <int EPI, int ACT, int NT>
__device__ __forceinline__ void conv_tile_epilogue(float16v (&acc)[2][NT], const GemmArgs& a, char* lds_wave, int m0,
                                                   int n0, int wm, int wn, int lane) {
    const int half = lane >> 5;
    const int r31 = lane & 31;
    const int colbase = n0 + wn * NT * 32 + r31;
    char* wbase = conv_stage_base<NT>(lds_wave, lane);
    if constexpr (EPI == 1) {
        static_for<0, NT>([&](auto ni_) __attribute__((always_inline)) {
            constexpr int ni = decltype(ni_)::value;
            const int col = colbase + ni * 32;
            float s = 0.f, ss = 0.f;
            static_for<0, 2>([&](auto mi_) __attribute__((always_inline)) {
                const float16v av = acc[decltype(mi_)::value][ni];
                static_for<0, 16>([&](auto r_) __attribute__((always_inline)) { const float v = av[decltype(r_)::value]; s += v; ss += v * v; });
            });
#pragma unroll
            for (int o = 1; o <= 8; o <<= 1) { s += __shfl_xor(s, o); ss += __shfl_xor(ss, o); }
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
                conv_stage_tile<NT, mi, ni>(v, wbase, lane);
            });
        });
        conv_stage_flush<NT>(a, lds_wave, m0, n0, wm, wn, lane);
    }
    if constexpr (EPI == 0) {
        const float oscale = a.out_scale;
        const bool want_stats = a.out_stats != nullptr;
        float* stats = a.out_stats + ((size_t)(m0 / 64 + wm) * a.N + colbase) * 2;
        static_for<0, NT>([&](auto ni_) __attribute__((always_inline)) {
            constexpr int ni = decltype(ni_)::value;
            const float bias = a.bias != nullptr ? a.bias[colbase + ni * 32] : 0.f;
            float s = 0.f, ss = 0.f;
            static_for<0, 2>([&](auto mi_) __attribute__((always_inline)) {
                constexpr int mi = decltype(mi_)::value;
                float v[16];
                static_for<0, 16>([&](auto r_) __attribute__((always_inline)) {
                    constexpr int r = decltype(r_)::value;
                    v[r] = act_fast<ACT>(acc[mi][ni][r] + bias) * oscale;
                    s += v[r]; ss += v[r] * v[r];
                });
                conv_stage_tile<NT, mi, ni>(v, wbase, lane);
            });
            if (want_stats) {
                s += __shfl_xor(s, 32);
                ss += __shfl_xor(ss, 32);
                if (lane < 32) { stats[ni * 64] = s; stats[ni * 64 + 1] = ss; }
            }
        });
        conv_stage_flush<NT>(a, lds_wave, m0, n0, wm, wn, lane);
    }
    if constexpr (EPI == 2) {
        const int ldo = a.ldo;
        const int rowbase = m0 + wm * 64 + 4 * half;
        const int epi_act = a.epi_act;
        const float oscale = a.out_scale;
        const bool want_stats = a.out_stats != nullptr;
        const bool has_mul = a.mul != nullptr;
        const bool f32out = a.out_f32 != 0;
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
}
