// Residual-block tail fused into conv2's epilogue (conv_pp_kernel, EPI 3):
//   t      = conv2 output (this workgroup holds 4 whole boards x all 320 channels in its accumulators)
//   gate   = sigmoid(W2 act(W1 mean_squares(t) + b1) + b2)            squeeze-excite, resnet.py:59-68 (optional)
//   y      = x + gate * t                                             the residual stream          -> a.out
//   y2     = act(GroupNorm16(y; next block's bn1))                    the next conv1's operand     -> a.y2 (optional)
// i.e. what conv2's plain epilogue + se_gate_kernel + ew_board_kernel did with three launches and two more trips of
// the tensor through HBM.  Everything a board needs is inside the workgroup, so the only new global traffic is the
// read of x.  The phases, all 8 waves together (LDS is free once the main loop is over):
//   A  per-board channel means of t from the accumulators                         -> LDS pool[4][C]
//   B  W1 (C x Hd f32) by global_load_lds -> LDS, hidden = act(W1 pool + b1);  W2 likewise, gate -> LDS; each lane
//      picks up the 5 gate values of its accumulator columns
//   C  gate * t staged as fp16 in the wave's private LDS image [64 rows][160 ch]   (conv_stage_tile)
//   D  lane = (16-byte channel chunk, row mod 3): add x (global, 16-byte loads issued up front), store y, write y back
//      to the image, per-channel sums -> GroupNorm statistics of the wave's 10 groups by shuffles
//   E  second pass over the image: y2 = act(y * scale + shift), 16-byte stores
// Phases C-E touch only the wave's own image: no workgroup barrier after B.
//
// (The attention block's tail -- residual + LayerNorm over C + next GroupNorm -- was fused into the proj conv the same
//  way, in the accumulator layout with DPP row reductions: correct, but 374 us against 92 + 228 us for proj +
//  ew_board_kernel, because a row's LayerNorm statistics need cross-lane and cross-wave reductions and a per-element
//  gather of x; not kept.)
// >>> conv_tail16.h: the same tail for conv_pp16_kernel's accumulator layout (4 x 10 tiles of 16x16, conv_epilogue16.h):
// phases A (channel means), B5 (gate values of the lane's channels), PRE and C (staging) index the accumulators; B, D, E
// work on LDS images and are unchanged.
#pragma once
#include "conv_epilogue16.h"

__device__ __forceinline__ void tail16_glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// C == 320 (one N block), 8 waves: wave = (board wm, channel half wn), NT == 5.
// PRE: t is first normalised and activated, t <- act(GroupNorm16(t; a.pre_gamma, a.pre_beta)) -- the chess-feature
// convs (resnet.py:229-244: x += act(norm(conv(x)))); no squeeze-excite in that case.
#ifdef SW_STAMP
__device__ unsigned long long* g_tail_stamp;    // [blocks][8] realtime stamps of the tail's phases (tools/ubench)
#define TAIL_STAMP(k) do { if (tid == 0) g_tail_stamp[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TAIL_STAMP(k) do {} while (0)
#endif

template <int ACT, bool PRE = false>
__device__ __forceinline__ void conv_tail_epilogue16(float4v (&acc)[4][10], const GemmArgs& a, char* smem, int m0, int wm,
                                                     int wn, int wave, int lane) {
    constexpr int NT = 5, C = 320, NG = 10;
    const int tid = wave * 64 + lane;
    const int c15 = lane & 15;
    float* pool = reinterpret_cast<float*>(smem);              // [C][4 boards]   5120 B (one 16-byte read per channel)
    float* gate = pool + 4 * C;                                // [4][C]          5120 B
    float* hid = gate + 4 * C;                                 // [Hd<=128][4]    2048 B
    float* part = hid + 4 * 128;                               // [parts<=8][4][Hd<=128] 16384 B
    float* wst = reinterpret_cast<float*>(smem + 32768);       // staged W1 / W2: C*Hd*4 <= 131072 B - 32768
    const int Hd = a.se_hidden;
    const bool se = a.se_w1 != nullptr;
    // phase D/E lane mapping and the second output's GroupNorm parameters (fetched now: a late load is an exposed
    // global-memory latency in a kernel with one workgroup per CU)
    constexpr int NCH = NT * 4;                                  // 20 chunks per 160-channel row
    constexpr int NIT = 22;                                      // ceil(64 / 3)
    const int chunk = lane % NCH, rsub = lane / NCH;
    const bool lane_on = rsub < 3;
    float gg[8], bb[8];
    if (a.y2 != nullptr) {
        const int c0 = wn * 160 + chunk * 8;
        const float4 g0 = *reinterpret_cast<const float4*>(a.gn_gamma + c0), g1 = *reinterpret_cast<const float4*>(a.gn_gamma + c0 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(a.gn_beta + c0), b1 = *reinterpret_cast<const float4*>(a.gn_beta + c0 + 4);
        gg[0] = g0.x; gg[1] = g0.y; gg[2] = g0.z; gg[3] = g0.w; gg[4] = g1.x; gg[5] = g1.y; gg[6] = g1.z; gg[7] = g1.w;
        bb[0] = b0.x; bb[1] = b0.y; bb[2] = b0.z; bb[3] = b0.w; bb[4] = b1.x; bb[5] = b1.y; bb[6] = b1.z; bb[7] = b1.w;
    }
    float gv[NG];
    TAIL_STAMP(0);
#ifdef TAIL_NO_SE
    if (false) {
#else
    if (se && !PRE) {
#endif
        // B1 (issued first, lands while phase A runs): W1 and W2 as fp16 -> LDS, both at once (C*Hd*2 bytes each, a
        // multiple of 1024 for C = 320 and Hd a multiple of 8, which the launcher checks)
        const int wbytes = C * Hd * 2;
        const int npieces = wbytes >> 10;
        _Float16* wsth = reinterpret_cast<_Float16*>(wst);
        for (int p = wave; p < npieces; p += 8)
            tail16_glds16(reinterpret_cast<const char*>(a.se_w1h) + p * 1024 + lane * 16, reinterpret_cast<char*>(wsth) + p * 1024);
        for (int p = wave; p < npieces; p += 8)
            tail16_glds16(reinterpret_cast<const char*>(a.se_w2h) + p * 1024 + lane * 16, reinterpret_cast<char*>(wsth) + wbytes + p * 1024);
        const _Float16* w1l = wsth;                             // [C][Hd]
        const _Float16* w2l = wsth + C * Hd;                    // [Hd][C]
        // A: channel means of this wave's board / channel half
        static_for<0, NG>([&](auto ni_) __attribute__((always_inline)) {
            constexpr int ni = decltype(ni_)::value;
            float s = 0.f;
            static_for<0, 4>([&](auto mi_) __attribute__((always_inline)) {
                static_for<0, 4>([&](auto r_) __attribute__((always_inline)) { s += acc[decltype(mi_)::value][ni][decltype(r_)::value]; });
            });
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (lane < 16) pool[(wn * 160 + ni * 16 + c15) * 4 + wm] = s * (1.f / 64.f);
        });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // B2: hidden = act(W1 pool + b1); thread = (hidden unit j, channel part), the 4 boards at once; w1 is [C][Hd]
        const int parts = 512 / Hd > 8 ? 8 : 512 / Hd;
        {
            const int j = tid % Hd, p = tid / Hd;
            if (p < parts) {
                const int cpp = (C + parts - 1) / parts;
                const int cbeg = p * cpp, cend = cbeg + cpp < C ? cbeg + cpp : C;
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 6
                for (int c = cbeg; c < cend; ++c) {
                    const float w = (float)w1l[c * Hd + j];
                    const float4 pv = *reinterpret_cast<const float4*>(pool + c * 4);
                    s0 = __builtin_fmaf(w, pv.x, s0); s1 = __builtin_fmaf(w, pv.y, s1); s2 = __builtin_fmaf(w, pv.z, s2); s3 = __builtin_fmaf(w, pv.w, s3);   // explicit: all four boards alike (se_gate_kernel's header)
                }
                float* pp = part + (p * 4) * 128 + j;
                pp[0] = s0; pp[128] = s1; pp[256] = s2; pp[384] = s3;
            }
        }
        __syncthreads();
        // B3: the hidden units
        for (int i = tid; i < 4 * Hd; i += 512) {
            const int b = i / Hd, j = i - b * Hd;
            float s = a.se_b1[j];
            for (int p = 0; p < parts; ++p) s += part[(p * 4 + b) * 128 + j];
            hid[j * 4 + b] = act_fast<ACT>(s);
        }
        __syncthreads();
        // B4: gate = sigmoid(W2 hidden + b2); thread = channel (first 320 threads), the 4 boards at once; w2 is [Hd][C]
        if (tid < C) {
            const float b2 = a.se_b2[tid];
            float s0 = b2, s1 = b2, s2 = b2, s3 = b2;
#pragma unroll 8
            for (int j = 0; j < Hd; ++j) {
                const float w = (float)w2l[j * C + tid];
                const float4 hv = *reinterpret_cast<const float4*>(hid + j * 4);
                s0 = __builtin_fmaf(w, hv.x, s0); s1 = __builtin_fmaf(w, hv.y, s1); s2 = __builtin_fmaf(w, hv.z, s2); s3 = __builtin_fmaf(w, hv.w, s3);
            }
            gate[tid] = __builtin_amdgcn_rcpf(1.f + __expf(-s0));
            gate[C + tid] = __builtin_amdgcn_rcpf(1.f + __expf(-s1));
            gate[2 * C + tid] = __builtin_amdgcn_rcpf(1.f + __expf(-s2));
            gate[3 * C + tid] = __builtin_amdgcn_rcpf(1.f + __expf(-s3));
        }
        __syncthreads();
        // B5: the gate values of this lane's accumulator columns
        static_for<0, NG>([&](auto ni_) __attribute__((always_inline)) {
            constexpr int ni = decltype(ni_)::value;
            gv[ni] = gate[wm * C + wn * 160 + ni * 16 + c15];
        });
        __syncthreads();                                          // the images below overwrite pool / gate / wst
    } else {
        static_for<0, NG>([&](auto ni_) __attribute__((always_inline)) { gv[decltype(ni_)::value] = 1.f; });
    }
    float pv[NG];                                                 // PRE: per-column shift (gv = scale)
    if constexpr (PRE) {
        // GroupNorm(16 channels x 64 squares) of t on the accumulators: the wave owns whole groups (as EPI 1)
        static_for<0, NG>([&](auto ni_) __attribute__((always_inline)) {
            constexpr int ni = decltype(ni_)::value;
            const int col = wn * 160 + ni * 16 + c15;
            float s = 0.f, ss = 0.f;
            static_for<0, 4>([&](auto mi_) __attribute__((always_inline)) {
                const float4v av = acc[decltype(mi_)::value][ni];
                static_for<0, 4>([&](auto r_) __attribute__((always_inline)) { const float v = av[decltype(r_)::value]; s += v; ss += v * v; });
            });
            s = wave_sum64(s); ss = wave_sum64(ss);
            const float mean = s * (1.f / 1024.f);
            float var = ss * (1.f / 1024.f) - mean * mean;
            var = var > 0.f ? var : 0.f;
            gv[ni] = rsqrtf(var + 1e-5f) * a.pre_gamma[col];
            pv[ni] = a.pre_beta[col] - mean * gv[ni];
        });
    }

    TAIL_STAMP(1);
    // C: gate * t -> the wave's fp16 image; the loads of x are issued between the tile columns, into the registers
    // the staged accumulators free (x is 2-3 us away and nothing else runs on this CU)
    char* img = smem + wave * (NT * 64 * 64);
    char* wbase = conv_stage_base16(img, lane);
    const uint32_t ldo2 = (uint32_t)a.ldo * 2u;
    const size_t tile_off = ((size_t)(m0 + wm * 64) * a.ldo + wn * 160) * 2;      // wave-uniform
    const char* xin = reinterpret_cast<const char*>(a.res) + tile_off;
    char* yout = reinterpret_cast<char*>(a.out) + tile_off;
    const int rows_valid = a.Mvalid - (m0 + wm * 64);
    const uint32_t lane_goff = (uint32_t)rsub * ldo2 + (uint32_t)chunk * 16u;
    const uint32_t lane_loff = (uint32_t)(rsub * NCH + chunk) * 16u;
    half8 xv[NIT];
    static_for<0, NG>([&](auto ni_) __attribute__((always_inline)) {
        constexpr int ni = decltype(ni_)::value;
        static_for<0, 4>([&](auto mi_) __attribute__((always_inline)) {
            constexpr int mi = decltype(mi_)::value;
            float v[4];
            static_for<0, 4>([&](auto r_) __attribute__((always_inline)) {
                constexpr int r = decltype(r_)::value;
                if constexpr (PRE) v[r] = act_fast<ACT>(acc[mi][ni][r] * gv[ni] + pv[ni]);
                else v[r] = acc[mi][ni][r] * gv[ni];
            });
            conv_stage_tile16<mi, ni>(v, wbase, lane);
        });
        __builtin_amdgcn_sched_barrier(0);
        // the 22 loads of x spread over the 10 channel tiles (3 after each of the first two, 2 after the others)
        static_for<(ni < 2 ? ni * 3 : 6 + (ni - 2) * 2), (ni < 2 ? ni * 3 + 3 : 6 + (ni - 1) * 2)>([&](auto it_) __attribute__((always_inline)) {
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

    TAIL_STAMP(2);
    // D: y = x + image; lane = (chunk of 8 channels, row mod 3), rows rsub, rsub+3, ...; lanes 60..63 idle
    // The sum of two fp16 numbers rounded to fp16 is what the fp32 add + conversion gives, so y is computed with packed
    // fp16 adds (4 instructions per 8 channels); the GroupNorm sums (this lane's 8 channels x its rows) use the
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
    TAIL_STAMP(3);
#ifdef TAIL_NO_Y2
    return;
#endif
    if (a.y2 == nullptr) return;

    // GroupNorm statistics of y: over the 3 row classes (lanes chunk, chunk+20, chunk+40), then over the group's 16
    // channels = this lane's 8 + the neighbour chunk's 8
    {
        const float s1 = __shfl(gs, chunk + NCH), s2 = __shfl(gs, chunk + 2 * NCH);
        const float q1 = __shfl(gss, chunk + NCH), q2 = __shfl(gss, chunk + 2 * NCH);
        gs = __shfl(gs, chunk) + s1 + s2;                        // every lane: totals of its chunk (same order everywhere)
        gss = __shfl(gss, chunk) + q1 + q2;
        const float so = __shfl_xor(gs, 1), qo = __shfl_xor(gss, 1);      // partner chunk (chunk ^ 1 is lane ^ 1 for lanes < 60)
        const float lo_s = (chunk & 1) ? so : gs, hi_s = (chunk & 1) ? gs : so;
        const float lo_q = (chunk & 1) ? qo : gss, hi_q = (chunk & 1) ? gss : qo;
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
    // E: y2 = act(GroupNorm(y)) from the image
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
    TAIL_STAMP(4);
}
