// conv_sw_kernel: the 3x3 320->320 implicit-GEMM conv with ONE wave per SIMD (4 waves, 256 threads, up to 512 VGPRs per
// lane) instead of conv_pp_kernel's two ping-pong groups.  Same workgroup tile (256 rows = 4 boards x 320 channels), same
// LDS images (activation chunk double buffer, ring of four weight half-tiles, both filled by global_load_lds), same
// packed weights; what changes is who overlaps what:
//   * a wave owns 128 rows (2 boards) x 160 channels = 4x5 MFMA 32x32 tiles (320 accumulator registers), so a 16-deep k
//     slice is 20 MFMAs fed by 9 fragment reads (4 activation + 5 weight) -- 0.45 reads per MFMA against 0.7;
//   * the fragment reads of slice s+1 and the DMA of half-tile y+4 are issued BETWEEN the MFMAs of slice s by the same wave
//     (software pipeline, two fragment register sets); there is no hand-over between waves and ONE s_barrier per
//     half-tile (40 MFMAs), which only orders the LDS ring.
// Ring protocol (y = half-tile, slices (y,0), (y,1); every wave issues 6 DMA pieces per half-tile: 5 weight pieces and one
// piece of the next chunk's activations or a filler):
//   B_y = [s_waitcnt vmcnt(12), lgkmcnt(0); s_barrier] at the start of slice (y,1)
//   RAW  half-tile y+1 is first read by the reads of (y+1,0), issued after B_y; its pieces were issued after B_{y-3}, and at
//        B_y only the two youngest groups (y+2, y+3) may still be in flight: vmcnt(12)
//   WAR  the pieces of y+4 overwrite the slot of y and are issued after B_y, before which every wave has retired its reads
//        of (y,0) and (y,1) (lgkmcnt(0))
#include "../../matrix0_amd/csrc/kernel_common.h"
#include "../../matrix0_amd/csrc/conv_epilogue.h"

__device__ __forceinline__ void sw_glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// inline asm lives in plain functions: clang does not accept asm operands that are captured variables of a generic lambda
template <int OFF>
__device__ __forceinline__ void sw_dsread(half8& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <bool AGPR>
__device__ __forceinline__ void sw_mfma(float16v& c, const half8& x, const half8& y) {
    if constexpr (AGPR) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(x), "v"(y));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(x), "v"(y));
}

#ifdef SW_STAMP
__device__ unsigned long long* g_sw_stamp;      // [blocks][4]: s_memtime / s_memrealtime at main-loop start and end
#endif

template <int EPI, int ACT>
__global__ __launch_bounds__(256) void conv_sw_kernel(GemmArgs a) {
    constexpr int NT = 5, MT = 4;
    constexpr int A_BYTES = 256 * 128;    // 4 boards x 64 squares x 64 channels fp16, 128-byte rows
    constexpr int WH_BYTES = 320 * 64;    // 320 output channels x 32 k fp16, 64-byte rows
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* A_lds = smem;                   // [2][A_BYTES]
    char* W_lds = smem + 2 * A_BYTES;     // [4][WH_BYTES]
    char* Z_lds = W_lds + 4 * WH_BYTES;   // one all-zero square (128 B)
    char* D_lds = Z_lds + 128;            // [4][1024] sink of the filler DMA pieces

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1;              // boards 2wm, 2wm+1 of the tile
    const int wn = wave >> 1;             // channel half
    const int m0 = blockIdx.x * 256;
    const int Cin = a.Cin;
    const int nchunk = Cin >> 6;
    const int NH = nchunk * 18;           // half-tiles
    const int half = lane >> 5;
    const int r31 = lane & 31;

    if (tid < 8) reinterpret_cast<uint4*>(Z_lds)[tid] = make_uint4(0, 0, 0, 0);

    const char* in_bytes = reinterpret_cast<const char*>(a.in);
    const size_t w_kt_stride = (size_t)gridDim.y * (2 * WH_BYTES);
    const char* w_base = reinterpret_cast<const char*>(a.w) + (size_t)blockIdx.y * (2 * WH_BYTES);

    // ---- prologue: chunk 0 activations (32 pieces, 8 per wave), half-tiles 0..3 (20 pieces each, 5 per wave) ----
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = wave * 8 + i;
        const int p = 8 * q + (lane >> 3);
        const int cl = lane & 7;
        sw_glds16(in_bytes + ((size_t)(m0 + p) * Cin) * 2 + 16 * (cl ^ ((p >> 1) & 7)), A_lds + q * 1024);
    }
#pragma unroll
    for (int y0 = 0; y0 < 4; ++y0) {
        const char* src = w_base + (size_t)(y0 >> 1) * w_kt_stride + (size_t)(y0 & 1) * WH_BYTES;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int pc = wave * 5 + i;
            sw_glds16(src + pc * 1024 + lane * 16, W_lds + y0 * WH_BYTES + pc * 1024);
        }
    }

    // ---- steady-state DMA state (wave-uniform, advanced incrementally) ----
    const uint32_t w_lane = (uint32_t)lane * 16u;
    int t_next = 4;                                                        // half-tile the next issue group fetches
    const char* w_ptr = w_base + (size_t)2 * w_kt_stride + (size_t)wave * 5 * 1024;     // half-tile 4 = K-tile 2, half 0
    int w_slot = 0;                                                        // LDS offset of slot t_next & 3
    // next chunk's activations: this wave's 8 pieces q = 8 wave + i; lane part of the source address:
    const uint32_t a_lane = (uint32_t)(lane >> 3) * (uint32_t)Cin * 2u;   // + 16 * (cl ^ key(p)) added per piece (key varies with q)
    const char* a_ptr = in_bytes;
    int a_dst = 0, a_left = 0, a_q = 0;
    auto issue_group = [&]() __attribute__((always_inline)) {
        char* dst = W_lds + w_slot + wave * 5 * 1024;
#pragma unroll
        for (int i = 0; i < 5; ++i) sw_glds16(w_ptr + i * 1024 + w_lane, dst + i * 1024);
        {
            const bool have = a_left > 0;
            // piece q covers rows 8q..8q+7; row p = 8q + (lane>>3): key = (p >> 1) & 7 = ((4q) + (lane >> 4)) & 7
            const uint32_t key = (uint32_t)((4 * a_q + (lane >> 4)) & 7);
            const uint32_t lo = a_lane + 16u * ((uint32_t)(lane & 7) ^ key);
            sw_glds16((have ? a_ptr : in_bytes) + lo, have ? A_lds + a_dst : D_lds + wave * 1024);
            a_ptr += have ? (size_t)8 * Cin * 2 : 0;
            a_dst += have ? 1024 : 0;
            a_q += have ? 1 : 0;
            a_left -= have ? 1 : 0;
        }
        const bool more = t_next + 1 < NH;
        const size_t inc = (t_next & 1) ? (w_kt_stride - WH_BYTES) : (size_t)WH_BYTES;
        w_ptr += more ? inc : 0;
        t_next += 1;
        w_slot = (t_next & 3) * WH_BYTES;
    };

    // Accumulators: 20 tiles x 16 registers = 320 per lane, more than the 256 AGPRs: tiles (mi, ni) with mi < 3, or mi == 3 and
    // ni == 0, live in AGPRs (16 tiles), the last four in VGPRs.  The MFMAs are inline asm with the register class in the
    // constraint -- left to itself hipcc keeps rotating accumulators between the two files (2 080 v_accvgpr moves per K-tile).
    float16v acc[MT][NT];
    static_for<0, MT>([&](auto mi) __attribute__((always_inline)) {
        static_for<0, NT>([&](auto ni) __attribute__((always_inline)) {
            acc[decltype(mi)::value][decltype(ni)::value] = float16v{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f,
                                                                      0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        });
    });
#define SW_IN_AGPR(mi, ni) ((mi) * NT + (ni) < 16)

    // per-lane constants of the fragment reads (LDS byte offsets: the dynamic segment starts at LDS address 0)
    const uint32_t A_off = 0, W_off = 2 * A_BYTES, Z_off = 2 * A_BYTES + 4 * WH_BYTES;
    const int wfx = ((r31 >> 2) & 3) ^ half;
    const uint32_t wrow_off = (uint32_t)((wn * NT * 32 + r31) * 64);
    int prow[MT], py[MT], px[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        prow[mi] = wm * 128 + mi * 32 + r31;
        py[mi] = (prow[mi] >> 3) & 7;
        px[mi] = prow[mi] & 7;
    }
    // fragment addresses of tap `tp` in activation buffer at LDS offset `Abuf` (zero square outside the board)
    auto tap_addr = [&](uint32_t Abuf, int tp, uint32_t (&ab)[MT], int (&af)[MT]) __attribute__((always_inline)) {
        const int t3 = tp / 3;
        const int dy = t3 - 1, dx = tp - t3 * 3 - 1;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            const int yy = py[mi] + dy, xx = px[mi] + dx;
            const bool ok = (unsigned)yy < 8u && (unsigned)xx < 8u;
            const int pp = prow[mi] + dy * 8 + dx;
            ab[mi] = ok ? Abuf + (uint32_t)pp * 128u : Z_off;
            af[mi] = ok ? (((pp >> 1) & 7) ^ half) : 0;
        }
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    half8 fa[2][MT], fb[2][NT];
    uint32_t abase[MT];
    int afx[MT];
    tap_addr(A_off, 0, abase, afx);
#define SW_DSREAD(dst, addr, off) sw_dsread<(off)>(dst, addr)
    {   // slice (0,0) into set 0
        const uint32_t Wb = W_off + wrow_off + 16u * (uint32_t)wfx;
        static_for<0, MT>([&](auto mi_) __attribute__((always_inline)) {
            constexpr int mi = decltype(mi_)::value;
            const uint32_t ad = abase[mi] + 16u * (uint32_t)afx[mi];
            SW_DSREAD(fa[0][mi], ad, 0);
        });
        static_for<0, NT>([&](auto ni_) __attribute__((always_inline)) {
            constexpr int ni = decltype(ni_)::value;
            SW_DSREAD(fb[0][ni], Wb, ni * 2048);
        });
    }

#ifdef SW_STAMP
    const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    // One K-tile (tap) per iteration = 4 slices j; slice j computes on set j&1 while the reads of slice j+1 go to the other.
    int y = 0;
#pragma unroll 1
    for (int c = 0; c < nchunk; ++c) {
        const uint32_t Ab = A_off + (uint32_t)(c & 1) * A_BYTES;
        a_left = c + 1 < nchunk ? 8 : 0;
        a_q = wave * 8;
        a_ptr = in_bytes + ((size_t)(m0 + 64 * wave) * Cin + (size_t)(c + 1) * 64) * 2;
        a_dst = ((c + 1) & 1) * A_BYTES + wave * 8 * 1024;
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            static_for<0, 4>([&](auto j_) __attribute__((always_inline)) {
                constexpr int j = decltype(j_)::value;
                constexpr int set = j & 1;
                constexpr int nset = (j + 1) & 1;
                if constexpr ((j & 1) == 1) {
                    // B_{y+h}: ring hand-over (see the header)
                    asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                // next slice: (tap, j+1), or slice 0 of the next tap (new fragment addresses first)
                if constexpr (j == 3) {
                    const int ntap = tap == 8 ? 0 : tap + 1;
                    const uint32_t nAb = tap == 8 ? A_off + (uint32_t)((c + 1) & 1) * A_BYTES : Ab;
                    tap_addr(nAb, ntap, abase, afx);
                }
                constexpr int nj = (j + 1) & 3;
                const int nyh = y + (j == 3 ? 2 : ((j + 1) >> 1));
                const uint32_t nWb = W_off + (uint32_t)(nyh & 3) * WH_BYTES + wrow_off + 16u * (uint32_t)(wfx ^ ((nj & 1) << 1));
                uint32_t nad[MT];
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) nad[mi] = abase[mi] + 16u * (uint32_t)(afx[mi] ^ (nj << 1));
                __builtin_amdgcn_sched_barrier(0);
                // the 20 MFMAs of this slice with the next slice's 9 reads and (odd slices) the 6 DMA pieces between them
                static_for<0, MT>([&](auto mi_) __attribute__((always_inline)) {
                    constexpr int mi = decltype(mi_)::value;
                    static_for<0, NT>([&](auto ni_) __attribute__((always_inline)) {
                        constexpr int ni = decltype(ni_)::value;
                        constexpr int i = mi * NT + ni;
                        sw_mfma<SW_IN_AGPR(mi, ni)>(acc[mi][ni], fa[set][mi], fb[set][ni]);
                        if constexpr (i < MT) {
                            SW_DSREAD(fa[nset][i], nad[i], 0);
                        } else if constexpr (i < MT + NT) {
                            SW_DSREAD(fb[nset][i - MT], nWb, (i - MT) * 2048);
                        } else if constexpr (i == 12 && (j & 1) == 1) {
                            issue_group();
                        }
                    });
                });
                __builtin_amdgcn_sched_barrier(0);
            });
            y += 2;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef SW_STAMP
    if (tid == 0) {
        unsigned long long* o = g_sw_stamp + (size_t)blockIdx.x * 4;
        o[0] = st_c0; o[1] = st_r0; o[2] = __builtin_amdgcn_s_memtime(); o[3] = __builtin_amdgcn_s_memrealtime();
    }
#endif

#ifdef SW_NO_EPILOGUE
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) asm volatile("" :: "v"(acc[mi][ni]));
#else
    // the two boards of this wave through the 64-row epilogue of the 8-wave kernel, one after the other
    {
        const int n0 = blockIdx.y * 320;
        float16v (&lo)[2][NT] = *reinterpret_cast<float16v (*)[2][NT]>(&acc[0]);
        float16v (&hi)[2][NT] = *reinterpret_cast<float16v (*)[2][NT]>(&acc[2]);
        conv_tile_epilogue<EPI, ACT, NT>(lo, a, smem + (wave * 2 + 0) * (NT * 64 * 64), m0, n0, 2 * wm + 0, wn, lane);
        conv_tile_epilogue<EPI, ACT, NT>(hi, a, smem + (wave * 2 + 1) * (NT * 64 * 64), m0, n0, 2 * wm + 1, wn, lane);
    }
#endif
}

template <int EPI, int ACT>
static hipError_t launch_conv_sw_e(const GemmArgs& a, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_sw_kernel<EPI, ACT>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    dim3 grid(a.Mrows / 256, a.Npad / 320);
    hipLaunchKernelGGL((conv_sw_kernel<EPI, ACT>), grid, dim3(256), 160 * 1024, st, a);
    return hipGetLastError();
}

hipError_t launch_conv_sw(const GemmArgs& a, hipStream_t st) {
    if (a.Cin % 64 != 0 || a.Npad % 320 != 0 || a.Mrows % 256 != 0 || a.Cin < 128) return hipErrorInvalidValue;
    if (a.res != nullptr || a.mul != nullptr || a.out_f32 != 0) return hipErrorInvalidValue;
    if (a.gn_gamma != nullptr) {
        if (a.epi_act == ACT_SILU) return launch_conv_sw_e<1, ACT_SILU>(a, st);
        if (a.epi_act == ACT_RELU) return launch_conv_sw_e<1, ACT_RELU>(a, st);
        return hipErrorInvalidValue;
    }
    return a.epi_act == ACT_NONE ? launch_conv_sw_e<0, ACT_NONE>(a, st) : hipErrorInvalidValue;
}
