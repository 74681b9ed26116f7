// Shared device helpers of the network kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "net_kernels.h"

#include <stdlib.h>
#include <type_traits>
#include <utility>

// Host side: run `f` once per HIP device (hipFuncSetAttribute is per device; a process may hold networks on several GPUs,
// and SelfplayPool threads launch concurrently).  f returns hipError_t; a failure is retried on the next call.
#include <atomic>
struct DeviceOnce {
    std::atomic<uint32_t> done{0};          // bit d = device d configured (devices >= 32 run f every time)
    template <typename F>
    hipError_t run(F&& f) {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess) return hipErrorInvalidDevice;
        if ((unsigned)d < 32u && (done.load(std::memory_order_acquire) >> d) & 1u) return hipSuccess;
        const hipError_t e = f();
        if (e == hipSuccess && (unsigned)d < 32u) done.fetch_or(1u << d, std::memory_order_release);
        return e;
    }
};

// RULE for every kernel that stages through LDS-DMA (__builtin_amdgcn_global_load_lds: conv_zs / conv_pp16 / attn_block /
// conv_big): a COUNTED `s_waitcnt vmcnt(N)` (N > 0) may only be used while nothing but LDS-DMA operations of this wave is
// outstanding.  vmcnt counts loads into registers, stores and LDS-DMA together, but a later LDS-DMA can retire before an
// earlier load into registers (measured on gfx950, round 3: `vmcnt(1)` meant as "the bias load, not the DMA after it" let
// waves read their bias registers early; 17-89 of 24 576 boards differed from run to run).  So: issue register loads after
// the last counted wait of a sequence and wait for them with vmcnt(0), or keep them out of the DMA window altogether.
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

// Compile-time loop: the body sees its index as an integral_constant, so every accumulator index is a
// constant in the AST (a runtime- or late-unrolled index keeps the MFMA accumulators in scratch memory).
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case ACT_RELU: return v > 0.f ? v : 0.f;
        case ACT_SILU: return v / (1.f + __expf(-v));
        case ACT_LEAKY: return v > 0.f ? v : 0.05f * v;
        case ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
        case ACT_TANH: return tanhf(v);
        default: return v;
    }
}

