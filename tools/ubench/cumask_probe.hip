// Which CUs does a hipExtStreamCreateWithCUMask stream run on?  Prints, per mask preset, the set of (XCC id, SE id, CU id)
// the workgroups of a 4096-block launch landed on.   hipcc -O2 --offload-arch=gfx950 cumask_probe.hip -o cumask_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <set>
#include <map>
#include <tuple>

__global__ void probe(uint32_t* out) {
    if (threadIdx.x == 0) {
        uint32_t xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid;
    }
    // keep the CU busy a little so that blocks spread over all enabled CUs
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 20000) {}
}

int main() {
    const int NB = 4096;
    uint32_t* d; hipMalloc(&d, NB * 8);
    struct Preset { const char* name; uint32_t w[8]; };
    std::vector<Preset> ps;
    ps.push_back({"all", {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}});
    ps.push_back({"low128", {~0u, ~0u, ~0u, ~0u, 0, 0, 0, 0}});
    ps.push_back({"high128", {0, 0, 0, 0, ~0u, ~0u, ~0u, ~0u}});
    ps.push_back({"bits b%8<4", {0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu}});
    ps.push_back({"bits b%8>=4", {0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u}});
    ps.push_back({"even bits", {0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u}});
    ps.push_back({"first 32", {~0u, 0, 0, 0, 0, 0, 0, 0}});
    for (auto& p : ps) {
        hipStream_t st;
        hipError_t e = hipExtStreamCreateWithCUMask(&st, 8, p.w);
        if (e != hipSuccess) { printf("%s: create failed %s\n", p.name, hipGetErrorString(e)); continue; }
        hipMemsetAsync(d, 0xff, NB * 8, st);
        hipLaunchKernelGGL(probe, dim3(NB), dim3(64), 0, st, d);
        e = hipStreamSynchronize(st);
        std::vector<uint32_t> h(NB * 2);
        hipMemcpy(h.data(), d, NB * 8, hipMemcpyDeviceToHost);
        std::map<int, std::set<std::pair<int, int>>> per;     // xcc -> (se, cu)
        for (int b = 0; b < NB; ++b) {
            const uint32_t xcc = h[2 * b] & 0xf, hw = h[2 * b + 1];
            const int cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
            per[xcc].insert({se * 2 + sh, cu});
        }
        int tot = 0;
        printf("%-12s (%s):", p.name, hipGetErrorString(e));
        for (auto& kv : per) { printf(" xcc%d:%zu", kv.first, kv.second.size()); tot += (int)kv.second.size(); }
        printf("  total %d CUs\n", tot);
        if (p.w[1] == 0 && p.w[0] == ~0u) { for (auto& kv : per) { printf("   xcc%d:", kv.first); for (auto& sc : kv.second) printf(" (%d,%d)", sc.first, sc.second); printf("\n"); } }
        hipStreamDestroy(st);
    }
    return 0;
}
