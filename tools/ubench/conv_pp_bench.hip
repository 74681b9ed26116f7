// Stand-alone timing / phase trace of conv_pp_kernel (3x3, 320 -> 320 channels) on synthetic data.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DPP_TRACE -DPP_TRACE_KT0=20 -DPP_TRACE_BLOCK=300] \
//         tools/ubench/conv_pp_bench.hip -o conv_pp_bench && ./conv_pp_bench [boards] [iters]
// With -DPP_TRACE it prints, for waves 0 and 4 of one workgroup and 4 consecutive K-tiles, the cycle stamps
// L-start / L-end(before barrier) / C-start(after barrier) / C-end(MFMAs issued) of each phase.
#include "../../matrix0_amd/csrc/conv_pp.hip"
#include <stdio.h>
#include <vector>

int main(int argc, char** argv) {
    const int boards = argc > 1 ? atoi(argv[1]) : 4096;
    const int iters = argc > 2 ? atoi(argv[2]) : 20;
    const int M = boards * 64, C = 320;
    std::vector<_Float16> hin((size_t)M * C), hw((size_t)9 * C * C);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((s >> 11) & 0xffff) / 65536.f - 0.5f; };
    for (auto& v : hin) v = (_Float16)rnd();
    for (auto& v : hw) v = (_Float16)(rnd() * 0.05f);
    _Float16 *din, *dw, *dout; float* dstats;
    hipMalloc(&din, hin.size() * 2); hipMalloc(&dw, hw.size() * 2); hipMalloc(&dout, (size_t)M * C * 2);
    hipMalloc(&dstats, (size_t)boards * C * 8);
    hipMemcpy(din, hin.data(), hin.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    GemmArgs a{};
    a.in = din; a.w = dw; a.out = dout; a.out_stats = dstats; a.Mrows = M; a.Mvalid = M; a.Cin = C; a.N = C; a.Npad = C;
    a.ldo = C; a.out_scale = 1.f; a.w_pp = 1;
#ifdef BENCH_TAIL   // conv2 with the residual-block tail fused (EPI 3): x, squeeze-excite weights, next GroupNorm, y2
    {
        const int Hd = 80;
        std::vector<float> w1((size_t)C * Hd), w2((size_t)Hd * C), b1(Hd, 0.01f), b2(C, 0.02f), gam(C, 1.f), bet(C, 0.f);
        for (auto& v : w1) v = rnd() * 0.1f;
        for (auto& v : w2) v = rnd() * 0.1f;
        float *dw1, *dw2, *db1, *db2, *dg, *dbt; _Float16 *dres, *dy2;
        hipMalloc(&dw1, w1.size() * 4); hipMalloc(&dw2, w2.size() * 4); hipMalloc(&db1, Hd * 4); hipMalloc(&db2, C * 4);
        hipMalloc(&dg, C * 4); hipMalloc(&dbt, C * 4); hipMalloc(&dres, (size_t)M * C * 2); hipMalloc(&dy2, (size_t)M * C * 2);
        hipMemcpy(dw1, w1.data(), w1.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dw2, w2.data(), w2.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(db1, b1.data(), Hd * 4, hipMemcpyHostToDevice); hipMemcpy(db2, b2.data(), C * 4, hipMemcpyHostToDevice);
        hipMemcpy(dg, gam.data(), C * 4, hipMemcpyHostToDevice); hipMemcpy(dbt, bet.data(), C * 4, hipMemcpyHostToDevice);
        hipMemcpy(dres, din, (size_t)M * C * 2, hipMemcpyDeviceToDevice);
        a.out_stats = nullptr; a.res = dres; a.y2 = dy2; a.gn_gamma = dg; a.gn_beta = dbt; a.epi_act = ACT_SILU;
        a.se_w1 = dw1; a.se_b1 = db1; a.se_w2 = dw2; a.se_b2 = db2; a.se_hidden = Hd;
    }
#endif
#ifdef PP_TRACE
    unsigned long long* dtr; hipMalloc(&dtr, 2 * 4 * 4 * 4 * 8); hipMemset(dtr, 0, 2 * 4 * 4 * 4 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_pp_trace), &dtr, sizeof(dtr));
#endif
    hipStream_t st; hipStreamCreate(&st);
    for (int i = 0; i < 3; ++i) launch_conv_pp(a, st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, st);
    for (int i = 0; i < iters; ++i) launch_conv_pp(a, st);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / iters, fl = 2.0 * M * C * C * 9.0;
    printf("conv_pp boards=%d: %.1f us/launch, %.1f TFLOP/s  (%s)\n", boards, us, fl / us / 1e6, hipGetErrorString(hipGetLastError()));
#ifdef PP_TRACE
    unsigned long long h[2 * 4 * 4 * 4];
    hipMemcpy(h, dtr, sizeof(h), hipMemcpyDeviceToHost);
    const unsigned long long t0 = h[0];
    for (int g = 0; g < 2; ++g)
        for (int k = 0; k < 4; ++k)
            for (int j = 0; j < 4; ++j) {
                const unsigned long long* p = h + ((g * 4 + k) * 4 + j) * 4;
                printf("group %d kt+%d phase %d: Lstart %6lld  Lend %6lld  Cstart %6lld  Cend %6lld   | L %4lld bar %4lld C %4lld\n", g, k, j,
                       (long long)(p[0] - t0), (long long)(p[1] - t0), (long long)(p[2] - t0), (long long)(p[3] - t0),
                       (long long)(p[1] - p[0]), (long long)(p[2] - p[1]), (long long)(p[3] - p[2]));
            }
#endif
    return 0;
}
