// Stand-alone timing / phase trace of conv_pp_kernel (3x3, 320 -> 320 channels) on synthetic data.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DPP_TRACE -DPP_TRACE_KT0=20 -DPP_TRACE_BLOCK=300] \
//         tools/ubench/conv_pp_bench.hip -o conv_pp_bench && ./conv_pp_bench [boards] [iters]
// With -DPP_TRACE it prints, for waves 0 and 4 of one workgroup and 4 consecutive K-tiles, the cycle stamps
// L-start / L-end(before barrier) / C-start(after barrier) / C-end(MFMAs issued) of each phase.
#include "conv_pp.hip"
#include "conv_sw.hip"
#include "../../matrix0_amd/csrc/conv_pp16.hip"
#include "../../matrix0_amd/csrc/conv_zs.hip"
#include "conv_z2.hip"
#include "conv_zd.hip"
#ifdef BENCH_ZD
#define launch_conv_pp launch_conv_zd
#endif
#ifdef BENCH_Z2
#define launch_conv_pp launch_conv_z2
#endif
#ifdef BENCH_ZS
#define launch_conv_pp launch_conv_zs
#endif
#ifdef BENCH_SW
#define launch_conv_pp launch_conv_sw
#endif
#ifdef BENCH_P16
#define launch_conv_pp launch_conv_pp16
#endif
#include <stdio.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include <map>

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
#if defined(BENCH_GN)   // conv1: GroupNorm + SiLU epilogue (EPI 1)
    {
        std::vector<float> gam(C, 1.f), bet(C, 0.f), tbl((size_t)boards * C * 2);
        for (size_t i = 0; i < tbl.size(); i += 2) { tbl[i] = 1.f + rnd() * 0.2f; tbl[i + 1] = rnd() * 0.2f; }
        float *dg, *dbt, *dtb;
        hipMalloc(&dg, C * 4); hipMalloc(&dbt, C * 4); hipMalloc(&dtb, tbl.size() * 4);
        hipMemcpy(dg, gam.data(), C * 4, hipMemcpyHostToDevice); hipMemcpy(dbt, bet.data(), C * 4, hipMemcpyHostToDevice);
        hipMemcpy(dtb, tbl.data(), tbl.size() * 4, hipMemcpyHostToDevice);
        a.out_stats = nullptr; a.gn_gamma = dg; a.gn_beta = dbt; a.epi_act = ACT_SILU;
        (void)dtb;
    }
#endif
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
        {
            std::vector<_Float16> h1(w1.size()), h2(w2.size());
            for (size_t i = 0; i < w1.size(); ++i) h1[i] = (_Float16)w1[i];
            for (size_t i = 0; i < w2.size(); ++i) h2[i] = (_Float16)w2[i];
            _Float16 *dh1, *dh2; hipMalloc(&dh1, h1.size() * 2); hipMalloc(&dh2, h2.size() * 2);
            hipMemcpy(dh1, h1.data(), h1.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dh2, h2.data(), h2.size() * 2, hipMemcpyHostToDevice);
            a.se_w1h = dh1; a.se_w2h = dh2;
            // conv_zs_kernel: MFMA B-fragment pieces (net.hip)
            const int NT1 = (Hd + 15) / 16, KS2 = (Hd + 31) / 32;
            std::vector<_Float16> wf((size_t)(10 * NT1 + 20 * KS2) * 512, (_Float16)0.f);
            for (int nt = 0; nt < NT1; ++nt) for (int ks = 0; ks < 10; ++ks) for (int l = 0; l < 64; ++l) for (int e = 0; e < 8; ++e) {
                const int c = 32 * ks + 8 * (l >> 4) + e, j = 16 * nt + (l & 15);
                if (j < Hd) wf[((size_t)(nt * 10 + ks) * 64 + l) * 8 + e] = (_Float16)w1[(size_t)c * Hd + j];
            }
            for (int nt = 0; nt < 20; ++nt) for (int ks = 0; ks < KS2; ++ks) for (int l = 0; l < 64; ++l) for (int e = 0; e < 8; ++e) {
                const int j = 32 * ks + 8 * (l >> 4) + e, c = 16 * nt + (l & 15);
                if (j < Hd) wf[((size_t)(10 * NT1 + nt * KS2 + ks) * 64 + l) * 8 + e] = (_Float16)w2[(size_t)j * C + c];
            }
            _Float16* dwf; hipMalloc(&dwf, wf.size() * 2); hipMemcpy(dwf, wf.data(), wf.size() * 2, hipMemcpyHostToDevice);
            a.se_wf = dwf;
        }
    }
#endif
#ifdef PP_TRACE
    unsigned long long* dtr; hipMalloc(&dtr, 2 * 4 * 4 * 4 * 8); hipMemset(dtr, 0, 2 * 4 * 4 * 4 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_pp_trace), &dtr, sizeof(dtr));
#endif
#ifdef SW_STAMP
    unsigned long long* dst_; hipMalloc(&dst_, (size_t)(M / 256) * 16 * 8); hipMemset(dst_, 0, (size_t)(M / 256) * 16 * 8);
#if defined(BENCH_SW)
    hipMemcpyToSymbol(HIP_SYMBOL(g_sw_stamp), &dst_, sizeof(dst_));
#elif defined(BENCH_Z2)
    hipMemcpyToSymbol(HIP_SYMBOL(g_z2_stamp), &dst_, sizeof(dst_));
#elif defined(BENCH_ZD)
    hipMemcpyToSymbol(HIP_SYMBOL(g_zd_stamp), &dst_, sizeof(dst_));
    {   // BENCH_CMPZD also launches conv_zs_kernel, whose stamps need a home of their own
        unsigned long long* dzs; hipMalloc(&dzs, (size_t)(M / 256) * 16 * 8); hipMemset(dzs, 0, (size_t)(M / 256) * 16 * 8);
        hipMemcpyToSymbol(HIP_SYMBOL(g_zs_stamp), &dzs, sizeof(dzs));
    }
#elif defined(BENCH_ZS)
    hipMemcpyToSymbol(HIP_SYMBOL(g_zs_stamp), &dst_, sizeof(dst_));
#elif defined(BENCH_P16)
    hipMemcpyToSymbol(HIP_SYMBOL(g_p16_stamp), &dst_, sizeof(dst_));
#else
    hipMemcpyToSymbol(HIP_SYMBOL(g_pp_stamp), &dst_, sizeof(dst_));
#endif
#endif
#if defined(SW_STAMP) && defined(BENCH_ZS) && defined(BENCH_TAIL)
    unsigned long long* dztail_; hipMalloc(&dztail_, (size_t)(M / 256) * 8 * 8); hipMemset(dztail_, 0, (size_t)(M / 256) * 8 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_zs_tail_stamp), &dztail_, sizeof(dztail_));
#endif
#if defined(SW_STAMP) && defined(BENCH_P16) && defined(BENCH_TAIL)
    unsigned long long* dtail_; hipMalloc(&dtail_, (size_t)(M / 256) * 8 * 8); hipMemset(dtail_, 0, (size_t)(M / 256) * 8 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_tail_stamp), &dtail_, sizeof(dtail_));
#endif
    hipStream_t st; hipStreamCreate(&st);
#ifdef BENCH_CMPZD   // conv_zd_kernel (square tiles of 16 boards, all padding skipped) against conv_zs_kernel: same sums in another order
    {
        const size_t nb = (size_t)M * C * 2;
        std::vector<_Float16> o1((size_t)M * C), o2((size_t)M * C), o3((size_t)M * C);
        hipMemset(dout, 0, nb);
        hipError_t e1 = launch_conv_zs(a, st); hipStreamSynchronize(st);
        hipMemcpy(o1.data(), dout, nb, hipMemcpyDeviceToHost);
        hipMemset(dout, 0xff, nb);
        hipError_t e2 = launch_conv_zd(a, st); hipError_t e2s = hipStreamSynchronize(st);
        hipMemcpy(o2.data(), dout, nb, hipMemcpyDeviceToHost);
        hipMemset(dout, 0, nb);
        launch_conv_zd(a, st); hipStreamSynchronize(st);
        hipMemcpy(o3.data(), dout, nb, hipMemcpyDeviceToHost);
        size_t bad = 0, rep = 0, nan = 0; double md = 0, mx = 0; size_t worst = 0;
        for (size_t i = 0; i < o1.size(); ++i) {
            const double d = fabs((double)o1[i] - (double)o2[i]);
            if (!(d == d)) { ++nan; continue; }
            if (d > 4e-3 * (1.0 + fabs((double)o1[i]))) ++bad;
            if (d > md) { md = d; worst = i; }
            if (fabs((double)o1[i]) > mx) mx = fabs((double)o1[i]);
            if (memcmp(&o2[i], &o3[i], 2) != 0) ++rep;
        }
        printf("compare conv_zs (%s) vs conv_zd (%s / %s): %zu of %zu beyond 4e-3 rel, %zu NaN, max |d| %.3g at row %zu col %zu (|out| max %.3g); repeat launch: %zu differing; out[0..3] = %g %g %g %g | %g %g %g %g\n",
               hipGetErrorString(e1), hipGetErrorString(e2), hipGetErrorString(e2s), bad, o1.size(), nan, md, worst / C, worst % C, mx, rep,
               (double)o1[0], (double)o1[1], (double)o1[2], (double)o1[3], (double)o2[0], (double)o2[1], (double)o2[2], (double)o2[3]);
    }
#endif
#ifdef BENCH_CMPZ2   // conv_z2_kernel (2 workgroups per CU) against conv_zs_kernel on the same operands: bit for bit
    {
        const size_t nb = (size_t)M * C * 2;
        std::vector<_Float16> o1((size_t)M * C), o2((size_t)M * C);
        hipMemset(dout, 0, nb);
        hipError_t e1 = launch_conv_zs(a, st); hipStreamSynchronize(st);
        hipMemcpy(o1.data(), dout, nb, hipMemcpyDeviceToHost);
        std::vector<float> s1, s2;
        if (a.out_stats) { s1.resize((size_t)boards * C * 2); hipMemcpy(s1.data(), dstats, s1.size() * 4, hipMemcpyDeviceToHost); }
        hipMemset(dout, 0, nb);
        hipError_t e2 = launch_conv_z2(a, st); hipStreamSynchronize(st);
        hipMemcpy(o2.data(), dout, nb, hipMemcpyDeviceToHost);
        if (a.out_stats) { s2.resize(s1.size()); hipMemcpy(s2.data(), dstats, s2.size() * 4, hipMemcpyDeviceToHost); }
        size_t bad = 0; double md = 0, ms = 0;
        for (size_t i = 0; i < o1.size(); ++i) { const double d = fabs((double)o1[i] - (double)o2[i]); if (d != 0) ++bad; if (d > md) md = d; }
        for (size_t i = 0; i < s1.size(); ++i) { const double d = fabs((double)s1[i] - (double)s2[i]) / (1.0 + fabs((double)s1[i])); if (d > ms) ms = d; }
        printf("compare conv_zs (%s) vs conv_z2 (%s): out %zu differing elements of %zu (max |d| %.3g), stats max rel %.3g; out[0..3] = %g %g %g %g\n",
               hipGetErrorString(e1), hipGetErrorString(e2), bad, o1.size(), md, ms, (double)o2[0], (double)o2[1], (double)o2[2], (double)o2[3]);
    }
#endif
#ifdef BENCH_CMP   // conv_zs_kernel against conv_pp16_kernel on the same operands: outputs must agree bit for bit (stats / gate: closely)
    {
        const size_t nb = (size_t)M * C * 2;
        std::vector<_Float16> o1((size_t)M * C), o2((size_t)M * C), p1((size_t)M * C), p2((size_t)M * C);
        hipMemset(dout, 0, nb); if (a.y2) hipMemset(a.y2, 0, nb);
        hipError_t e1 = launch_conv_pp16(a, st); hipStreamSynchronize(st);
        hipMemcpy(o1.data(), dout, nb, hipMemcpyDeviceToHost); if (a.y2) hipMemcpy(p1.data(), a.y2, nb, hipMemcpyDeviceToHost);
        std::vector<float> s1, s2;
        if (a.out_stats) { s1.resize((size_t)boards * C * 2); hipMemcpy(s1.data(), dstats, s1.size() * 4, hipMemcpyDeviceToHost); }
        hipMemset(dout, 0, nb); if (a.y2) hipMemset(a.y2, 0, nb);
        hipError_t e2 = launch_conv_zs(a, st); hipStreamSynchronize(st);
        hipMemcpy(o2.data(), dout, nb, hipMemcpyDeviceToHost); if (a.y2) hipMemcpy(p2.data(), a.y2, nb, hipMemcpyDeviceToHost);
        if (a.out_stats) { s2.resize(s1.size()); hipMemcpy(s2.data(), dstats, s2.size() * 4, hipMemcpyDeviceToHost); }
        size_t bad = 0, bad2 = 0; double md = 0, md2 = 0, ms = 0;
        for (size_t i = 0; i < o1.size(); ++i) { const double d = fabs((double)o1[i] - (double)o2[i]); if (d != 0) ++bad; if (d > md) md = d; }
        if (a.y2) for (size_t i = 0; i < p1.size(); ++i) { const double d = fabs((double)p1[i] - (double)p2[i]); if (d != 0) ++bad2; if (d > md2) md2 = d; }
        for (size_t i = 0; i < s1.size(); ++i) { const double d = fabs((double)s1[i] - (double)s2[i]) / (1.0 + fabs((double)s1[i])); if (d > ms) ms = d; }
        printf("compare conv_pp16 (%s) vs conv_zs (%s): out %zu differing elements (max |d| %.3g), y2 %zu (max %.3g), stats max rel %.3g; out[0..3] = %g %g %g %g\n",
               hipGetErrorString(e1), hipGetErrorString(e2), bad, md, bad2, md2, ms, (double)o2[0], (double)o2[1], (double)o2[2], (double)o2[3]);
    }
#endif
#ifdef BENCH_Z2     // argv[3] = start delay of the second workgroup of every CU in us
    if (argc > 3) { const int d = (int)(atof(argv[3]) * 100.0); hipMemcpyToSymbol(HIP_SYMBOL(g_z2_delay), &d, sizeof(int)); }
#endif
    for (int i = 0; i < 3; ++i) launch_conv_pp(a, st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, st);
    for (int i = 0; i < iters; ++i) { a.ksplit = i & 1; launch_conv_pp(a, st); }
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / iters, fl = 2.0 * M * C * C * 9.0;
    printf("conv_pp boards=%d: %.1f us/launch, %.1f TFLOP/s  (%s)\n", boards, us, fl / us / 1e6, hipGetErrorString(hipGetLastError()));
#if defined(SW_STAMP) && !defined(BENCH_ZD)
    {
        std::vector<unsigned long long> hs((size_t)(M / 256) * 4);
        hipMemcpy(hs.data(), dst_, hs.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> cyc, clk;
        for (int b = 0; b < M / 256; ++b) {
            const double dc = (double)(hs[b * 4 + 2] - hs[b * 4 + 0]), dr = (double)(hs[b * 4 + 3] - hs[b * 4 + 1]);
            cyc.push_back(dc); clk.push_back(dc / dr * 100.0);
        }
        std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
        printf("main loop per workgroup tile: median %.0f cycles (min %.0f max %.0f), in-kernel clock median %.0f MHz; ideal MFMA issue 115200 cycles -> %.1f %%\n",
               cyc[cyc.size() / 2], cyc.front(), cyc.back(), clk[clk.size() / 2], 100.0 * 115200.0 / cyc[cyc.size() / 2]);
    }
#endif
#if defined(SW_STAMP) && defined(BENCH_ZS) && defined(BENCH_TAIL)
    {
        const int nb = M / 256;
        std::vector<unsigned long long> ht((size_t)nb * 8);
        hipMemcpy(ht.data(), dztail_, ht.size() * 8, hipMemcpyDeviceToHost);
        double d[4] = {0, 0, 0, 0};
        for (int b = 0; b < nb; ++b) for (int k = 0; k < 4; ++k) d[k] += (double)(ht[b * 8 + k + 1] - ht[b * 8 + k]);
        printf("zs tail phases per workgroup (wave 0): SE gate %.2f us, stage + x loads %.2f us, y = x + t / store / stats %.2f us, y2 %.2f us\n",
               d[0] / nb * 0.01, d[1] / nb * 0.01, d[2] / nb * 0.01, d[3] / nb * 0.01);
    }
#endif
#if defined(SW_STAMP) && defined(BENCH_P16) && defined(BENCH_TAIL)
    {
        const int nb = M / 256;
        std::vector<unsigned long long> ht((size_t)nb * 8);
        hipMemcpy(ht.data(), dtail_, ht.size() * 8, hipMemcpyDeviceToHost);
        double d[4] = {0, 0, 0, 0};
        for (int b = 0; b < nb; ++b) for (int k = 0; k < 4; ++k) d[k] += (double)(ht[b * 8 + k + 1] - ht[b * 8 + k]);
        printf("tail phases per workgroup (wave 0): SE gate %.2f us, stage + x loads %.2f us, y = x + t / store / stats %.2f us, y2 %.2f us\n",
               d[0] / nb * 0.01, d[1] / nb * 0.01, d[2] / nb * 0.01, d[3] / nb * 0.01);
    }
#endif
#if defined(SW_STAMP) && defined(BENCH_ZS)
    {   // per-CU timeline of the last launch of conv_zs_kernel: entry -> loop start -> loop end -> exit (after the epilogue), 10-ns ticks
        const int nb = M / 256;
        std::vector<unsigned long long> hs((size_t)nb * 8);
        hipMemcpy(hs.data(), dst_, hs.size() * 8, hipMemcpyDeviceToHost);
        struct Ev { unsigned long long entry, ls, le, x; };
        std::map<unsigned long long, std::vector<Ev>> bycu;
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int b = 0; b < nb; ++b) {
            Ev e{hs[(size_t)nb * 4 + b * 4 + 0], hs[b * 4 + 1], hs[b * 4 + 3], hs[(size_t)nb * 4 + b * 4 + 1]};
            const unsigned long long id = (hs[(size_t)nb * 4 + b * 4 + 3] << 32) | (hs[(size_t)nb * 4 + b * 4 + 2] & 0xffffff00ull);
            bycu[id].push_back(e);
            t0 = std::min(t0, e.entry); t1 = std::max(t1, e.x);
        }
        double pro = 0, loop = 0, epi = 0, gap = 0; int ng = 0, n = 0;
        std::vector<double> busy, first, last, cnt;
        for (auto& kv : bycu) {
            auto& v = kv.second;
            std::sort(v.begin(), v.end(), [](const Ev& a, const Ev& b) { return a.entry < b.entry; });
            double bsum = 0;
            for (size_t i = 0; i < v.size(); ++i) {
                pro += (double)(v[i].ls - v[i].entry); loop += (double)(v[i].le - v[i].ls); epi += (double)(v[i].x - v[i].le); ++n;
                bsum += (double)(v[i].x - v[i].entry);
                if (i) { gap += (double)(v[i].entry - v[i - 1].x); ++ng; }
            }
            busy.push_back(bsum * 0.01); first.push_back((double)(v.front().entry - t0) * 0.01); last.push_back((double)(t1 - v.back().x) * 0.01);
            cnt.push_back((double)v.size());
        }
        std::sort(first.begin(), first.end()); std::sort(last.begin(), last.end()); std::sort(cnt.begin(), cnt.end()); std::sort(busy.begin(), busy.end());
        printf("timeline: %zu CU ids, span %.1f us; per workgroup: prologue %.2f us, main loop %.2f us, epilogue %.2f us; gap exit->next entry %.2f us (n=%d)\n",
               bycu.size(), (t1 - t0) * 0.01, pro / n * 0.01, loop / n * 0.01, epi / n * 0.01, ng ? gap / ng * 0.01 : 0.0, ng);
        printf("  per CU: tiles min %.0f median %.0f max %.0f; busy us min %.1f median %.1f max %.1f; first entry after launch start: median %.1f max %.1f us; idle at the end: median %.1f max %.1f us\n",
               cnt.front(), cnt[cnt.size() / 2], cnt.back(), busy.front(), busy[busy.size() / 2], busy.back(), first[first.size() / 2], first.back(),
               last[last.size() / 2], last.back());
    }
#endif
#if defined(SW_STAMP) && defined(BENCH_ZD)
    {   // conv_zd_kernel: per workgroup entry / loop start / loop end / exit (10-ns ticks) and loop cycles
        const int nwg = ((boards + 15) / 16 + 7) / 8 * 32;
        std::vector<unsigned long long> hs((size_t)nwg * 8);
        hipMemcpy(hs.data(), dst_, hs.size() * 8, hipMemcpyDeviceToHost);
        double pro = 0, loop = 0, epi = 0; std::vector<double> cyc; int n = 0;
        for (int b = 0; b < nwg; ++b) {
            if (hs[b * 8 + 3] == 0) continue;
            pro += (double)(hs[b * 8 + 1] - hs[b * 8 + 0]); loop += (double)(hs[b * 8 + 2] - hs[b * 8 + 1]); epi += (double)(hs[b * 8 + 3] - hs[b * 8 + 2]);
            cyc.push_back((double)hs[b * 8 + 4]); ++n;
        }
        std::sort(cyc.begin(), cyc.end());
        if (n) printf("conv_zd per workgroup (n=%d): prologue %.2f us, main loop %.2f us (median %.0f cycles -> %.0f MHz), epilogue %.2f us\n",
                      n, pro / n * 0.01, loop / n * 0.01, cyc[cyc.size() / 2], cyc[cyc.size() / 2] / (loop / n * 0.01), epi / n * 0.01);
    }
#endif
#if defined(SW_STAMP) && defined(BENCH_P16)
    {   // per-CU timeline of the last launch: entry -> loop start -> loop end -> (before epilogue), realtime ticks of 10 ns
        const int nb = M / 256;
        std::vector<unsigned long long> hs((size_t)nb * 8);
        hipMemcpy(hs.data(), dst_, hs.size() * 8, hipMemcpyDeviceToHost);
        struct Ev { unsigned long long entry, ls, le, x; };
        std::map<unsigned long long, std::vector<Ev>> bycu;
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int b = 0; b < nb; ++b) {
            Ev e{hs[(size_t)nb * 4 + b * 4 + 0], hs[b * 4 + 1], hs[b * 4 + 3], hs[(size_t)nb * 4 + b * 4 + 1]};
            const unsigned long long id = (hs[(size_t)nb * 4 + b * 4 + 3] << 32) | (hs[(size_t)nb * 4 + b * 4 + 2] & 0xffffff00ull);
            bycu[id].push_back(e);
            t0 = std::min(t0, e.entry); t1 = std::max(t1, e.x);
        }
        double pro = 0, loop = 0, gap = 0; int ng = 0, n = 0;
        for (auto& kv : bycu) {
            auto& v = kv.second;
            std::sort(v.begin(), v.end(), [](const Ev& a, const Ev& b) { return a.entry < b.entry; });
            for (size_t i = 0; i < v.size(); ++i) {
                pro += (double)(v[i].ls - v[i].entry); loop += (double)(v[i].le - v[i].ls); ++n;
                if (i) { gap += (double)(v[i].entry - v[i - 1].x); ++ng; }
            }
        }
        {   // idle time between the last two launches on the device's own clock
            std::vector<unsigned long long> h2((size_t)nb * 16);
            hipMemcpy(h2.data(), dst_, h2.size() * 8, hipMemcpyDeviceToHost);
            unsigned long long ent[2] = {~0ull, ~0ull}, ex[2] = {0, 0};
            for (int par = 0; par < 2; ++par)
                for (int b = 0; b < nb; ++b) {
                    const unsigned long long* o = h2.data() + (size_t)nb * 4 + (size_t)par * nb * 8 + b * 4;
                    ent[par] = std::min(ent[par], o[0]); ex[par] = std::max(ex[par], o[1]);
                }
            const int last = (iters - 1) & 1, prev = last ^ 1;
            printf("between the last two launches: previous exit -> last first entry %.2f us; previous span %.1f us, last span %.1f us\n",
                   ((double)ent[last] - (double)ex[prev]) * 0.01, (ex[prev] - ent[prev]) * 0.01, (ex[last] - ent[last]) * 0.01);
        }
        printf("timeline: %zu CU ids, span %.1f us; per workgroup: prologue %.2f us, main loop %.2f us; gap end->next entry %.2f us (n=%d); first entry spread -> see span\n",
               bycu.size(), (t1 - t0) * 0.01, pro / n * 0.01, loop / n * 0.01, ng ? gap / ng * 0.01 : 0.0, ng);
    }
#endif
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
