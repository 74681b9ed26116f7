// Stand-alone check + timing of attn_block_kernel (a whole attention block of the 320-channel trunk) on synthetic data.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench/attn_block_bench.hip -o attn_block_bench
//   ./attn_block_bench [boards] [iters]
// The first 4 boards are compared with a plain fp32 CPU computation of the block (qkv and O rounded to fp16 where the
// kernel rounds them); then the launch is timed.
#include "../../matrix0_amd/csrc/attn_block.hip"
#ifdef AB_AFTER_CONV   // every attention launch behind six conv_zs launches, as in the tower (is its in-network time the convs' clock?)
#include "../../matrix0_amd/csrc/conv_zs.hip"
#endif
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include <algorithm>

static void pack(const std::vector<float>& wq, const std::vector<float>& wp, const std::vector<float>& rb,
                 std::vector<_Float16>& buf, std::vector<_Float16>& bb) {
    const int Cr = 320;
    buf.assign(attn_block_pack_bytes() / 2, (_Float16)0.f);
    for (int g = 0; g < 10; ++g) {
        for (int pc = 0; pc < 5; ++pc) {
            const size_t base = (size_t)(g * 7 + pc) * 6144;
            for (int col = 0; col < 96; ++col) {
                const int J = col >> 4, type = J >> 1, hl = J & 1, d = col & 15;
                for (int k = 0; k < 64; ++k) {
                    const int pos = (k >> 3) ^ ((col >> 1) & 7);
                    buf[base + (size_t)col * 64 + pos * 8 + (k & 7)] =
                        (_Float16)wq[(size_t)((type * 20 + 2 * g + hl) * 16 + d) * Cr + 64 * pc + k];
                }
            }
        }
        for (int hh = 0; hh < 2; ++hh) {
            const size_t base = (size_t)(g * 7 + 5 + hh) * 6144;
            for (int cl = 0; cl < 160; ++cl)
                for (int k = 0; k < 32; ++k) {
                    const int pos = (k >> 3) ^ ((4 - ((cl >> 2) & 3)) & 3);
                    buf[base + (size_t)cl * 32 + pos * 8 + (k & 7)] =
                        (_Float16)wp[(size_t)(160 * hh + cl) * Cr + (2 * g + (k >> 4)) * 16 + (k & 15)];
                }
        }
    }
    bb.assign((size_t)20 * 2 * 64 * 32, (_Float16)0.f);
    for (int h = 0; h < 20; ++h)
        for (int qt = 0; qt < 2; ++qt)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 32; ++e) {
                    const int kt = e >> 4, r = e & 15;
                    const int q = qt * 32 + (lane & 31), key = kt * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                    bb[(((size_t)h * 2 + qt) * 64 + lane) * 32 + e] = (_Float16)(rb[((size_t)h * 64 + q) * 64 + key] * 1.44269504088896f);
                }
}

int main(int argc, char** argv) {
    const int boards = argc > 1 ? atoi(argv[1]) : 4096;
    const int iters = argc > 2 ? atoi(argv[2]) : 20;
    const int C = 320, H = 20;
    const size_t M = (size_t)boards * 64;
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((s >> 11) & 0xffff) / 65536.f - 0.5f; };
    std::vector<_Float16> hx(M * C);
    for (auto& v : hx) v = (_Float16)(rnd() * 2.f);
    std::vector<float> wq((size_t)3 * C * C), wp((size_t)C * C), rb((size_t)H * 4096), lg(C), lb(C), g2(C), b2(C);
    for (auto& v : wq) v = (float)(_Float16)(rnd() * 0.15f);
    for (auto& v : wp) v = (float)(_Float16)(rnd() * 0.15f);
    for (auto& v : rb) v = rnd();
    for (int c = 0; c < C; ++c) { lg[c] = 1.f + 0.2f * rnd(); lb[c] = 0.2f * rnd(); g2[c] = 1.f + 0.2f * rnd(); b2[c] = 0.2f * rnd(); }
    // the last 4 boards repeat the first 4: they are handled in a workgroup's LAST pass over its board pairs (the kernel is
    // persistent), so their outputs must equal the first 4 boards' bit for bit
    if (boards >= 8) memcpy(&hx[(size_t)(boards - 4) * 64 * C], &hx[0], (size_t)4 * 64 * C * 2);
    std::vector<uint64_t> mask(64, 0);
    for (int i = 0; i < 64; ++i)
        for (int j = 0; j < 64; ++j) {
            int dr = i / 8 - j / 8, dc = i % 8 - j % 8, adr = abs(dr), adc = abs(dc);
            bool vis = dr == 0 || dc == 0 || adr == adc || (adr == 2 && adc == 1) || (adr == 1 && adc == 2);
            if (vis) mask[i] |= 1ull << j;
        }
    std::vector<_Float16> buf, bb;
    pack(wq, wp, rb, buf, bb);
    _Float16 *dx, *dy, *dy2, *dbb; void* dw; uint64_t* dm; float *dlg, *dlb, *dg2, *db2;
    hipMalloc(&dx, M * C * 2); hipMalloc(&dy, M * C * 2); hipMalloc(&dy2, M * C * 2); hipMalloc(&dw, buf.size() * 2);
    hipMalloc(&dbb, bb.size() * 2); hipMalloc(&dm, 512); hipMalloc(&dlg, C * 4); hipMalloc(&dlb, C * 4);
    hipMalloc(&dg2, C * 4); hipMalloc(&db2, C * 4);
    hipMemcpy(dx, hx.data(), M * C * 2, hipMemcpyHostToDevice); hipMemcpy(dw, buf.data(), buf.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dbb, bb.data(), bb.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dm, mask.data(), 512, hipMemcpyHostToDevice);
    hipMemcpy(dlg, lg.data(), C * 4, hipMemcpyHostToDevice); hipMemcpy(dlb, lb.data(), C * 4, hipMemcpyHostToDevice);
    hipMemcpy(dg2, g2.data(), C * 4, hipMemcpyHostToDevice); hipMemcpy(db2, b2.data(), C * 4, hipMemcpyHostToDevice);
    hipMemset(dy, 0, M * C * 2); hipMemset(dy2, 0, M * C * 2);
    AttnBlockArgs a{};
    a.x = dx; a.wpack = dw; a.bias = dbb; a.mask = dm; a.ln_g = dlg; a.ln_b = dlb; a.gn2_gamma = dg2; a.gn2_beta = db2;
    a.y = dy; a.y2 = dy2; a.B = boards; a.ln_count = C; a.act = ACT_SILU; a.mix = 0.3f; a.inv_sqrt_d = 0.25f;
    hipStream_t st; hipStreamCreate(&st);
#if defined(AB_STAMP) || defined(AB_STAMP2) || defined(AB_STAMP3)
    unsigned long long* dst_ab; hipMalloc(&dst_ab, (size_t)(boards / 2) * 32 * 8); hipMemset(dst_ab, 0, (size_t)(boards / 2) * 32 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_ab_stamp), &dst_ab, sizeof(dst_ab));
#endif
    hipError_t e = launch_attn_block(a, st);
    hipError_t e2 = hipStreamSynchronize(st);
    if (e != hipSuccess || e2 != hipSuccess) { printf("launch failed: %s / %s\n", hipGetErrorString(e), hipGetErrorString(e2)); return 1; }

#if AB_DBG == 3
    {
        std::vector<_Float16> got(3 * 6144);
        hipMemcpy(got.data(), dy, got.size() * 2, hipMemcpyDeviceToHost);
        int bad = 0;
        for (size_t i = 0; i < got.size(); ++i)
            if (memcmp(&got[i], &buf[i], 2) != 0) { if (bad < 10) printf("ring[%zu] = %f expected %f\n", i, (float)got[i], (float)buf[i]); ++bad; }
        printf("ring check: %d of %zu halfs differ\n", bad, got.size());
        return 0;
    }
#endif
    // ---- CPU check of the first boards
    const int nb = boards < 4 ? boards : 4;
    std::vector<_Float16> gy((size_t)nb * 64 * C), gy2((size_t)nb * 64 * C);
    hipMemcpy(gy.data(), dy, gy.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(gy2.data(), dy2, gy2.size() * 2, hipMemcpyDeviceToHost);
    double maxd = 0, maxd2 = 0; int nbad = 0;
    for (int b = 0; b < nb; ++b) {
        std::vector<float> qkv((size_t)64 * 3 * C), O((size_t)64 * C), y((size_t)64 * C);
        for (int t = 0; t < 64; ++t)
            for (int oc = 0; oc < 3 * C; ++oc) {
                float acc = 0.f;
                for (int k = 0; k < C; ++k) acc += wq[(size_t)oc * C + k] * (float)hx[((size_t)b * 64 + t) * C + k];
                qkv[(size_t)t * 3 * C + oc] = (float)(_Float16)acc;
            }
        for (int h = 0; h < H; ++h)
            for (int q = 0; q < 64; ++q) {
                float sc[64], pu[64], pm[64], su = 0.f, sm = 0.f;
                for (int k = 0; k < 64; ++k) {
                    float d = 0.f;
                    for (int dd = 0; dd < 16; ++dd) d += qkv[(size_t)q * 3 * C + h * 16 + dd] * qkv[(size_t)k * 3 * C + C + h * 16 + dd];
                    d = d * 0.25f + rb[((size_t)h * 64 + q) * 64 + k];
                    d = d < -50.f ? -50.f : (d > 50.f ? 50.f : d);
                    sc[k] = d;
                    pu[k] = expf(d); su += pu[k];
                    pm[k] = ((mask[q] >> k) & 1) ? pu[k] : 0.f; sm += pm[k];
                }
                for (int dd = 0; dd < 16; ++dd) {
                    float o = 0.f;
                    for (int k = 0; k < 64; ++k) {
                        const float p = (float)(_Float16)(0.7f * pm[k] / sm + 0.3f * pu[k] / su);
                        o += p * qkv[(size_t)k * 3 * C + 2 * C + h * 16 + dd];
                    }
                    O[(size_t)q * C + h * 16 + dd] = (float)(_Float16)o;
                }
            }
        for (int t = 0; t < 64; ++t) {
            float v[320]; float s1 = 0.f, s2 = 0.f;
            for (int oc = 0; oc < C; ++oc) {
                float acc = 0.f;
                for (int k = 0; k < C; ++k) acc += wp[(size_t)oc * C + k] * O[(size_t)t * C + k];
                v[oc] = acc + (float)hx[((size_t)b * 64 + t) * C + oc];
                s1 += v[oc]; s2 += v[oc] * v[oc];
            }
            const float mean = s1 / C, rstd = 1.f / sqrtf(s2 / C - mean * mean + 1e-5f);
            for (int oc = 0; oc < C; ++oc) y[(size_t)t * C + oc] = (v[oc] - mean) * rstd * lg[oc] + lb[oc];
        }
        for (int gq = 0; gq < 20; ++gq) {
            float s1 = 0.f, s2 = 0.f;
            for (int t = 0; t < 64; ++t) for (int c = 0; c < 16; ++c) { const float v = y[(size_t)t * C + gq * 16 + c]; s1 += v; s2 += v * v; }
            const float mu = s1 / 1024.f, rstd = 1.f / sqrtf(s2 / 1024.f - mu * mu + 1e-5f);
            for (int t = 0; t < 64; ++t) for (int c = 0; c < 16; ++c) {
                const int ch = gq * 16 + c;
                const size_t gi = ((size_t)b * 64 + t) * C + ch;
                const float yr = (float)(_Float16)y[(size_t)t * C + ch];
                const float z = yr * g2[ch] * rstd + (b2[ch] - mu * g2[ch] * rstd);
                const float r2 = z / (1.f + expf(-z));
                const double d1 = fabs((double)(float)gy[gi] - y[(size_t)t * C + ch]), d2 = fabs((double)(float)gy2[gi] - r2);
                if (!(d1 < 0.03)) { if (nbad < 10) printf("bad y  b %d t %d ch %d: gpu %f cpu %f\n", b, t, ch, (float)gy[gi], y[(size_t)t * C + ch]); ++nbad; }
                if (!(d2 < 0.03)) { if (nbad < 10) printf("bad y2 b %d t %d ch %d: gpu %f cpu %f\n", b, t, ch, (float)gy2[gi], r2); ++nbad; }
                if (d1 > maxd) maxd = d1;
                if (d2 > maxd2) maxd2 = d2;
            }
        }
    }
    printf("check %d boards: max |dy| %.5f  max |dy2| %.5f  bad %d\n", nb, maxd, maxd2, nbad);
    if (boards >= 8) {
        std::vector<_Float16> ly((size_t)4 * 64 * C), ly2((size_t)4 * 64 * C);
        hipMemcpy(ly.data(), dy + (size_t)(boards - 4) * 64 * C, ly.size() * 2, hipMemcpyDeviceToHost);
        hipMemcpy(ly2.data(), dy2 + (size_t)(boards - 4) * 64 * C, ly2.size() * 2, hipMemcpyDeviceToHost);
        const int d1 = memcmp(ly.data(), gy.data(), ly.size() * 2), d2 = memcmp(ly2.data(), gy2.data(), ly2.size() * 2);
        printf("last 4 boards vs first 4 (same input): y %s, y2 %s\n", d1 ? "DIFFER" : "identical", d2 ? "DIFFER" : "identical");
        if (d1 || d2) ++nbad;
    }

    {   // FNV-1a over both outputs of the whole launch: two builds of the kernel are bit-identical iff these agree
        std::vector<uint64_t> all(M * C * 2 / 8);
        uint64_t hsh[2];
        for (int o = 0; o < 2; ++o) {
            hipMemcpy(all.data(), o ? dy2 : dy, M * C * 2, hipMemcpyDeviceToHost);
            uint64_t h = 1469598103934665603ull;
            for (uint64_t v : all) { h ^= v; h *= 1099511628211ull; }
            hsh[o] = h;
        }
        printf("output hash: y %016llx  y2 %016llx\n", (unsigned long long)hsh[0], (unsigned long long)hsh[1]);
        if (const char* hp = getenv("AB_HASH_OUT")) {          // per-board hashes of y, for diffing two builds / runs
            hipMemcpy(all.data(), dy, M * C * 2, hipMemcpyDeviceToHost);
            FILE* f = fopen(hp, "w");
            const size_t per = (size_t)64 * C * 2 / 8;
            for (int b = 0; b < boards; ++b) {
                uint64_t h = 1469598103934665603ull;
                for (size_t i = 0; i < per; ++i) { h ^= all[(size_t)b * per + i]; h *= 1099511628211ull; }
                fprintf(f, "%016llx\n", (unsigned long long)h);
            }
            fclose(f);
        }
    }
#ifdef AB_AFTER_CONV
    {
        std::vector<_Float16> hw((size_t)9 * C * C);
        for (auto& v : hw) v = (_Float16)(rnd() * 0.05f);
        _Float16 *dcw, *dt; hipMalloc(&dcw, hw.size() * 2); hipMalloc(&dt, M * C * 2);
        hipMemcpy(dcw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
        GemmArgs g{};
        g.in = dx; g.w = dcw; g.out = dt; g.Mrows = (int)M; g.Mvalid = (int)M; g.Cin = C; g.N = C; g.Npad = C; g.ldo = C; g.out_scale = 1.f; g.w_pp = 1;
        g.gn_gamma = dg2; g.gn_beta = db2; g.epi_act = ACT_SILU;
        const int nit = iters < 40 ? iters : 40;
        std::vector<hipEvent_t> ev(2 * nit + 2);
        for (auto& e_ : ev) hipEventCreate(&e_);
        for (int w = 0; w < 2; ++w) { for (int k = 0; k < 6; ++k) launch_conv_zs(g, st); launch_attn_block(a, st); }
        hipEventRecord(ev[2 * nit], st);
        for (int i = 0; i < nit; ++i) {
            for (int k = 0; k < 6; ++k) launch_conv_zs(g, st);
            hipEventRecord(ev[2 * i], st);
            launch_attn_block(a, st);
            hipEventRecord(ev[2 * i + 1], st);
        }
        hipEventRecord(ev[2 * nit + 1], st);
        hipStreamSynchronize(st);
        double tot = 0, mn = 1e9, mx = 0; float all = 0.f;
        for (int i = 0; i < nit; ++i) { float m_ = 0.f; hipEventElapsedTime(&m_, ev[2 * i], ev[2 * i + 1]); tot += m_; mn = m_ < mn ? m_ : mn; mx = m_ > mx ? m_ : mx; }
        hipEventElapsedTime(&all, ev[2 * nit], ev[2 * nit + 1]);
        printf("behind six conv_zs launches each (%d rounds, %s): attn_block %.1f us / launch (min %.1f max %.1f); six convs %.1f us each\n", nit,
               hipGetErrorString(hipGetLastError()), tot / nit * 1e3, mn * 1e3, mx * 1e3, (all - tot) / nit / 6.0 * 1e3);
    }
#endif
    for (int i = 0; i < 3; ++i) launch_attn_block(a, st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, st);
    for (int i = 0; i < iters; ++i) launch_attn_block(a, st);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1000.0 / iters;
    const double flop = 2.0 * (double)M * C * (4.0 * C) + 4.0 * (double)boards * H * 64 * 64 * 16;
#ifdef AB_STAMP
    {
        const int nb2 = boards / 2;
        std::vector<unsigned long long> hs((size_t)nb2 * 16);
        hipMemcpy(hs.data(), dst_ab, hs.size() * 8, hipMemcpyDeviceToHost);
        const char* names[] = {"prologue -> first group's qkv done", "group 5: proj of 4 + qkv GEMM", "staging", "attention", "(attention end -> next sequence)",
                               "main loop total", "drain + residual + LayerNorm", "flush y", "GroupNorm + y2 math", "flush y2",
                               "  sequence start -> piece 1 barrier", "  piece 1 (proj)", "  pieces 2, 3 (qkv)", "  pieces 4, 5 (qkv)", "  piece 6 (qkv) to the end"};
        const int a_[] = {0, 8, 5, 6, 7, 0, 2, 3, 4, 9, 8, 12, 13, 14, 15}, b_[] = {1, 5, 6, 7, 11, 2, 3, 4, 9, 10, 12, 13, 14, 15, 5};
        for (int k = 0; k < 15; ++k) {
            std::vector<double> d;
            for (int b = 0; b < nb2; ++b) d.push_back((double)(hs[(size_t)b * 16 + b_[k]] - hs[(size_t)b * 16 + a_[k]]));
            std::sort(d.begin(), d.end());
            printf("  %-40s median %8.0f cycles\n", names[k], d[d.size() / 2]);
        }
    }
#endif
#ifdef AB_STAMP3
    {   // ten register-held stamps inside piece 3 (a qkv piece) of group 5's sequence, waves 0 and 4
        const int nb2 = boards / 2;
        std::vector<unsigned long long> hs((size_t)nb2 * 32);
        hipMemcpy(hs.data(), dst_ab, hs.size() * 8, hipMemcpyDeviceToHost);
        const char* names[] = {"barrier passed -> 5 reads of the next half issued", "2 DMA instructions issued", "MFMAs of the first half issued (6)",
                               "5 reads of the next piece's first half issued", "lgkmcnt(5): this half's fragments there", "MFMAs of the second half issued (6)",
                               "(to the boundary)", "lgkmcnt(0)", "vmcnt(4) + barrier passed"};
        for (int wv = 0; wv < 2; ++wv) {
            printf(" wave %d:\n", 4 * wv);
            for (int k = 0; k < 9; ++k) {
                std::vector<double> d;
                for (int b = 0; b < nb2; ++b) d.push_back((double)(hs[((size_t)b * 2 + wv) * 16 + k + 1] - hs[((size_t)b * 2 + wv) * 16 + k]));
                std::sort(d.begin(), d.end());
                printf("  %-52s median %6.0f  (10 %% %6.0f, 90 %% %6.0f) cycles\n", names[k], d[d.size() / 2], d[d.size() / 10], d[d.size() * 9 / 10]);
            }
        }
    }
#endif
#ifdef AB_STAMP2
    {   // register-held stamps of the sequence "proj of group 4 + qkv of group 5" and the staging after it, waves 0 and 4 of every workgroup
        const int nb2 = boards / 2;
        std::vector<unsigned long long> hs((size_t)nb2 * 32);
        hipMemcpy(hs.data(), dst_ab, hs.size() * 8, hipMemcpyDeviceToHost);
        const char* names[] = {"entry -> start barrier passed", "piece 0 (proj): barrier -> barrier", "piece 1 (proj)", "piece 2 (qkv)", "piece 3 (qkv)",
                               "piece 4 (qkv)", "piece 5 (qkv)", "piece 6 (qkv): barrier -> last MFMA issued", "staging: -> barrier passed"};
        for (int wv = 0; wv < 2; ++wv) {
            printf(" wave %d:\n", 4 * wv);
            for (int k = 0; k < 9; ++k) {
                std::vector<double> d;
                for (int b = 0; b < nb2; ++b) d.push_back((double)(hs[((size_t)b * 2 + wv) * 16 + k + 1] - hs[((size_t)b * 2 + wv) * 16 + k]));
                std::sort(d.begin(), d.end());
                printf("  %-45s median %7.0f  (10 %% %7.0f, 90 %% %7.0f) cycles\n", names[k], d[d.size() / 2], d[d.size() / 10], d[d.size() * 9 / 10]);
            }
        }
        std::vector<double> sk;      // skew: wave 4's entry minus wave 0's
        for (int b = 0; b < nb2; ++b) sk.push_back((double)hs[((size_t)b * 2 + 1) * 16] - (double)hs[((size_t)b * 2) * 16]);
        std::sort(sk.begin(), sk.end());
        printf(" entry of wave 4 minus entry of wave 0: median %.0f (10 %% %.0f, 90 %% %.0f)\n", sk[sk.size() / 2], sk[sk.size() / 10], sk[sk.size() * 9 / 10]);
    }
#endif
    printf("attn_block boards %d: %.1f us / launch, %.3f PFLOP/s, %.2f TB/s (x + y + y2)\n", boards, us, flop / us * 1e-9,
           3.0 * M * C * 2 / us * 1e-6);
    return nbad ? 2 : 0;
}
