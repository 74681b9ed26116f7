// Stand-alone check + timing of attn_block4_kernel (round 4: 16 role-specialised waves) against a plain fp32 CPU computation of the
// block (qkv and O rounded to fp16 where the kernel rounds them) and against round 3's attn_block_kernel on the same synthetic data.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ubench/attn_block4_bench.hip -o attn_block4_bench
//   ./attn_block4_bench [boards] [iters]
//   -DA4_DRY    barrier-count check only (the main loop's barriers become counters; all 16 waves must report the same count)
//   -DA4_STAMP  s_memtime stamps of wave 0 (attention role) and wave 8 (GEMM role) of every workgroup
#include "../../matrix0_amd/csrc/attn_block.hip"
#include "attn_block4.hip"
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include <algorithm>

// one 12 KB piece (group g, piece pc) in the LDS image order both kernels share
static void pack_piece(const std::vector<float>& wq, const std::vector<float>& wp, int g, int pc, _Float16* dst, bool kperm = false) {
    const int Cr = 320;
    if (pc < 5) {
        for (int col = 0; col < 96; ++col) {
            const int J = col >> 4, type = J >> 1, hl = J & 1, d = col & 15;
            for (int k = 0; k < 64; ++k) {
                const int pos = (k >> 3) ^ ((col >> 1) & 7);
                const int ksrc = kperm ? (k & ~31) + attn_block4_qkv_kperm(k & 31) : k;
                dst[(size_t)col * 64 + pos * 8 + (k & 7)] = (_Float16)wq[(size_t)((type * 20 + 2 * g + hl) * 16 + d) * Cr + 64 * pc + ksrc];
            }
        }
    } else {
        const int hh = pc - 5;
        for (int cl = 0; cl < 160; ++cl)
            for (int k = 0; k < 32; ++k) {
                const int pos = (k >> 3) ^ ((4 - ((cl >> 2) & 3)) & 3);
                dst[(size_t)cl * 32 + pos * 8 + (k & 7)] = (_Float16)wp[(size_t)(160 * hh + cl) * Cr + (2 * g + (k >> 4)) * 16 + (k & 15)];
            }
    }
}

static void pack_bias(const std::vector<float>& rb, std::vector<_Float16>& bb) {
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
    // the last 4 boards repeat the first 4: their outputs must equal the first 4 boards' bit for bit
    if (boards >= 8) memcpy(&hx[(size_t)(boards - 4) * 64 * C], &hx[0], (size_t)4 * 64 * C * 2);
    std::vector<uint64_t> mask(64, 0);
    for (int i = 0; i < 64; ++i)
        for (int j = 0; j < 64; ++j) {
            int dr = i / 8 - j / 8, dc = i % 8 - j % 8, adr = abs(dr), adc = abs(dc);
            bool vis = dr == 0 || dc == 0 || adr == adc || (adr == 2 && adc == 1) || (adr == 1 && adc == 2);
            if (vis) mask[i] |= 1ull << j;
        }
    std::vector<_Float16> buf1(attn_block_pack_bytes() / 2, (_Float16)0.f), buf2(attn_block4_pack_bytes() / 2, (_Float16)0.f), bb;
    for (int g = 0; g < 10; ++g)
        for (int pc = 0; pc < 7; ++pc) pack_piece(wq, wp, g, pc, &buf1[(size_t)(g * 7 + pc) * 6144]);
    {
        std::vector<int> seen(80, 0);
        for (int g = 0; g < 10; ++g) {
            for (int j = 0; j < 6; ++j) {          // qkv tile j of group g: [16 channels][320 k], 640-byte rows, chunk ^ (row>>1)&7 in 128 B
                _Float16* dst = &buf2[(size_t)attn_block4_qkv_pos(g, j) * 5120];
                seen[attn_block4_qkv_pos(g, j)]++;
                const int type = j >> 1, hl = j & 1;
                for (int col = 0; col < 16; ++col)
                    for (int k = 0; k < 320; ++k) {
                        const int s = k >> 5, kl = k & 31, c = 4 * s + (kl >> 3);
                        const int pos = (c & ~7) | ((c ^ (col >> 1)) & 7);
                        dst[(size_t)col * 320 + pos * 8 + (kl & 7)] =
                            (_Float16)wq[(size_t)((type * 20 + 2 * g + hl) * 16 + col) * 320 + 32 * s + attn_block4_qkv_kperm(kl)];
                    }
            }
            for (int hh = 0; hh < 2; ++hh) {       // proj k-step g, channels 160 hh ..: [160][32 k], 64-byte rows
                _Float16* dst = &buf2[(size_t)attn_block4_proj_pos(g, hh) * 5120];
                seen[attn_block4_proj_pos(g, hh)]++;
                for (int cl = 0; cl < 160; ++cl)
                    for (int k = 0; k < 32; ++k) {
                        const int pos = (k >> 3) ^ ((4 - ((cl >> 2) & 3)) & 3);
                        dst[(size_t)cl * 32 + pos * 8 + (k & 7)] = (_Float16)wp[(size_t)(160 * hh + cl) * 320 + (2 * g + (k >> 4)) * 16 + (k & 15)];
                    }
            }
        }
        for (int t = 0; t < 80; ++t) if (seen[t] != 1) { printf("stream position %d used %d times\n", t, seen[t]); return 3; }
    }
    pack_bias(rb, bb);
    _Float16 *dx, *dy, *dy2, *dyo, *dy2o, *dbb; void *dw1, *dw2; uint64_t* dm; float *dlg, *dlb, *dg2, *db2;
    hipMalloc(&dx, M * C * 2); hipMalloc(&dy, M * C * 2); hipMalloc(&dy2, M * C * 2); hipMalloc(&dyo, M * C * 2); hipMalloc(&dy2o, M * C * 2);
    hipMalloc(&dw1, buf1.size() * 2); hipMalloc(&dw2, buf2.size() * 2);
    hipMalloc(&dbb, bb.size() * 2); hipMalloc(&dm, 512); hipMalloc(&dlg, C * 4); hipMalloc(&dlb, C * 4);
    hipMalloc(&dg2, C * 4); hipMalloc(&db2, C * 4);
    hipMemcpy(dx, hx.data(), M * C * 2, hipMemcpyHostToDevice);
    hipMemcpy(dw1, buf1.data(), buf1.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dw2, buf2.data(), buf2.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dbb, bb.data(), bb.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dm, mask.data(), 512, hipMemcpyHostToDevice);
    hipMemcpy(dlg, lg.data(), C * 4, hipMemcpyHostToDevice); hipMemcpy(dlb, lb.data(), C * 4, hipMemcpyHostToDevice);
    hipMemcpy(dg2, g2.data(), C * 4, hipMemcpyHostToDevice); hipMemcpy(db2, b2.data(), C * 4, hipMemcpyHostToDevice);
    hipMemset(dy, 0, M * C * 2); hipMemset(dy2, 0, M * C * 2); hipMemset(dyo, 0, M * C * 2); hipMemset(dy2o, 0, M * C * 2);
    AttnBlockArgs a{};
    a.x = dx; a.wpack = dw2; a.bias = dbb; a.mask = dm; a.ln_g = dlg; a.ln_b = dlb; a.gn2_gamma = dg2; a.gn2_beta = db2;
    a.y = dy; a.y2 = dy2; a.B = boards; a.ln_count = C; a.act = ACT_SILU; a.mix = 0.3f; a.inv_sqrt_d = 0.25f;
    AttnBlockArgs ao = a;
    ao.wpack = dw1; ao.y = dyo; ao.y2 = dy2o;
    hipStream_t st; hipStreamCreate(&st);
#if defined(A4_STAMP) || defined(A4_STAMP2)
    unsigned long long* dst_ab; hipMalloc(&dst_ab, (size_t)(boards / 2) * 64 * 8); hipMemset(dst_ab, 0, (size_t)(boards / 2) * 64 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_a4_stamp), &dst_ab, sizeof(dst_ab));
#endif
    hipError_t e = launch_attn_block4(a, st);
    hipError_t e2 = hipStreamSynchronize(st);
    if (e != hipSuccess || e2 != hipSuccess) { printf("launch failed: %s / %s\n", hipGetErrorString(e), hipGetErrorString(e2)); return 1; }
#ifdef A4_DRY
    {
        std::vector<int> cnt((size_t)(boards / 2) * 16);
        hipMemcpy(cnt.data(), dy, cnt.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (size_t b = 0; b < cnt.size() / 16; ++b)
            for (int w = 0; w < 16; ++w) if (cnt[b * 16 + w] != cnt[0]) ++bad;
        printf("dry run: barriers per wave of workgroup 0:");
        for (int w = 0; w < 16; ++w) printf(" %d", cnt[w]);
        printf("   mismatches over %zu workgroups: %d\n", cnt.size() / 16, bad);
        return bad ? 4 : 0;
    }
#endif
    launch_attn_block(ao, st);
    hipStreamSynchronize(st);
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
    printf("check %d boards vs CPU: max |dy| %.5f  max |dy2| %.5f  bad %d\n", nb, maxd, maxd2, nbad);
    {   // against round 3's kernel over the whole launch
        std::vector<_Float16> n1(M * C), o1(M * C);
        double md[2] = {0, 0}; size_t ndiff[2] = {0, 0};
        for (int o = 0; o < 2; ++o) {
            hipMemcpy(n1.data(), o ? dy2 : dy, M * C * 2, hipMemcpyDeviceToHost);
            hipMemcpy(o1.data(), o ? dy2o : dyo, M * C * 2, hipMemcpyDeviceToHost);
            for (size_t i = 0; i < M * C; ++i) {
                const double d = fabs((double)(float)n1[i] - (double)(float)o1[i]);
                if (d > md[o]) md[o] = d;
                if (memcmp(&n1[i], &o1[i], 2) != 0) ++ndiff[o];
                if (!(d < 0.02)) ++nbad;
            }
        }
        printf("vs attn_block_kernel (round 3), all %d boards: max |dy| %.5f (%zu halfs differ)  max |dy2| %.5f (%zu differ)\n", boards, md[0],
               ndiff[0], md[1], ndiff[1]);
    }
    if (boards >= 8) {
        std::vector<_Float16> ly((size_t)4 * 64 * C), ly2((size_t)4 * 64 * C);
        hipMemcpy(ly.data(), dy + (size_t)(boards - 4) * 64 * C, ly.size() * 2, hipMemcpyDeviceToHost);
        hipMemcpy(ly2.data(), dy2 + (size_t)(boards - 4) * 64 * C, ly2.size() * 2, hipMemcpyDeviceToHost);
        const int d1 = memcmp(ly.data(), gy.data(), ly.size() * 2), d2 = memcmp(ly2.data(), gy2.data(), ly2.size() * 2);
        printf("last 4 boards vs first 4 (same input): y %s, y2 %s\n", d1 ? "DIFFER" : "identical", d2 ? "DIFFER" : "identical");
        if (d1 || d2) ++nbad;
    }
    uint64_t hsh[2];
    {   // FNV-1a over both outputs of the whole launch: two builds / runs are bit-identical iff these agree
        std::vector<uint64_t> all(M * C * 2 / 8);
        for (int o = 0; o < 2; ++o) {
            hipMemcpy(all.data(), o ? dy2 : dy, M * C * 2, hipMemcpyDeviceToHost);
            uint64_t h = 1469598103934665603ull;
            for (uint64_t v : all) { h ^= v; h *= 1099511628211ull; }
            hsh[o] = h;
        }
        printf("output hash: y %016llx  y2 %016llx\n", (unsigned long long)hsh[0], (unsigned long long)hsh[1]);
    }
    auto time_it = [&](bool neu) {
        for (int i = 0; i < 3; ++i) neu ? launch_attn_block4(a, st) : launch_attn_block(ao, st);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, st);
        for (int i = 0; i < iters; ++i) neu ? launch_attn_block4(a, st) : launch_attn_block(ao, st);
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
        float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
        return ms * 1000.0 / iters;
    };
    const double flop = 2.0 * (double)M * C * (4.0 * C) + 4.0 * (double)boards * H * 64 * 64 * 16;
    const double us_old = time_it(false), us_new = time_it(true), us_old2 = time_it(false), us_new2 = time_it(true);
    {   // determinism: the repeated launches above must have left the very same outputs
        std::vector<uint64_t> all(M * C * 2 / 8);
        for (int o = 0; o < 2; ++o) {
            hipMemcpy(all.data(), o ? dy2 : dy, M * C * 2, hipMemcpyDeviceToHost);
            uint64_t h = 1469598103934665603ull;
            for (uint64_t v : all) { h ^= v; h *= 1099511628211ull; }
            if (h != hsh[o]) { printf("NOT DETERMINISTIC: output %d hash %016llx after the timed launches\n", o, (unsigned long long)h); ++nbad; }
        }
    }
#ifdef A4_STAMP2
    {   // per wave 0 / 4 (attention) and 8 / 12 (GEMM): work time between barriers and wait time at them, intervals 18-23
        const int nb2 = boards / 2;
        std::vector<unsigned long long> hs((size_t)nb2 * 64);
        hipMemcpy(hs.data(), dst_ab, hs.size() * 8, hipMemcpyDeviceToHost);
        for (int wv = 0; wv < 4; ++wv) {
            printf(" wave %2d (%s):", 4 * wv, wv < 2 ? "attention" : "GEMM");
            for (int k = 0; k < 11; ++k) {
                std::vector<double> d;
                for (int b = 0; b < nb2; ++b) d.push_back((double)(hs[((size_t)b * 4 + wv) * 16 + k + 1] - hs[((size_t)b * 4 + wv) * 16 + k]));
                std::sort(d.begin(), d.end());
                printf(" %s%5.0f", (k & 1) ? "work " : "wait ", d[d.size() / 2]);
            }
            printf("\n");
        }
    }
#endif
#ifdef A4_STAMP
    {
        const int nb2 = boards / 2;
        std::vector<unsigned long long> hs((size_t)nb2 * 32);
        hipMemcpy(hs.data(), dst_ab, hs.size() * 8, hipMemcpyDeviceToHost);
        const char* namesA[] = {"prologue + qkv(0) (3 intervals)", "periods 0-4 (15 intervals)", "period 5 (3 intervals)", "periods 6-8 (9 intervals)",
                                "period 9 (3 intervals, beside proj)", "proj (7 intervals)", "final barrier", "", ""};
        const char* namesG[] = {"prologue + qkv(0) (3 intervals)", "periods 0-4 (15 intervals)", "period 5 (3 intervals)", "periods 6-8 (9 intervals)",
                                "proj (10 intervals)", "final barrier", "LayerNorm + flush y", "GroupNorm + y2", ""};
        for (int role = 0; role < 2; ++role) {
            printf(" %s wave:\n", role ? "GEMM" : "attention");
            const char** names = role ? namesG : namesA;
            for (int k = 0; k < (role ? 8 : 7); ++k) {
                std::vector<double> d;
                for (int b = 0; b < nb2; ++b) d.push_back((double)(hs[((size_t)b * 2 + role) * 16 + k + 1] - hs[((size_t)b * 2 + role) * 16 + k]));
                std::sort(d.begin(), d.end());
                printf("  %-40s median %8.0f cycles\n", names[k], d[d.size() / 2]);
            }
        }
    }
#endif
    printf("attn_block  (round 3) boards %d: %.1f / %.1f us per launch, %.3f PFLOP/s\n", boards, us_old, us_old2, flop / us_old2 * 1e-9);
    printf("attn_block4 (round 4) boards %d: %.1f / %.1f us per launch, %.3f PFLOP/s\n", boards, us_new, us_new2, flop / us_new2 * 1e-9);
    return nbad ? 2 : 0;
}
