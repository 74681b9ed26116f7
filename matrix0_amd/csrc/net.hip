// Host side of the network forward (see net.h).
#include "net.h"
#include <math.h>
#include <string.h>
#include <algorithm>
#include <stdlib.h>

#define HIPCHK(x)                                                                         \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            err = std::string(#x) + ": " + hipGetErrorString(e_);                         \
            return M0_ERR_HIP;                                                            \
        }                                                                                 \
    } while (0)

static inline int ceil_to(int v, int m) { return (v + m - 1) / m * m; }

static const char* kSslNames[5] = {"piece", "threat", "pin", "fork", "control"};
static const int kSslOut[5] = {13, 1, 1, 1, 3};

const char* Net::check_supported(const m0_net_cfg& c) {
    if (!c.norm_group) return "HIP path supports norm='group' only (BatchNorm configs are not built)";
    if (!c.preact) return "HIP path supports preact=true only";
    if (c.channels % 32 != 0 || c.channels < 32 || c.channels > 384) return "channels must be a multiple of 32 in [32,384]";
    if (c.planes < 1 || c.planes > 32) return "planes must be in [1,32]";
    if (c.attention && (c.attention_heads <= 0 || c.channels % c.attention_heads != 0 ||
                        c.channels / c.attention_heads != 16))
        return "attention head_dim (channels/attention_heads) must be 16";
    if (c.activation != M0_ACT_SILU && c.activation != M0_ACT_RELU) return "activation must be silu or relu";
    if (c.blocks < 1) return "blocks must be >= 1";
    if (c.policy_factor_rank < 0) return "policy_factor_rank must be >= 0";
    if (c.self_supervised && c.ssl_tasks && (c.channels / 2) % 16 != 0) return "ssl heads need channels/2 % 16 == 0";
    return nullptr;
}

Net::Net(const m0_net_cfg& cfg, int device, hipStream_t stream) : cfg_(cfg), device_(device), stream_(stream) {
    // kernel-variant switches (A/B runs): read ONCE, here, so that the layout decisions of forward() and the dispatcher agree
    auto off = [](const char* name) { const char* e = getenv(name); return e && e[0] == '0'; };
    sw_.fuse_tail = !off("M0_FUSE_TAIL");     // =0: conv2 + se_gate + ew_board as separate kernels
    sw_.fuse_small = !off("M0_FUSE_SMALL");   // =0: stem / head convs write raw tensors + statistics for ew_board passes (rounds 1-3)
    sw_.fuse_attn = !off("M0_FUSE_ATTN");     // =0: qkv GEMM + attn_core + proj GEMM + ew_board as separate kernels
    sw_.splitk = !off("M0_SPLITK");           // =0: value_fc1 never splits K
    sw_.conv_zs = !off("M0_CONV_ZS");         // =0: conv_pp16_kernel instead of conv_zs_kernel
    C_ = cfg.channels;
    Cp_ = (C_ > 256 && C_ < 320) ? 320 : C_;
    Cs_ = ceil_to(std::max(16, C_ / 2), 32);
    int k = cfg.attention_every_k;
    int stride = std::max(1, cfg.infer_attention_stride);
    int att_seen = 0;
    for (int i = 0; i < cfg.blocks; ++i) {
        tower_.push_back({0, (int)res_.size(), false});
        res_.emplace_back();
        if (cfg.attention && k > 0 && (i % k) == (k - 1)) {
            ++att_seen;
            bool skip = stride > 1 && (att_seen % stride) != 0;
            tower_.push_back({1, (int)att_.size(), skip});
            att_.emplace_back();
        }
    }
}

Net* Net::shared_view(hipStream_t stream) const {
    Net* v = new Net(*this);                 // member-wise copy: the packed-weight descriptors point at this network's buffers
    v->stream_ = stream;
    v->dev_allocs_.clear(); v->dev_sizes_.clear();          // not owned
    v->ws_allocs_.clear(); v->wsB_ = 0; v->wsM_ = 0;         // own workspace, allocated at its first forward
    v->sd_.clear();
    v->pev_.clear(); v->pflop_.clear(); v->ptail_.clear(); v->pev_used_ = 0; v->profile_ = false;
    v->prof_ms_ = v->prof_flop_ = v->prof_tail_ms_ = 0; v->prof_launches_ = v->prof_tail_launches_ = 0;
    return v;
}

Net::~Net() {
    for (hipEvent_t e : pev_) (void)hipEventDestroy(e);
    for (void* p : dev_allocs_) (void)hipFree(p);
    for (void* p : ws_allocs_) (void)hipFree(p);
}

void* Net::dalloc(size_t bytes, bool ws) {
    void* p = nullptr;
    if (bytes == 0) bytes = 16;
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    // cleared ON THE STREAM THAT WILL USE IT (a NULL-stream hipMemset is not ordered with the non-blocking streams: round 2's
    // regrowth race).  Weight buffers are filled next by blocking NULL-stream copies, so their clear is waited for here.
    (void)hipMemsetAsync(p, 0, bytes, stream_);
    if (!ws) (void)hipStreamSynchronize(stream_);
    (ws ? ws_allocs_ : dev_allocs_).push_back(p);
    if (!ws) dev_sizes_.push_back(bytes);
    return p;
}

float* Net::upload_f32(const std::vector<float>& v) {
    float* d = (float*)dalloc(v.size() * 4, false);
    if (d && !v.empty()) (void)hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice);
    return d;
}

int Net::load(const char* name, const void* data, int dtype, const int64_t* shape, int ndim, std::string& err) {
    if (finalized_) { err = "weights already finalized"; return M0_ERR_STATE; }
    if (!name || (!data && ndim >= 0)) { err = "null argument"; return M0_ERR_INVALID; }
    HostTensor t;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
    t.data.resize(n);
    if (dtype == 0) memcpy(t.data.data(), data, n * 4);
    else if (dtype == 1) {
        const _Float16* h = (const _Float16*)data;
        for (size_t i = 0; i < n; ++i) t.data[i] = (float)h[i];
    } else { err = "dtype must be 0 (f32) or 1 (f16)"; return M0_ERR_INVALID; }
    std::string key(name);
    // resnet.py:1405-1416: legacy rename policy_fc.* -> policy_fc1.* only matters for factorised heads
    if (cfg_.policy_factor_rank > 0 && key.rfind("policy_fc.", 0) == 0) key = "policy_fc1." + key.substr(10);
    sd_[key] = std::move(t);
    return M0_OK;
}

const HostTensor* Net::get(const std::string& k, std::string& err) {
    auto it = sd_.find(k);
    if (it == sd_.end()) { err = "missing state-dict key: " + k; return nullptr; }
    return &it->second;
}

int Net::pack_gemm(PackedGemm& g, const std::string& wkey, const std::string& bkey, int taps, int Cin_real,
                   int Cin_pad, int N_real, int N_pad, int k_perm_ch, std::string& err, int qkv_heads, int qkv_heads_pad) {
    const HostTensor* w = get(wkey, err);
    if (!w) return M0_ERR_INVALID;
    size_t expect = (size_t)N_real * Cin_real * taps;
    if (w->data.size() != expect) { err = "shape mismatch for " + wkey; return M0_ERR_INVALID; }
    nparams_ += expect;
    const int KC = conv_gemm_kc(Cin_pad, N_pad);
    const int nchunk = Cin_pad / KC;
    const bool pp = KC == 64 && taps == 9;       // 3x3 big tile: the half-tile layout of conv_zs_kernel / conv_pp16_kernel
    const int nblk = N_pad / 320;
    std::vector<_Float16> p((size_t)taps * Cin_pad * N_pad, (_Float16)0.f);
    for (int nr = 0; nr < N_real; ++nr)
        for (int k = 0; k < Cin_real; ++k)
            for (int t = 0; t < taps; ++t) {
                float v = w->data[((size_t)nr * Cin_real + k) * taps + t];
                int n = nr;
                if (qkv_heads > 0) {           // qkv rows (t, head, dim) -> the same with the padded head count
                    const int d = nr % 16, th = nr / 16, hh = th % qkv_heads, tt = th / qkv_heads;
                    n = (tt * qkv_heads_pad + hh) * 16 + d;
                }
                int kk = k;
                if (k_perm_ch > 0) {           // reference flatten index c*64+sq -> ours sq*Cp+c
                    int c = k / 64, sq = k % 64;
                    kk = sq * k_perm_ch + c;
                }
                int chunk = kk / KC, kc = kk % KC;
                if (pp) {       // conv_pp16 / conv_pp kernels: half-K-tiles of 320 x 32 k, 64-byte rows, 16-byte chunk index
                                // ^ ((4 - (row >> 2)) & 3): conflict-free for the 16x16x32 and the 32x32x16 fragment reads
                    const int kt = chunk * 9 + t, by = n / 320, nl = n % 320, h = kc >> 5, k32 = kc & 31;
                    const int kx = (((k32 >> 3) ^ ((4 - ((nl >> 2) & 3)) & 3)) << 3) | (k32 & 7);
                    p[((((size_t)kt * nblk + by) * 2 + h) * 320 + nl) * 32 + kx] = (_Float16)v;
                    continue;
                }
                if (KC == 64)   // big tile: LDS image order, 16-byte chunk index XOR (row>>1)&7 (conv_big_kernel)
                    kc = (((kc >> 3) ^ ((n >> 1) & 7)) << 3) | (kc & 7);
                p[(((size_t)t * nchunk + chunk) * N_pad + n) * KC + kc] = (_Float16)v;
            }
    g.w = (_Float16*)dalloc(p.size() * 2, false);
    if (!g.w) { err = "hipMalloc failed"; return M0_ERR_HIP; }
    (void)hipMemcpy(g.w, p.data(), p.size() * 2, hipMemcpyHostToDevice);
    g.taps = taps; g.Cin = Cin_pad; g.N = N_pad; g.pp = pp;
    g.bias = nullptr;
    if (!bkey.empty()) {
        const HostTensor* b = get(bkey, err);
        if (!b) return M0_ERR_INVALID;
        if ((int)b->data.size() != N_real) { err = "shape mismatch for " + bkey; return M0_ERR_INVALID; }
        nparams_ += N_real;
        std::vector<float> bp(N_pad, 0.f);
        std::copy(b->data.begin(), b->data.end(), bp.begin());
        g.bias = upload_f32(bp);
    }
    return M0_OK;
}

// attn_block_kernel operands (attn_block.hip): the block's qkv and proj weights as one stream of 12 KB pieces in LDS
// image order -- per group g of two heads: five pieces [96 cols = q,k,v x 2 heads x 16][64 k] (128-byte rows, 16-byte
// chunk ^ (col>>1)&7) and two pieces [160 output channels][32 k = 2 heads x 16] (64-byte rows, chunk ^ (ch>>2)&3, padded
// to 12 KB) -- and the relative-position bias per (head, query half, lane) in MFMA accumulator order.
int Net::pack_attn_block(AttnW& a, const std::string& prefix, std::string& err) {
    const HostTensor* wq = get(prefix + ".qkv.weight", err); if (!wq) return M0_ERR_INVALID;
    const HostTensor* wp = get(prefix + ".proj.weight", err); if (!wp) return M0_ERR_INVALID;
    const int Cr = C_, Hr = cfg_.attention_heads;
    if (Cp_ != 320 || Cr != Hr * 16 || wq->data.size() != (size_t)3 * Cr * Cr || wp->data.size() != (size_t)Cr * Cr) {
        err = "shape mismatch for " + prefix; return M0_ERR_INVALID;
    }
    std::vector<_Float16> buf(attn_block_pack_bytes() / 2, (_Float16)0.f);
    auto qkv_w = [&](int type, int h, int d, int k) -> float {
        return (h < Hr && k < Cr) ? wq->data[(size_t)((type * Hr + h) * 16 + d) * Cr + k] : 0.f;
    };
    auto proj_w = [&](int oc, int h, int d) -> float {
        return (oc < Cr && h < Hr) ? wp->data[(size_t)oc * Cr + h * 16 + d] : 0.f;
    };
    for (int g = 0; g < 10; ++g) {
        for (int pc = 0; pc < 5; ++pc) {
            const size_t base = (size_t)(g * 7 + pc) * 6144;
            for (int col = 0; col < 96; ++col) {
                const int J = col >> 4, type = J >> 1, hl = J & 1, d = col & 15;
                for (int k = 0; k < 64; ++k) {
                    const int pos = (k >> 3) ^ ((col >> 1) & 7);
                    buf[base + (size_t)col * 64 + pos * 8 + (k & 7)] = (_Float16)qkv_w(type, 2 * g + hl, d, 64 * pc + k);
                }
            }
        }
        for (int hh = 0; hh < 2; ++hh) {
            const size_t base = (size_t)(g * 7 + 5 + hh) * 6144;
            for (int cl = 0; cl < 160; ++cl)
                for (int k = 0; k < 32; ++k) {
                    const int pos = (k >> 3) ^ ((4 - ((cl >> 2) & 3)) & 3);
                    buf[base + (size_t)cl * 32 + pos * 8 + (k & 7)] = (_Float16)proj_w(160 * hh + cl, 2 * g + (k >> 4), k & 15);
                }
        }
    }
    a.blk_w = dalloc(buf.size() * 2, false);
    if (!a.blk_w) { err = "hipMalloc failed"; return M0_ERR_HIP; }
    (void)hipMemcpy(a.blk_w, buf.data(), buf.size() * 2, hipMemcpyHostToDevice);
    std::vector<_Float16> bb((size_t)20 * 2 * 64 * 32, (_Float16)0.f);
    if (cfg_.attention_relbias) {
        const HostTensor* rb = get(prefix + ".rel_bias", err); if (!rb) return M0_ERR_INVALID;
        for (int h = 0; h < Hr; ++h)
            for (int qt = 0; qt < 2; ++qt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 32; ++e) {
                        const int kt = e >> 4, r = e & 15;
                        const int q = qt * 32 + (lane & 31), key = kt * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                        bb[(((size_t)h * 2 + qt) * 64 + lane) * 32 + e] =
                            (_Float16)(rb->data[((size_t)h * 64 + q) * 64 + key] * 1.44269504088896f);
                    }
    }
    a.blk_bias = (_Float16*)dalloc(bb.size() * 2, false);
    if (!a.blk_bias) { err = "hipMalloc failed"; return M0_ERR_HIP; }
    (void)hipMemcpy(a.blk_bias, bb.data(), bb.size() * 2, hipMemcpyHostToDevice);
    return M0_OK;
}

int Net::upload_norm(NormParams& n, const std::string& prefix, int C_real, int C_pad, std::string& err) {
    const HostTensor* g = get(prefix + ".weight", err);
    if (!g) return M0_ERR_INVALID;
    const HostTensor* b = get(prefix + ".bias", err);
    if (!b) return M0_ERR_INVALID;
    if ((int)g->data.size() != C_real || (int)b->data.size() != C_real) { err = "shape mismatch for " + prefix; return M0_ERR_INVALID; }
    nparams_ += 2 * (size_t)C_real;
    std::vector<float> gp(C_pad, 0.f), bp(C_pad, 0.f);
    std::copy(g->data.begin(), g->data.end(), gp.begin());
    std::copy(b->data.begin(), b->data.end(), bp.begin());
    n.gamma = upload_f32(gp);
    n.beta = upload_f32(bp);
    return M0_OK;
}

#define TRY(x) do { int rc_ = (x); if (rc_ != M0_OK) return rc_; } while (0)

int Net::finalize(std::string& err) {
    if (finalized_) return M0_OK;
    if (hipSetDevice(device_) != hipSuccess) { err = "hipSetDevice failed"; return M0_ERR_HIP; }
    const int C = C_, P = Cp_;          // real / in-memory trunk width (padded channels carry zeros everywhere)
    nparams_ = 0;
    TRY(pack_gemm(stem_, "stem.0.weight", "", 9, cfg_.planes, 32, C, P, 0, err));
    TRY(upload_norm(stem_n_, "stem.1", C, P, err));
    if (cfg_.chess_features) {
        const HostTensor* pe = get("chess_features.position_encoding", err);
        if (!pe) return M0_ERR_INVALID;
        if ((int)pe->data.size() != C * 64) { err = "shape mismatch for position_encoding"; return M0_ERR_INVALID; }
        nparams_ += pe->data.size();
        std::vector<float> t((size_t)64 * P, 0.f);
        for (int c = 0; c < C; ++c)
            for (int n = 0; n < 64; ++n) t[(size_t)n * P + c] = pe->data[(size_t)c * 64 + n];
        posenc_ = upload_f32(t);
        if (cfg_.piece_square_tables) {
            TRY(pack_gemm(pst_, "chess_features.pst_conv.weight", "", 1, C, P, C, P, 0, err));
            TRY(upload_norm(pst_n_, "chess_features.pst_norm", C, P, err));
        }
        TRY(pack_gemm(inter_, "chess_features.interaction_conv.weight", "", 9, C, P, C, P, 0, err));
        TRY(upload_norm(inter_n_, "chess_features.interaction_norm", C, P, err));
    }
    int ti = 0;
    for (auto& L : tower_) {
        std::string p = "tower." + std::to_string(ti++);
        if (L.kind == 0) {
            ResBlockW& r = res_[L.index];
            TRY(pack_gemm(r.conv1, p + ".conv1.weight", "", 9, C, P, C, P, 0, err));
            TRY(pack_gemm(r.conv2, p + ".conv2.weight", "", 9, C, P, C, P, 0, err));
            TRY(upload_norm(r.bn1, p + ".bn1", C, P, err));
            TRY(upload_norm(r.bn2, p + ".bn2", C, P, err));
            if (cfg_.se) {
                int hd = std::max(8, (int)(C * cfg_.se_ratio));
                r.se_hidden = hd;
                const HostTensor* w1 = get(p + ".se_fc1.weight", err); if (!w1) return M0_ERR_INVALID;
                const HostTensor* b1 = get(p + ".se_fc1.bias", err); if (!b1) return M0_ERR_INVALID;
                const HostTensor* w2 = get(p + ".se_fc2.weight", err); if (!w2) return M0_ERR_INVALID;
                const HostTensor* b2 = get(p + ".se_fc2.bias", err); if (!b2) return M0_ERR_INVALID;
                if ((int)w1->data.size() != hd * C || (int)w2->data.size() != hd * C || (int)b1->data.size() != hd ||
                    (int)b2->data.size() != C) { err = "shape mismatch for " + p + ".se_*"; return M0_ERR_INVALID; }
                nparams_ += (size_t)2 * hd * C + hd + C;
                std::vector<float> w1t((size_t)P * hd, 0.f);  // [P][hd] from [hd][C]
                for (int j = 0; j < hd; ++j)
                    for (int c = 0; c < C; ++c) w1t[(size_t)c * hd + j] = w1->data[(size_t)j * C + c];
                r.se_w1 = upload_f32(w1t);
                r.se_b1 = upload_f32(b1->data);
                std::vector<float> w2t((size_t)hd * P, 0.f);  // [hd][P] from [C][hd]: coalesced across channels
                for (int c = 0; c < C; ++c)
                    for (int j = 0; j < hd; ++j) w2t[(size_t)j * P + c] = w2->data[(size_t)c * hd + j];
                r.se_w2 = upload_f32(w2t);
                {   // fp16 copies for the fused tail
                    std::vector<_Float16> h1(w1t.size()), h2(w2t.size());
                    for (size_t i = 0; i < w1t.size(); ++i) h1[i] = (_Float16)w1t[i];
                    for (size_t i = 0; i < w2t.size(); ++i) h2[i] = (_Float16)w2t[i];
                    r.se_w1h = (_Float16*)dalloc(h1.size() * 2, false);
                    r.se_w2h = (_Float16*)dalloc(h2.size() * 2, false);
                    if (!r.se_w1h || !r.se_w2h) { err = "hipMalloc failed"; return M0_ERR_HIP; }
                    (void)hipMemcpy(r.se_w1h, h1.data(), h1.size() * 2, hipMemcpyHostToDevice);
                    (void)hipMemcpy(r.se_w2h, h2.data(), h2.size() * 2, hipMemcpyHostToDevice);
                }
                if (P == 320 && hd <= 96) {
                    // conv_zs_kernel's tail runs the two FCs on the matrix cores: B-fragment pieces of v_mfma_f32_16x16x32_f16,
                    // lane (c15 = lane & 15, q = lane >> 4) holds column c15, k = 8q..8q+7 of its tile (conv_zs_tail.h):
                    //   W1 piece (nt, ks):   W1[channel 32 ks + 8 q + e][hidden 16 nt + c15]     nt < ceil(hd/16), ks < 10
                    //   W2 piece (nt, ks):   W2[hidden 32 ks + 8 q + e][channel 16 nt + c15]     nt < 20, ks < ceil(hd/32)
                    const int NT1 = (hd + 15) / 16, KS2 = (hd + 31) / 32;
                    std::vector<_Float16> wf((size_t)(10 * NT1 + 20 * KS2) * 512, (_Float16)0.f);
                    for (int nt = 0; nt < NT1; ++nt)
                        for (int ks = 0; ks < 10; ++ks)
                            for (int l = 0; l < 64; ++l)
                                for (int e = 0; e < 8; ++e) {
                                    const int c = 32 * ks + 8 * (l >> 4) + e, j = 16 * nt + (l & 15);
                                    if (j < hd) wf[((size_t)(nt * 10 + ks) * 64 + l) * 8 + e] = (_Float16)w1t[(size_t)c * hd + j];
                                }
                    for (int nt = 0; nt < 20; ++nt)
                        for (int ks = 0; ks < KS2; ++ks)
                            for (int l = 0; l < 64; ++l)
                                for (int e = 0; e < 8; ++e) {
                                    const int j = 32 * ks + 8 * (l >> 4) + e, c = 16 * nt + (l & 15);
                                    if (j < hd) wf[((size_t)(10 * NT1 + nt * KS2 + ks) * 64 + l) * 8 + e] = (_Float16)w2t[(size_t)j * P + c];
                                }
                    r.se_wf = dalloc(wf.size() * 2, false);
                    if (!r.se_wf) { err = "hipMalloc failed"; return M0_ERR_HIP; }
                    (void)hipMemcpy(r.se_wf, wf.data(), wf.size() * 2, hipMemcpyHostToDevice);
                }
                std::vector<float> b2p(P, 0.f);
                std::copy(b2->data.begin(), b2->data.end(), b2p.begin());
                r.se_b2 = upload_f32(b2p);
            }
        } else {
            AttnW& a = att_[L.index];
            if (L.skip) {   // never executed at inference; count parameters, keep nothing resident
                for (const char* k : {".qkv.weight", ".proj.weight", ".norm.weight", ".norm.bias", ".rel_bias"}) {
                    auto it = sd_.find(p + k);
                    if (it != sd_.end()) nparams_ += it->second.data.size();
                }
                continue;
            }
            TRY(pack_gemm(a.qkv, p + ".qkv.weight", "", 1, C, P, 3 * C, 3 * P, 0, err, P != C ? cfg_.attention_heads : 0, P / 16));
            TRY(pack_gemm(a.proj, p + ".proj.weight", "", 1, C, P, C, P, 0, err));
            TRY(upload_norm(a.ln, p + ".norm", C, P, err));
            if (P == 320) TRY(pack_attn_block(a, p, err));
            if (cfg_.attention_relbias) {
                const HostTensor* rb = get(p + ".rel_bias", err); if (!rb) return M0_ERR_INVALID;
                if ((int)rb->data.size() != cfg_.attention_heads * 4096) { err = "shape mismatch for rel_bias"; return M0_ERR_INVALID; }
                nparams_ += rb->data.size();
                {   // stored pre-multiplied by log2(e): the attention kernel exponentiates with exp2
                    std::vector<float> rbs((size_t)(P / 16) * 4096, 0.f);      // padded heads: zero bias
                    for (size_t i = 0; i < rb->data.size(); ++i) rbs[i] = rb->data[i] * 1.44269504088896f;
                    a.rel_bias = upload_f32(rbs);
                }
            }
        }
    }
    // attention visibility mask, resnet.py:104-130
    {
        std::vector<uint64_t> m(64, 0);
        for (int i = 0; i < 64; ++i)
            for (int j = 0; j < 64; ++j) {
                int dr = i / 8 - j / 8, dc = i % 8 - j % 8;
                int adr = abs(dr), adc = abs(dc);
                bool vis = dr == 0 || dc == 0 || adr == adc || (adr == 2 && adc == 1) || (adr == 1 && adc == 2) ||
                           (adr <= 1 && adc <= 1);
                if (vis) m[i] |= (1ull << j);
            }
        mask_dev_ = (uint64_t*)dalloc(64 * 8, false);
        (void)hipMemcpy(mask_dev_, m.data(), 64 * 8, hipMemcpyHostToDevice);
    }
    // policy head
    TRY(pack_gemm(ph_conv_, "policy_head.0.weight", "", 1, C_, Cp_, 64, 64, 0, err));
    TRY(upload_norm(ph_n_, "policy_head.1", 64, 64, err));
    if (cfg_.policy_factor_rank > 0) {
        int r = cfg_.policy_factor_rank, rp = ceil_to(r, 32);
        TRY(pack_gemm(pfc1_, "policy_fc1.weight", "policy_fc1.bias", 1, 4096, 4096, r, rp, 64, err));
        TRY(pack_gemm(pfc2_, "policy_fc2.weight", "policy_fc2.bias", 1, r, rp, M0_POLICY_SIZE, M0_POLICY_SIZE, 0, err));
    } else {
        TRY(pack_gemm(pfc1_, "policy_fc.weight", "policy_fc.bias", 1, 4096, 4096, M0_POLICY_SIZE, M0_POLICY_SIZE, 64, err));
    }
    {
        const HostTensor* ls = get("_policy_logit_scale_raw", err); if (!ls) return M0_ERR_INVALID;
        nparams_ += 1;
        double raw = ls->data.empty() ? 0.0 : ls->data[0];
        double sp = raw > 20.0 ? raw : log1p(exp(raw));
        logit_scale_ = (float)std::min(5.0, sp + 1e-3);
    }
    // value head
    TRY(pack_gemm(vh0_, "value_head.0.weight", "", 1, C_, Cp_, 128, 128, 0, err));
    TRY(upload_norm(vh1_n_, "value_head.1", 128, 128, err));
    TRY(pack_gemm(vh3_, "value_head.3.weight", "", 1, 128, 128, 128, 128, 0, err));
    TRY(upload_norm(vh4_n_, "value_head.4", 128, 128, err));
    {   // policy_head.0 (64 channels) and value_head.0 (128) read the same trunk: one GEMM with N = 192, columns [policy | value], both
        // in the small-tile layout [Cin / 32][N][32]; GroupNorm parameters concatenated the same way
        const int nch = Cp_ / 32;
        std::vector<_Float16> a((size_t)nch * 64 * 32), b((size_t)nch * 128 * 32), c((size_t)nch * 192 * 32);
        (void)hipMemcpy(a.data(), ph_conv_.w, a.size() * 2, hipMemcpyDeviceToHost);
        (void)hipMemcpy(b.data(), vh0_.w, b.size() * 2, hipMemcpyDeviceToHost);
        for (int ch = 0; ch < nch; ++ch) {
            std::copy(a.begin() + (size_t)ch * 64 * 32, a.begin() + (size_t)(ch + 1) * 64 * 32, c.begin() + (size_t)ch * 192 * 32);
            std::copy(b.begin() + (size_t)ch * 128 * 32, b.begin() + (size_t)(ch + 1) * 128 * 32, c.begin() + ((size_t)ch * 192 + 64) * 32);
        }
        hv_.w = (_Float16*)dalloc(c.size() * 2, false);
        if (!hv_.w) { err = "hipMalloc failed"; return M0_ERR_HIP; }
        (void)hipMemcpy(hv_.w, c.data(), c.size() * 2, hipMemcpyHostToDevice);
        hv_.taps = 1; hv_.Cin = Cp_; hv_.N = 192; hv_.pp = false; hv_.bias = nullptr;
        std::vector<float> g(192), be(192);
        (void)hipMemcpy(g.data(), ph_n_.gamma, 64 * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(g.data() + 64, vh1_n_.gamma, 128 * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(be.data(), ph_n_.beta, 64 * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(be.data() + 64, vh1_n_.beta, 128 * 4, hipMemcpyDeviceToHost);
        hv_n_.gamma = upload_f32(g); hv_n_.beta = upload_f32(be);
        if (!hv_n_.gamma || !hv_n_.beta) { err = "hipMalloc failed"; return M0_ERR_HIP; }
    }
    {
        int h1 = 2 * C, h1p = ceil_to(h1, 32), h2 = C, h2p = ceil_to(C, 32);
        TRY(pack_gemm(vfc1_, "value_fc1.weight", "value_fc1.bias", 1, 8192, 8192, h1, h1p, 128, err));
        TRY(pack_gemm(vfc2_, "value_fc2.weight", "value_fc2.bias", 1, h1, h1p, h2, h2p, 0, err));
        TRY(pack_gemm(vgate_, "value_gate.0.weight", "value_gate.0.bias", 1, h2, h2p, h2, h2p, 0, err));
        TRY(pack_gemm(vfc3_, "value_fc3.weight", "value_fc3.bias", 1, h2, h2p, 1, 32, 0, err));
    }
    // ssl heads
    if (cfg_.self_supervised) {
        for (int t = 0; t < 5; ++t) {
            if (!(cfg_.ssl_tasks & (1 << t))) continue;
            SslHeadW h;
            h.task = kSslNames[t];
            h.out_ch = kSslOut[t];
            h.hidden = C / 2;
            std::string p = std::string("ssl_heads.") + kSslNames[t];
            TRY(pack_gemm(h.c0, p + ".0.weight", "", 1, C_, Cp_, C / 2, Cs_, 0, err));
            TRY(upload_norm(h.n, p + ".1", C / 2, Cs_, err));
            TRY(pack_gemm(h.c1, p + ".3.weight", "", 1, C / 2, Cs_, h.out_ch, 32, 0, err));
            ssl_.push_back(h);
        }
    }
    sd_.clear();
    (void)hipDeviceSynchronize();       // uploads (blocking NULL-stream copies) have landed before any non-blocking stream reads them
    finalized_ = true;
    return M0_OK;
}

int Net::ssl_channels_total() const {
    int n = 0;
    for (auto& h : ssl_) n += h.out_ch;
    return n;
}

double Net::flops_per_position(bool with_ssl) const {
    // MAC count of the executed graph (SURVEY App. A.4 convention: 2 FLOP per MAC)
    const double C = C_;
    double mac = 64.0 * cfg_.planes * 9 * C;                                   // stem
    if (cfg_.chess_features) mac += 64.0 * C * C * (cfg_.piece_square_tables ? 1 : 0) + 64.0 * C * C * 9;
    for (auto& L : tower_) {
        if (L.kind == 0) {
            mac += 2 * 64.0 * C * C * 9;
            if (cfg_.se) mac += 2.0 * C * res_[L.index].se_hidden;
        } else if (!L.skip) {
            mac += 64.0 * C * 3 * C + 64.0 * C * C;                            // qkv + proj
            double nb = (cfg_.attention_unmasked_mix > 0.f && cfg_.attention_unmasked_mix < 1.f) ? 2.0 : 1.0;
            mac += 64.0 * 64.0 * C * (1.0 + nb);                               // QK^T + AV per branch
        }
    }
    mac += 64.0 * C * 64;
    if (cfg_.policy_factor_rank > 0) mac += 4096.0 * cfg_.policy_factor_rank + (double)cfg_.policy_factor_rank * M0_POLICY_SIZE;
    else mac += 4096.0 * M0_POLICY_SIZE;
    mac += 64.0 * C * 128 + 64.0 * 128 * 128 + 8192.0 * 2 * C + 2.0 * C * C + C * C + C;
    if (with_ssl) for (auto& h : ssl_) mac += 64.0 * C * (C / 2) + 64.0 * (C / 2) * h.out_ch;
    return 2.0 * mac;
}

int Net::ensure_workspace(int B, std::string& err) {
    const int Bp = ceil_to(B, 4);
    const int Mfc = ceil_to(B, 256);
    if (Bp <= wsB_ && Mfc <= wsM_) return M0_OK;
    if (hipSetDevice(device_) != hipSuccess) { err = "hipSetDevice failed"; return M0_ERR_HIP; }
    (void)hipDeviceSynchronize();
    for (void* p : ws_allocs_) (void)hipFree(p);
    ws_allocs_.clear();
    const size_t C = Cp_;
    const size_t nb = Bp, nh = std::max(Bp, Mfc);
    const size_t Cst = std::max<size_t>(C, 128);
    auto H = [&](size_t elems) { return (_Float16*)dalloc(elems * 2, true); };
    auto F = [&](size_t elems) { return (float*)dalloc(elems * 4, true); };
    X0_ = H(nb * 64 * 32);
    XA_ = H(nb * 64 * C); XB_ = H(nb * 64 * C); T1_ = H(nb * 64 * C); T2_ = H(nb * 64 * C); AA_ = H(nb * 64 * C);
    QKV_ = H(nb * 64 * 3 * C); O_ = H(nb * 64 * C);
    SX_ = F(nb * Cst * 2); S1_ = F(nb * Cst * 2); S2_ = F(nb * Cst * 2); G_ = F(nb * Cst);
    PH_ = H(nh * 64 * 64); PH2_ = H(nh * 64 * 64);
    VH_ = H(nh * 64 * 128); VH2_ = H(nh * 64 * 128);
    const size_t rp = cfg_.policy_factor_rank > 0 ? ceil_to(cfg_.policy_factor_rank, 32) : 32;
    F1_ = H((size_t)Mfc * rp);
    F2_ = H((size_t)Mfc * ceil_to(2 * C_, 32)); F3_ = H((size_t)Mfc * ceil_to(C_, 32)); F4_ = H((size_t)Mfc * ceil_to(C_, 32));
    SH_ = H(nb * 64 * Cs_); SH2_ = H(nb * 64 * Cs_); SO_ = H(nb * 64 * 32);
    VAL_ = F((size_t)Mfc * 32);
    SPK_ = F((size_t)8 * Mfc * ceil_to(2 * C_, 32));
    if (!X0_ || !XA_ || !XB_ || !AA_ || !T1_ || !T2_ || !QKV_ || !O_ || !SX_ || !S1_ || !S2_ || !PH_ || !PH2_ || !VH_ ||
        !VH2_ || !F1_ || !F2_ || !F3_ || !F4_ || !SH_ || !SH2_ || !SO_ || !VAL_ || !SPK_) {
        err = "workspace hipMalloc failed";
        wsB_ = wsM_ = 0;
        return M0_ERR_HIP;
    }
    // The clears were enqueued on stream_.  A forward may run on ANOTHER stream (the match engine runs network B on network
    // A's stream, selfplay.hip::one_step), so the (rare) regrowth ends by waiting for them: a late clear would zero
    // activations a kernel already wrote.  The wait also surfaces an asynchronous memset failure.
    // (tests/test_net_gpu.py::test_workspace_regrowth_keeps_results, tests/test_arena_gpu.py::test_step_right_after_create)
    if (hipStreamSynchronize(stream_) != hipSuccess || hipGetLastError() != hipSuccess) {
        err = "workspace clear failed"; wsB_ = wsM_ = 0; return M0_ERR_HIP;
    }
    wsB_ = Bp; wsM_ = Mfc;
    return M0_OK;
}

hipError_t Net::run_gemm(const PackedGemm& g, const _Float16* in, void* out, int Mrows, int Mvalid,
                         const NormParams* out_norm, int epi_act, const _Float16* mul, float* out_stats,
                         bool out_f32, float out_scale, hipStream_t st) {
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in; a.w = g.w; a.out = out;
    a.gn_gamma = out_norm ? out_norm->gamma : nullptr;
    a.gn_beta = out_norm ? out_norm->beta : nullptr;
    a.bias = g.bias; a.mul = mul; a.out_stats = out_stats;
    a.Mrows = Mrows; a.Mvalid = Mvalid; a.Cin = g.Cin; a.N = g.N; a.Npad = g.N; a.ldo = g.N;
    a.epi_act = epi_act; a.out_f32 = out_f32 ? 1 : 0; a.out_scale = out_scale; a.w_pp = g.pp ? 1 : 0;
    a.no_zs = sw_.conv_zs ? 0 : 1;
    const bool timed = profile_ && g.taps == 9 && conv_gemm_tile_n(g.Cin, g.N) == 320;
    if (timed) {
        if (pev_used_ + 2 > pev_.size()) {
            for (int i = 0; i < 2; ++i) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return hipErrorOutOfMemory; pev_.push_back(e); }
        }
        (void)hipEventRecord(pev_[pev_used_], st);
    }
    hipError_t rc;
    // a long-K 1x1 GEMM (value_fc1: K = 8192; 32 tiles on 256 CUs at 4096 boards): split K eight ways into fp32 partial
    // tiles, then a fixed-order reduction with the bias and the activation.  Taken at EVERY batch size: the summation order of a
    // board's value must not depend on how many other boards share the launch (tests/test_net_gpu.py: bitwise batch invariance up
    // to the 24 832 boards of a self-play pass); at large batches the partial tiles cost ~0.1 % of the forward.
    const bool splitk = sw_.splitk && g.taps == 1 && conv_gemm_tile_n(g.Cin, g.N) == 320 && g.Cin >= 4096 &&
                        (g.Cin >> 6) % 8 == 0 && !out_norm && !mul && !out_stats && !out_f32 && out_scale == 1.f && SPK_ != nullptr &&
                        (size_t)g.N <= (size_t)ceil_to(2 * C_, 32) && Mrows <= wsM_;
    if (splitk) {
        a.out = SPK_; a.bias = nullptr; a.epi_act = ACT_NONE; a.out_f32 = 1; a.ksplit = 8;
        rc = launch_conv_gemm(a, g.taps, st);
        if (rc == hipSuccess) rc = launch_splitk_reduce(SPK_, 8, Mrows, g.N, g.bias, epi_act, (_Float16*)out, st);
    } else {
        rc = launch_conv_gemm(a, g.taps, st);
    }
    if (timed) {
        (void)hipEventRecord(pev_[pev_used_ + 1], st);
        pev_used_ += 2;
        pflop_.push_back(2.0 * (double)Mvalid * (double)g.N * (double)g.Cin * 9.0);
        ptail_.push_back(0);
    }
    return rc;
}

hipError_t Net::run_conv_tail(const ResBlockW& r, const _Float16* in, const _Float16* x, _Float16* y,
                              const NormParams* next_bn1, _Float16* y2, int act, int Mrows, hipStream_t st) {
    const PackedGemm& g = r.conv2;
    GemmArgs a;
    memset(&a, 0, sizeof(a));
    a.in = in; a.w = g.w; a.out = y;
    a.Mrows = Mrows; a.Mvalid = Mrows; a.Cin = g.Cin; a.N = g.N; a.Npad = g.N; a.ldo = g.N;
    a.epi_act = act; a.out_scale = 1.f; a.w_pp = g.pp ? 1 : 0; a.no_zs = sw_.conv_zs ? 0 : 1;
    a.res = x;
    if (next_bn1 && y2) { a.y2 = y2; a.gn_gamma = next_bn1->gamma; a.gn_beta = next_bn1->beta; }
    if (cfg_.se) { a.se_w1 = r.se_w1; a.se_b1 = r.se_b1; a.se_w2 = r.se_w2; a.se_b2 = r.se_b2; a.se_hidden = r.se_hidden;
                   a.se_w1h = r.se_w1h; a.se_w2h = r.se_w2h; a.se_wf = r.se_wf; }
    const bool timed = profile_;
    if (timed) {
        if (pev_used_ + 2 > pev_.size()) {
            for (int i = 0; i < 2; ++i) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return hipErrorOutOfMemory; pev_.push_back(e); }
        }
        (void)hipEventRecord(pev_[pev_used_], st);
    }
    hipError_t rc = launch_conv_gemm(a, 9, st);
    if (timed) {
        (void)hipEventRecord(pev_[pev_used_ + 1], st);
        pev_used_ += 2;
        pflop_.push_back(2.0 * (double)Mrows * (double)g.N * (double)g.Cin * 9.0);
        ptail_.push_back(1);
    }
    return rc;
}

void Net::harvest_profile() {
    for (size_t i = 0; i + 1 < pev_used_; i += 2) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pev_[i], pev_[i + 1]) == hipSuccess) {
            prof_ms_ += ms; prof_flop_ += pflop_[i / 2]; prof_launches_++;
            if (ptail_[i / 2]) { prof_tail_ms_ += ms; prof_tail_launches_++; }
        }
    }
    pev_used_ = 0;
    pflop_.clear();
    ptail_.clear();
}

#define KCHK(x)                                                                           \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            err = std::string("kernel launch failed: ") + #x + ": " + hipGetErrorString(e_); \
            return M0_ERR_HIP;                                                            \
        }                                                                                 \
    } while (0)

int Net::forward(const float* planes_dev, const _Float16* nhwc_dev, int B, float* logits_dev, float* value_dev,
                 float* ssl_dev, hipStream_t st, std::string& err) {
    if (!finalized_) { err = "weights not finalized"; return M0_ERR_STATE; }
    if (B <= 0) { err = "batch must be positive"; return M0_ERR_INVALID; }
    TRY(ensure_workspace(B, err));
    const int Bp = ceil_to(B, 4);
    const int Mc = Bp * 64;
    const int Mfc = ceil_to(B, 256);
    const int act = cfg_.activation == M0_ACT_SILU ? ACT_SILU : ACT_RELU;
    const int vact = cfg_.value_activation == M0_ACT_SILU ? ACT_SILU : (cfg_.value_activation == M0_ACT_LEAKY ? ACT_LEAKY : ACT_RELU);
    const int C = Cp_;                 // in-memory trunk width

    const _Float16* x0 = nhwc_dev;
    if (!x0) {
        if (!planes_dev) { err = "no input"; return M0_ERR_INVALID; }
        KCHK(launch_planes_to_nhwc(planes_dev, X0_, B, cfg_.planes, st));
        x0 = X0_;
    }
    const bool big = conv_gemm_tile_n(C, C) == 320;     // fused GN epilogue available (C % 320 == 0)
    const bool fuse_tail = big && C == 320 && sw_.fuse_tail &&
                           (!cfg_.se || (res_[0].se_hidden >= 4 && res_[0].se_hidden <= 128 && res_[0].se_hidden % 4 == 0));
    const bool fuse_attn = C == 320 && sw_.fuse_attn;
    // ew: elementwise glue; y2/gn2 = pre-activated input of the NEXT residual block (its bn1), or null
    auto ew = [&](const _Float16* t, const float* tst, const NormParams* gn, const ResBlockW* se, const _Float16* res,
                  const float* pos, const NormParams* ln, _Float16* y, float* ost, const NormParams* next_bn1,
                  _Float16* y2, int Cc, int boards) -> hipError_t {
        EwArgs e;
        memset(&e, 0, sizeof(e));
        e.t = t; e.t_stats = tst;
        if (gn) { e.gn_gamma = gn->gamma; e.gn_beta = gn->beta; }
        if (se) {
            SeGateArgs g;
            g.t_stats = tst; g.w1 = se->se_w1; g.b1 = se->se_b1; g.w2 = se->se_w2; g.b2 = se->se_b2;
            g.gate = G_; g.B = boards; g.C = Cc; g.hidden = se->se_hidden; g.act = act;
            hipError_t ge = launch_se_gate(g, st);
            if (ge != hipSuccess) return ge;
            e.gate = G_;
        }
        e.res = res; e.posenc = pos;
        if (ln) { e.ln_g = ln->gamma; e.ln_b = ln->beta; e.ln_count = C_; }
        e.y = y; e.out_stats = ost; e.C = Cc; e.act = act; e.stats_from_rounded = 0;
        if (next_bn1 && y2) { e.y2 = y2; e.gn2_gamma = next_bn1->gamma; e.gn2_beta = next_bn1->beta; }
        return launch_ew_board(e, boards, st);
    };
    // bn1 of the residual block that consumes the stream right after tower position `pos` (null if the
    // next executed layer is attention or the tower ends there)
    auto next_bn1_after = [&](size_t pos) -> const NormParams* {
        for (size_t j = pos + 1; j < tower_.size(); ++j) {
            if (tower_[j].kind == 1) { if (tower_[j].skip) continue; return nullptr; }
            return &res_[tower_[j].index].bn1;
        }
        return nullptr;
    };
    const NormParams* first_bn1 = nullptr;
    for (size_t j = 0; j < tower_.size(); ++j) {
        if (tower_[j].kind == 1) { if (tower_[j].skip) continue; break; }
        first_bn1 = &res_[tower_[j].index].bn1; break;
    }

    // stem (resnet.py:314-318) + chess features (229-244)
    // fused small kernels (round 4): GroupNorm + activation (+ positional encoding) in the conv's own epilogue
    const bool fuse_small = sw_.fuse_small && C % 64 == 0 && (act == ACT_SILU || act == ACT_RELU);
    auto gn_gemm = [&](const PackedGemm& g, int taps, const _Float16* in, _Float16* out, int ldo, const NormParams& n, const float* pos,
                       _Float16* out2, int ldo2, int nsplit) -> hipError_t {
        GemmArgs ga;
        memset(&ga, 0, sizeof(ga));
        ga.in = in; ga.w = g.w; ga.out = out; ga.gn_gamma = n.gamma; ga.gn_beta = n.beta; ga.posenc = pos;
        ga.Mrows = Mc; ga.Mvalid = Mc; ga.Cin = g.Cin; ga.N = g.N; ga.Npad = g.N; ga.ldo = ldo; ga.epi_act = act; ga.out_scale = 1.f;
        ga.out2 = out2; ga.ldo2 = ldo2; ga.nsplit = nsplit;
        return launch_conv_gemm(ga, taps, st);
    };
    _Float16* xa = XA_;
    _Float16* xb = XB_;
    if (!(fuse_small && cfg_.chess_features)) KCHK(run_gemm(stem_, x0, T1_, Mc, Mc, nullptr, 0, nullptr, S1_, false, 1.f, st));
    if (cfg_.chess_features) {
        if (fuse_small) KCHK(gn_gemm(stem_, 9, x0, xa, C, stem_n_, posenc_, nullptr, 0, 0));
        else KCHK(ew(T1_, S1_, &stem_n_, nullptr, nullptr, posenc_, nullptr, xa, nullptr, nullptr, nullptr, C, Bp));
        if (cfg_.piece_square_tables) {
            if (fuse_tail) {
                // piece-square-table 1x1 conv + GroupNorm/act + residual add in one kernel
                GemmArgs pa;
                memset(&pa, 0, sizeof(pa));
                pa.in = xa; pa.w = pst_.w; pa.out = xb;
                pa.Mrows = Mc; pa.Mvalid = Mc; pa.Cin = pst_.Cin; pa.N = pst_.N; pa.Npad = pst_.N; pa.ldo = pst_.N;
                pa.epi_act = act; pa.out_scale = 1.f;
                pa.res = xa; pa.pre_gamma = pst_n_.gamma; pa.pre_beta = pst_n_.beta;
                KCHK(launch_conv_gemm(pa, 1, st));
            } else {
                KCHK(run_gemm(pst_, xa, T1_, Mc, Mc, nullptr, 0, nullptr, S1_, false, 1.f, st));
                KCHK(ew(T1_, S1_, &pst_n_, nullptr, xa, nullptr, nullptr, xb, nullptr, nullptr, nullptr, C, Bp));
            }
            std::swap(xa, xb);
        }
        if (fuse_tail) {
            // interaction conv + GroupNorm/act + residual add + the first block's GroupNorm/act in one kernel
            GemmArgs ia;
            memset(&ia, 0, sizeof(ia));
            ia.in = xa; ia.w = inter_.w; ia.out = xb;
            ia.Mrows = Mc; ia.Mvalid = Mc; ia.Cin = inter_.Cin; ia.N = inter_.N; ia.Npad = inter_.N; ia.ldo = inter_.N;
            ia.epi_act = act; ia.out_scale = 1.f; ia.w_pp = inter_.pp ? 1 : 0; ia.no_zs = sw_.conv_zs ? 0 : 1;
            ia.res = xa; ia.pre_gamma = inter_n_.gamma; ia.pre_beta = inter_n_.beta;
            if (first_bn1) { ia.y2 = AA_; ia.gn_gamma = first_bn1->gamma; ia.gn_beta = first_bn1->beta; }
            KCHK(launch_conv_gemm(ia, 9, st));
        } else {
            KCHK(run_gemm(inter_, xa, T1_, Mc, Mc, nullptr, 0, nullptr, S1_, false, 1.f, st));
            KCHK(ew(T1_, S1_, &inter_n_, nullptr, xa, nullptr, nullptr, xb, nullptr, first_bn1, AA_, C, Bp));
        }
        std::swap(xa, xb);
    } else {
        KCHK(ew(T1_, S1_, &stem_n_, nullptr, nullptr, nullptr, nullptr, xa, nullptr, first_bn1, AA_, C, Bp));
    }
    // tower
    for (size_t li = 0; li < tower_.size(); ++li) {
        const TowerLayer& L = tower_[li];
        if (L.kind == 0) {
            const ResBlockW& r = res_[L.index];
            // pre-activation block (resnet.py:45-51): AA_ = act(GN1(x)) comes from the previous ew; conv1's
            // epilogue applies GN2+act in registers (big tile) so conv2 also reads a ready operand
            if (big) {
                KCHK(run_gemm(r.conv1, AA_, T1_, Mc, Mc, &r.bn2, act, nullptr, nullptr, false, 1.f, st));
            } else {
                KCHK(run_gemm(r.conv1, AA_, T2_, Mc, Mc, nullptr, 0, nullptr, S1_, false, 1.f, st));
                KCHK(ew(T2_, S1_, &r.bn2, nullptr, nullptr, nullptr, nullptr, T1_, nullptr, nullptr, nullptr, C, Bp));
            }
            if (fuse_tail) {
                // conv2 + squeeze-excite + residual add + the next block's GroupNorm/activation in one kernel
                KCHK(run_conv_tail(r, T1_, xa, xb, next_bn1_after(li), AA_, act, Mc, st));
            } else {
                KCHK(run_gemm(r.conv2, T1_, T2_, Mc, Mc, nullptr, 0, nullptr, S2_, false, 1.f, st));
                KCHK(ew(T2_, S2_, nullptr, cfg_.se ? &r : nullptr, xa, nullptr, nullptr, xb, nullptr, next_bn1_after(li), AA_, C, Bp));
            }
            std::swap(xa, xb);
        } else {
            if (L.skip) continue;
            const AttnW& w = att_[L.index];
            if (fuse_attn && w.blk_w != nullptr) {
                // the whole block (qkv, attention, proj, residual, LayerNorm, next block's GroupNorm/act) in one kernel
                AttnBlockArgs ab;
                memset(&ab, 0, sizeof(ab));
                ab.x = xa; ab.wpack = w.blk_w; ab.bias = w.blk_bias; ab.mask = mask_dev_;
                ab.ln_g = w.ln.gamma; ab.ln_b = w.ln.beta; ab.y = xb;
                if (const NormParams* nb = next_bn1_after(li)) {
                    ab.y2 = AA_; ab.gn2_gamma = nb->gamma; ab.gn2_beta = nb->beta;
                }
                ab.B = Bp; ab.ln_count = C_; ab.act = act; ab.mix = cfg_.attention_unmasked_mix;
                ab.inv_sqrt_d = 1.f / sqrtf((float)(C_ / cfg_.attention_heads));
                KCHK(launch_attn_block(ab, st));
                std::swap(xa, xb);
                continue;
            }
            KCHK(run_gemm(w.qkv, xa, QKV_, Mc, Mc, nullptr, 0, nullptr, nullptr, false, 1.f, st));
            AttnArgs aa;
            aa.qkv = QKV_; aa.rel_bias = w.rel_bias; aa.mask = mask_dev_; aa.o = O_;
            aa.B = Bp; aa.H = C / 16; aa.C = C; aa.mix = cfg_.attention_unmasked_mix;      // head_dim 16; padded heads are all-zero
            aa.inv_sqrt_d = 1.f / sqrtf((float)(C_ / cfg_.attention_heads));
            KCHK(launch_attn_core(aa, st));
            KCHK(run_gemm(w.proj, O_, T1_, Mc, Mc, nullptr, 0, nullptr, nullptr, false, 1.f, st));
            KCHK(ew(T1_, nullptr, nullptr, nullptr, xa, nullptr, &w.ln, xb, nullptr, next_bn1_after(li), AA_, C, Bp));
            std::swap(xa, xb);
        }
    }
    // policy head (resnet.py:699-711); fused: its conv and the value head's first conv read the trunk ONCE, GroupNorm + activation in
    // the epilogue (PH2_ <- policy, VH2_ <- value)
    if (fuse_small) KCHK(gn_gemm(hv_, 1, xa, PH2_, 64, hv_n_, nullptr, VH2_, 128, 64));
    else {
        KCHK(run_gemm(ph_conv_, xa, PH_, Mc, Mc, nullptr, 0, nullptr, S1_, false, 1.f, st));
        KCHK(ew(PH_, S1_, &ph_n_, nullptr, nullptr, nullptr, nullptr, PH2_, nullptr, nullptr, nullptr, 64, Bp));
    }
    if (cfg_.policy_factor_rank > 0) {
        KCHK(run_gemm(pfc1_, PH2_, F1_, Mfc, Mfc, nullptr, ACT_RELU, nullptr, nullptr, false, 1.f, st));
        KCHK(run_gemm(pfc2_, F1_, logits_dev, Mfc, B, nullptr, 0, nullptr, nullptr, true, logit_scale_, st));
    } else {
        KCHK(run_gemm(pfc1_, PH2_, logits_dev, Mfc, B, nullptr, 0, nullptr, nullptr, true, logit_scale_, st));
    }
    // value head (resnet.py:721-734)
    const _Float16* vflat = VH2_;
    if (fuse_small) {
        KCHK(gn_gemm(vh3_, 1, VH2_, VH_, 128, vh4_n_, nullptr, nullptr, 0, 0));
        vflat = VH_;
    } else {
        KCHK(run_gemm(vh0_, xa, VH_, Mc, Mc, nullptr, 0, nullptr, S1_, false, 1.f, st));
        KCHK(ew(VH_, S1_, &vh1_n_, nullptr, nullptr, nullptr, nullptr, VH2_, nullptr, nullptr, nullptr, 128, Bp));
        KCHK(run_gemm(vh3_, VH2_, VH_, Mc, Mc, nullptr, 0, nullptr, S1_, false, 1.f, st));
        KCHK(ew(VH_, S1_, &vh4_n_, nullptr, nullptr, nullptr, nullptr, VH2_, nullptr, nullptr, nullptr, 128, Bp));
    }
    KCHK(run_gemm(vfc1_, vflat, F2_, Mfc, Mfc, nullptr, vact, nullptr, nullptr, false, 1.f, st));
    KCHK(run_gemm(vfc2_, F2_, F3_, Mfc, Mfc, nullptr, vact, nullptr, nullptr, false, 1.f, st));
    KCHK(run_gemm(vgate_, F3_, F4_, Mfc, Mfc, nullptr, ACT_SIGMOID, F3_, nullptr, false, 1.f, st));
    KCHK(run_gemm(vfc3_, F4_, VAL_, Mfc, Mfc, nullptr, ACT_TANH, nullptr, nullptr, true, 1.f, st));
    KCHK(hipMemcpy2DAsync(value_dev, 4, VAL_, 32 * 4, 4, B, hipMemcpyDeviceToDevice, st));
    // ssl heads (resnet.py:738-745)
    if (ssl_dev && !ssl_.empty()) {
        const int ctot = ssl_channels_total();
        int coff = 0;
        const bool ssl_fused = fuse_small && (Cs_ == 160 || Cs_ == 128 || Cs_ == 64 || Cs_ == 32);
        for (auto& h : ssl_) {
            if (ssl_fused) {        // the head's first conv with GroupNorm + activation in its epilogue, all its channels in one workgroup
                KCHK(gn_gemm(h.c0, 1, xa, SH2_, Cs_, h.n, nullptr, nullptr, 0, 0));
            } else {
                KCHK(run_gemm(h.c0, xa, SH_, Mc, Mc, nullptr, 0, nullptr, S1_, false, 1.f, st));
                KCHK(ew(SH_, S1_, &h.n, nullptr, nullptr, nullptr, nullptr, SH2_, nullptr, nullptr, nullptr, Cs_, Bp));
            }
            KCHK(run_gemm(h.c1, SH2_, SO_, Mc, Mc, nullptr, 0, nullptr, nullptr, false, 1.f, st));
            KCHK(launch_nhwc_to_nchw_f32(SO_, ssl_dev, B, 32, h.out_ch, ctot, coff, st));
            coff += h.out_ch;
        }
    }
    return M0_OK;
}
