// Host side of the network: state-dict intake, repacking into kernel layouts,
// workspace management and the forward schedule.
// Mirrors PolicyValueNet (azchess/model/resnet.py:285-582 ctor, 656-760 forward)
// for the inference configuration the self-play path uses.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <map>
#include <string>
#include <vector>
#include "../../include/m0_engine.h"
#include "net_kernels.h"

struct HostTensor {
    std::vector<float> data;
    std::vector<int64_t> shape;
};

struct PackedGemm {
    _Float16* w = nullptr;   // [taps][Cin/KC][N][KC]
    float* bias = nullptr;   // [N] or null
    int taps = 1, Cin = 0, N = 0;
    bool pp = false;         // 3x3 big tile in the half-tile layout of conv_zs_kernel / conv_pp16_kernel [chunk*9+tap][N/320][k half][320][32]
};

struct NormParams {
    float* gamma = nullptr;
    float* beta = nullptr;
};

struct ResBlockW {
    PackedGemm conv1, conv2;
    NormParams bn1, bn2;
    float *se_w1 = nullptr, *se_b1 = nullptr, *se_w2 = nullptr, *se_b2 = nullptr;
    _Float16 *se_w1h = nullptr, *se_w2h = nullptr;   // fp16 copies for the fused tail (conv_tail16.h)
    void* se_wf = nullptr;                           // fp16 MFMA fragment pieces for conv_zs_kernel's tail (conv_zs_tail.h)
    int se_hidden = 0;
};

struct AttnW {
    PackedGemm qkv, proj;
    NormParams ln;
    float* rel_bias = nullptr;
    void* blk_w = nullptr;         // attn_block_kernel: qkv + proj weight pieces (320-channel trunk only)
    _Float16* blk_bias = nullptr;  // rel_bias * log2(e) in accumulator order, fp16
};

struct TowerLayer {
    int kind;    // 0 = residual block, 1 = attention
    int index;   // into res / att vectors
    bool skip;   // inference attention stride (resnet.py:678-687)
};

struct SslHeadW {
    std::string task;
    PackedGemm c0, c1;
    NormParams n;
    int out_ch = 0;
    int hidden = 0;
};

class Net {
public:
    // `stream`: the HIP stream every forward of this network runs on (allocations are cleared on it)
    explicit Net(const m0_net_cfg& cfg, int device, hipStream_t stream);
    ~Net();
    // A second instance over the SAME device weights (read-only after finalize) with its own stream and its own workspace: the
    // self-play engine evaluates the partial last round of a pass on it beside the main forward (selfplay.hip, tail split).  The
    // view owns no weights: it must not outlive the network it was made from.
    Net* shared_view(hipStream_t stream) const;
    static const char* check_supported(const m0_net_cfg& cfg);
    int load(const char* name, const void* data, int dtype, const int64_t* shape, int ndim, std::string& err);
    int finalize(std::string& err);
    bool ready() const { return finalized_; }
    // every device buffer finalize() made (packed GEMM weights, norm parameters, bias / mask tables), in creation order: the same
    // list with the same sizes on every process that finalized a network of the same configuration
    const std::vector<void*>& device_buffers() const { return dev_allocs_; }
    const std::vector<size_t>& device_buffer_bytes() const { return dev_sizes_; }

    // planes: device f32 [B][planes][64] (planes_dev) or device fp16 NHWC [B][64][32] (nhwc_dev)
    // outputs (device): logits f32 [B][4672], value f32 [B]; ssl (optional) f32 concatenated
    int forward(const float* planes_dev, const _Float16* nhwc_dev, int B, float* logits_dev, float* value_dev,
                float* ssl_dev, hipStream_t st, std::string& err);
    int ensure_workspace(int B, std::string& err);
    int ssl_channels_total() const;
    const m0_net_cfg& cfg() const { return cfg_; }
    size_t param_count() const { return nparams_; }
    double flops_per_position(bool with_ssl) const;
    _Float16* input_nhwc() { return X0_; }   // engine-side encoders write here directly
    // Dominant-kernel timing (roofline): when enabled every 3x3 C->C conv launch is bracketed by HIP events on
    // the launch stream; harvest after the stream has been synchronised.
    void set_profile(bool on) { profile_ = on; }
    hipError_t run_conv_tail(const ResBlockW& r, const _Float16* in, const _Float16* x, _Float16* y,
                             const NormParams* next_bn1, _Float16* y2, int act, int Mrows, hipStream_t st);
    void harvest_profile();
    double prof_conv_ms() const { return prof_ms_; }
    double prof_conv_flop() const { return prof_flop_; }
    long prof_conv_launches() const { return prof_launches_; }
    void reset_profile() { prof_ms_ = 0; prof_flop_ = 0; prof_launches_ = 0; prof_tail_ms_ = 0; prof_tail_launches_ = 0; }
    double prof_tail_ms() const { return prof_tail_ms_; }          // the part of the above spent in launches with a fused tail
    long prof_tail_launches() const { return prof_tail_launches_; }

private:
    m0_net_cfg cfg_;
    int device_;
    hipStream_t stream_ = nullptr;
    struct Switches { bool fuse_tail = true, fuse_attn = true, splitk = true, conv_zs = true, fuse_small = true; } sw_;   // read once (constructor)
    bool finalized_ = false;
    size_t nparams_ = 0;
    std::map<std::string, HostTensor> sd_;
    std::vector<void*> dev_allocs_;
    std::vector<size_t> dev_sizes_;          // bytes of dev_allocs_[i] (m0_net_broadcast_weights ships them in this order)
    std::vector<void*> ws_allocs_;

    int C_ = 0, Cs_ = 0 /*ssl hidden padded*/;
    int Cp_ = 0;   // trunk width in memory: C_, or 320 for 256 < C_ < 320 (zero-padded channels: the MFMA big-tile path)
    PackedGemm stem_;
    NormParams stem_n_;
    float* posenc_ = nullptr;
    PackedGemm pst_, inter_;
    NormParams pst_n_, inter_n_;
    std::vector<ResBlockW> res_;
    std::vector<AttnW> att_;
    std::vector<TowerLayer> tower_;
    PackedGemm ph_conv_, pfc1_, pfc2_;
    PackedGemm hv_;          // policy_head.0 and value_head.0 as ONE 1x1 GEMM over the trunk (N = 64 + 128), round 4
    NormParams hv_n_;        // their GroupNorm parameters, concatenated
    NormParams ph_n_;
    float logit_scale_ = 1.f;
    PackedGemm vh0_, vh3_, vfc1_, vfc2_, vgate_, vfc3_;
    NormParams vh1_n_, vh4_n_;
    std::vector<SslHeadW> ssl_;
    uint64_t* mask_dev_ = nullptr;

    bool profile_ = false;
    std::vector<hipEvent_t> pev_;
    std::vector<double> pflop_;
    std::vector<char> ptail_;
    size_t pev_used_ = 0;
    double prof_ms_ = 0, prof_flop_ = 0, prof_tail_ms_ = 0;
    long prof_launches_ = 0, prof_tail_launches_ = 0;

    // workspace
    int wsB_ = 0, wsM_ = 0;
    _Float16 *X0_ = nullptr, *XA_ = nullptr, *XB_ = nullptr, *T1_ = nullptr, *T2_ = nullptr, *QKV_ = nullptr,
             *O_ = nullptr, *AA_ = nullptr;
    float *SX_ = nullptr, *S1_ = nullptr, *S2_ = nullptr, *G_ = nullptr;   // G_: squeeze-excite gates [B][C]
    _Float16 *PH_ = nullptr, *PH2_ = nullptr, *VH_ = nullptr, *VH2_ = nullptr, *F1_ = nullptr, *F2_ = nullptr,
             *F3_ = nullptr, *F4_ = nullptr, *SH_ = nullptr, *SH2_ = nullptr, *SO_ = nullptr;
    float* VAL_ = nullptr;

    const HostTensor* get(const std::string& k, std::string& err);
    int pack_gemm(PackedGemm& g, const std::string& wkey, const std::string& bkey, int taps, int Cin_real,
                  int Cin_pad, int N_real, int N_pad, int k_perm_ch, std::string& err, int qkv_heads = 0,
                  int qkv_heads_pad = 0);
    float* SPK_ = nullptr;      // split-K partial tiles [8][Mfc][2C] f32 (value_fc1)
    int pack_attn_block(AttnW& a, const std::string& prefix, std::string& err);
    int upload_norm(NormParams& n, const std::string& prefix, int C_real, int C_pad, std::string& err);
    float* upload_f32(const std::vector<float>& v);
    void* dalloc(size_t bytes, bool ws);
    hipError_t run_gemm(const PackedGemm& g, const _Float16* in, void* out, int Mrows, int Mvalid,
                        const NormParams* out_norm, int epi_act, const _Float16* mul, float* out_stats,
                        bool out_f32, float out_scale, hipStream_t st);
};
