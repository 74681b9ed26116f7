// Launch interfaces of the network kernels (net_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2, ACT_LEAKY = 3, ACT_SIGMOID = 4, ACT_TANH = 5 };

struct GemmArgs {
    const _Float16* in;      // [Mrows][Cin]
    const _Float16* w;       // packed [taps][Cin/KC][Npad][KC] (big tile: 16-byte chunks XOR-swizzled, see pack_gemm)
    void* out;               // [Mrows][ldo] fp16 | f32
    const float* gn_gamma;   // [N]: epilogue GroupNorm16 (+epi_act) over each board's 64 rows, big tile only
    const float* gn_beta;    // [N]
    const float* bias;       // [N] or null
    const _Float16* mul;     // [Mrows][ldo] or null
    float* out_stats;        // [Mrows/64][N][2] or null
    int Mrows;               // multiple of 256
    int Mvalid;              // rows actually stored
    int Cin;                 // multiple of KC
    int N;                   // valid output columns
    int Npad;                // packed columns (multiple of the N tile)
    int ldo;                 // output row stride (elements)
    int epi_act;
    int out_f32;
    float out_scale;
    int ksplit;              // conv_big_kernel<1>: > 1 = split K over gridDim.y, workgroup z writes its fp32 partial tile to
                             // out + z * Mrows * ldo floats (out_f32 = 1, no bias / act); reduced by launch_splitk_reduce
    int w_pp;                // 3x3 big tile: w is in the half-tile layout of conv_zs_kernel / conv_pp16_kernel (pack_gemm)
    // residual-block tail fused into the epilogue (conv_tail.h), when res != null: out = res + gate * conv,
    // y2 = epi_act(GroupNorm16(out; gn_gamma, gn_beta)) when y2 != null, gate from se_* when se_w1 != null
    const _Float16* res;     // [Mrows][ldo]
    const float* pre_gamma;  // with res: conv output first through act(GroupNorm16(.; pre_gamma, pre_beta)) (no gate)
    const float* pre_beta;
    _Float16* y2;            // [Mrows][ldo] or null
    const float* se_w1;      // [C][Hd] (transposed)
    const float* se_b1;
    const float* se_w2;      // [Hd][C] (transposed)
    const float* se_b2;
    int se_hidden;
    const _Float16* se_w1h;  // fp16 copies of se_w1 / se_w2 (same layouts): conv_pp16's tail stages BOTH in LDS with one DMA
    const _Float16* se_w2h;  //   wave (half the bytes of the f32 matrices, which it brought in one after the other)
    int no_zs;               // 3x3 big tile: 1 = conv_pp16_kernel instead of conv_zs_kernel (M0_CONV_ZS=0, read once per network)
    const void* se_wf;       // conv_zs_kernel's tail: W1 and W2 as fp16 MFMA B-fragment pieces of 1 KiB (conv_zs_tail.h; net.hip packs)
    // small tile with gn_gamma != null (conv_gemm_kernel, EPI 1): out = epi_act(GroupNorm16(conv)) [+ posenc], fp16, through the
    // LDS-staged 16-byte-store epilogue.  Two consumers of one input in ONE launch (policy-head and value-head 1x1 convs over the
    // trunk): columns >= nsplit go to out2 (row stride ldo2, column - nsplit); gn_gamma / gn_beta cover all N columns.
    const float* posenc;     // [64 squares][N] f32 added after the activation (stem), or null
    void* out2;
    int ldo2;
    int nsplit;
};

struct EwArgs {
    const _Float16* t;       // [B][64][C]
    const float* t_stats;    // [B][C][2]
    const float* gn_gamma;   // GroupNorm16 + act when non-null
    const float* gn_beta;
    const float* gate;       // [B][C] squeeze-excite gate (se_gate_kernel), applied when non-null (and gn null)
    const _Float16* res;     // [B][64][C] or null
    const float* posenc;     // [64][C] or null
    const float* ln_g;       // LayerNorm over the first ln_count channels when non-null (the rest are zero padding)
    const float* ln_b;
    int ln_count;            // 0: C
    _Float16* y;             // [B][64][C]
    float* out_stats;        // [B][C][2] or null
    _Float16* y2;            // [B][64][C] or null: act(GroupNorm16(y; gn2_*)), the next block's conv1 input
    const float* gn2_gamma;
    const float* gn2_beta;
    int C;
    int act;
    int stats_from_rounded;
};

struct SeGateArgs {
    const float* t_stats;    // [B][C][2] per-(board, channel) (sum, sumsq) over the 64 squares (conv epilogue)
    const float* w1;         // [C][Hd] (transposed)
    const float* b1;         // [Hd]
    const float* w2;         // [Hd][C] (transposed)
    const float* b2;         // [C]
    float* gate;             // [B][C]
    int B, C, hidden, act;
};

struct AttnArgs {
    const _Float16* qkv;     // [B][64][3C]
    const float* rel_bias;   // [H][64][64] * log2(e), or null
    const uint64_t* mask;    // [64] bit j of word i = key j visible from query i
    _Float16* o;             // [B][64][C]
    int B, H, C;
    float mix;
    float inv_sqrt_d;
};

// attn_block_kernel (attn_block.hip): a whole attention block of the 320-channel trunk in one kernel
struct AttnBlockArgs {
    const _Float16* x;       // [B][64][320] trunk
    const void* wpack;       // attn_block_pack_bytes(): 70 (+3 pad) weight pieces of 12 KB in LDS image order (net.hip)
    const _Float16* bias;    // [20][2][64][32] rel_bias * log2(e) in accumulator order (zeros when the net has none)
    const uint64_t* mask;    // [64] bit j of word i = key j visible from query i
    const float* ln_g;       // LayerNorm [320] (zero beyond ln_count)
    const float* ln_b;
    const float* gn2_gamma;  // with y2: GroupNorm16 + act of the next residual block
    const float* gn2_beta;
    _Float16* y;             // [B][64][320] LayerNorm(x + attention(x))
    _Float16* y2;            // [B][64][320] or null
    int B;                   // boards, even
    int ln_count;            // real channel count of the LayerNorm
    int act;                 // ACT_SILU / ACT_RELU (y2)
    float mix;
    float inv_sqrt_d;
};
hipError_t launch_attn_block(const AttnBlockArgs& a, hipStream_t st);
// out[m][n] = act(sum_z part[z][m][n] + bias[n]) as fp16 (fixed summation order), n < N, row stride ld
hipError_t launch_splitk_reduce(const float* part, int splits, int M, int N, const float* bias, int act, _Float16* out,
                                hipStream_t st);
size_t attn_block_pack_bytes();

hipError_t launch_conv_gemm(const GemmArgs& a, int taps, hipStream_t st);
int conv_gemm_tile_n(int Cin, int Npad);
int conv_gemm_kc(int Cin, int Npad);
hipError_t launch_ew_board(const EwArgs& a, int boards, hipStream_t st);
hipError_t launch_se_gate(const SeGateArgs& a, hipStream_t st);
hipError_t launch_attn_core(const AttnArgs& a, hipStream_t st);
hipError_t launch_planes_to_nhwc(const float* x, void* y, int B, int P, hipStream_t st);
hipError_t launch_nhwc_to_nchw_f32(const void* x, float* y, int B, int ld, int n, int ctot, int coff, hipStream_t st);
