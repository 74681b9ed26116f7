/* libm0engine — C ABI of the MI355X-native Matrix0 self-play hot path.
 *
 * Drop-in boundary for two duck-typed seams of the reference (lukifer23/Matrix0):
 *
 *   1. inference backend  obj.infer_np(np.float32[B,19,8,8]) -> (np.float32[B,4672], np.float32[B])
 *        azchess/mcts.py:618-621, 1021-1023; azchess/selfplay/inference.py:585-645 (InferenceClient.infer_np)
 *        -> m0_infer()
 *   2. worker entry       selfplay_worker(proc_id, cfg_dict, ckpt_path, games, q, shared_memory_resource)
 *        azchess/selfplay/internal.py:94-95 (called from orchestrator.py:494, selfplay/__main__.py:68)
 *        -> m0_selfplay_start() / m0_selfplay_step() / m0_selfplay_poll()
 *
 *   plus the pure functions of azchess/encoding.py, exposed position-wise for parity tests and
 *   for callers that keep python-chess:  m0_encode_fen(), m0_legal_mask_fen(), m0_move_to_index_fen().
 *
 * Conventions: every function returns 0 on success or a negative code; the message is
 * available from m0_last_error() (thread-local).  No exception crosses the ABI.  All buffers
 * are caller-allocated host memory unless a parameter says "_dev".  Handles are opaque.
 * The library never falls back to a CPU path: without a HIP device m0_create() fails.
 */
#ifndef M0_ENGINE_H
#define M0_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define M0_OK 0
#define M0_ERR_INVALID (-1)
#define M0_ERR_UNSUPPORTED (-2)
#define M0_ERR_HIP (-3)
#define M0_ERR_STATE (-4)
#define M0_ERR_NONFINITE (-5) /* NaN/Inf in network output: mcts.py:1165-1177 raises */

#define M0_POLICY_SIZE 4672
#define M0_PLANES 19

/* activation codes */
#define M0_ACT_RELU 1
#define M0_ACT_SILU 2
#define M0_ACT_LEAKY 3

/* ssl task bits (order = NPZ field order of selfplay/internal.py:647-651) */
#define M0_SSL_PIECE 1
#define M0_SSL_THREAT 2
#define M0_SSL_PIN 4
#define M0_SSL_FORK 8
#define M0_SSL_CONTROL 16

/* NetConfig, azchess/model/resnet.py:247-282 (inference-relevant fields only). */
typedef struct m0_net_cfg {
    int planes;               /* 19 */
    int channels;             /* 320 */
    int blocks;               /* 24 */
    int attention;            /* bool */
    int attention_heads;      /* 20 */
    int attention_every_k;    /* 3 */
    int attention_relbias;    /* bool */
    float attention_unmasked_mix; /* 0.2 */
    int se;                   /* bool */
    float se_ratio;           /* 0.25 */
    int chess_features;       /* bool */
    int piece_square_tables;  /* bool */
    int policy_factor_rank;   /* 128; 0 = dense policy_fc */
    int norm_group;           /* 1 = GroupNorm (supported); 0 = BatchNorm (unsupported on the HIP path) */
    int activation;           /* M0_ACT_SILU | M0_ACT_RELU */
    int value_activation;     /* M0_ACT_SILU | M0_ACT_LEAKY | M0_ACT_RELU */
    int preact;               /* bool; only preact=1 is supported on the HIP path */
    int self_supervised;      /* bool */
    int ssl_tasks;            /* bitmask of M0_SSL_* */
    int infer_attention_stride; /* >=1 */
} m0_net_cfg;

typedef struct m0_net m0_net;

const char* m0_last_error(void);
const char* m0_version(void);

/* ---- network (seam 1) ---- */
m0_net* m0_net_create(const m0_net_cfg* cfg, int hip_device);
void m0_net_destroy(m0_net* net);
/* One call per state-dict key (reference key names, resnet.py state_dict()); dtype 0 = f32, 1 = f16.
 * Unknown keys are ignored (load_state_dict(strict=False), selfplay/internal.py:172-174). */
int m0_net_load_weight(m0_net* net, const char* name, const void* data, int dtype, const int64_t* shape, int ndim);
/* Repack to kernel layouts (fp16 MFMA operand tiles), upload; missing keys are an error. */
int m0_net_finalize(m0_net* net);
/* infer_np: planes f32 [B,19,8,8] -> policy logits f32 [B,4672], value f32 [B];
 * ssl (nullable) f32 [B, ssl_channels, 8, 8] in task order piece(13) threat(1) pin(1) fork(1) control(3).
 * Re-entrant: calls on one net are serialised internally (several MCTS objects may share a backend,
 * tests/test_stress.py:221-266).  NaN/Inf in the outputs -> M0_ERR_NONFINITE. */
int m0_net_infer(m0_net* net, const float* planes, int B, float* policy, float* value, float* ssl);
int m0_net_ssl_channels(const m0_net* net);
int64_t m0_net_param_count(const m0_net* net);
double m0_net_flops_per_position(const m0_net* net, int with_ssl);
/* Timed forward on synthetic resident inputs (bench/roofline): runs `iters` forwards of batch B on the
 * net's stream and returns the mean milliseconds per forward measured with HIP events on that stream. */
int m0_net_bench_forward(m0_net* net, int B, int iters, int with_ssl, float* ms_per_forward);

#ifdef __cplusplus
}
#endif
#endif /* M0_ENGINE_H */
