/* libm0engine — C ABI of the MI355X-native Matrix0 self-play hot path.
 *
 * Drop-in boundary for two duck-typed seams of the reference (lukifer23/Matrix0):
 *
 *   1. inference backend  obj.infer_np(np.float32[B,19,8,8]) -> (np.float32[B,4672], np.float32[B])
 *        azchess/mcts.py:618-621, 1021-1023; azchess/selfplay/inference.py:585-645 (InferenceClient.infer_np)
 *        -> m0_net_infer
 *   2. worker entry       selfplay_worker(proc_id, cfg_dict, ckpt_path, games, q, shared_memory_resource)
 *        azchess/selfplay/internal.py:94-95 (called from orchestrator.py:494, selfplay/__main__.py:68)
 *        -> m0_selfplay_create / m0_selfplay_step / m0_selfplay_poll
 *
 *   plus the pure functions of azchess/encoding.py, exposed position-wise for parity tests and
 *   for callers that keep python-chess:  m0_encode_fens, m0_move_to_index_fen.
 *
 * Conventions: every function returns 0 on success or a negative code; the message is
 * available from m0_last_error() (thread-local).  No exception crosses the ABI.  All buffers
 * are caller-allocated host memory unless a parameter says "_dev".  Handles are opaque.
 * The library never falls back to a CPU path: without a HIP device m0_net_create fails.
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
/* hipGetDeviceCount: MI355X visible to this process (0 when there is none; never an error). */
int m0_device_count(void);

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
/* Roofline instrumentation of the dominant kernel (3x3 C->C implicit-GEMM conv): when enabled, every launch is
 * bracketed by HIP events on the launch stream.  get: accumulated milliseconds, algorithmic FLOP
 * (2 * rows * Cout * Cin * 9 per launch) and launches since the last reset. */
int m0_net_profile_enable(m0_net* net, int on);
int m0_net_profile_get(m0_net* net, double* conv_ms, double* conv_flop, int64_t* launches, int reset);
/* Of those, the milliseconds and launches of the convs that also carry a fused block tail (conv2 of every residual
 * block, the interaction conv); call before a resetting m0_net_profile_get. */
int m0_net_profile_get_tail(m0_net* net, double* tail_ms, int64_t* tail_launches);


/* ---- optional weight broadcast over RCCL / xGMI (the reference has no counterpart: its workers each read the
 * checkpoint, selfplay/internal.py:150-190).  One process per GPU.  Rank 0 calls m0_dist_unique_id and ships the 128 bytes to the
 * other ranks by any out-of-band means (a file, a socket, MPI); every rank calls m0_dist_create (collective: it returns when all
 * `world` ranks have joined).  m0_net_broadcast_weights (collective) overwrites every packed device buffer of a FINALIZED network
 * with the root's: the other ranks finalize a network of the same configuration first, with any weights of the right shapes;
 * networks of different configurations are refused before anything is sent.  librccl is loaded at the first call; without it the
 * functions fail with M0_ERR_UNSUPPORTED / NULL.  (matrix0_amd/dist.py does the same through torch.distributed.) */
typedef struct m0_dist m0_dist;
int m0_dist_unique_id(void* id128);
m0_dist* m0_dist_create(int rank, int world, const void* id128, int hip_device);
void m0_dist_destroy(m0_dist* d);
int m0_dist_rank(const m0_dist* d);
int m0_dist_world(const m0_dist* d);
int m0_net_broadcast_weights(m0_net* net, m0_dist* d, int root);

/* ---- position-wise azchess/encoding.py on the device (batched) ----
 * encode_board (encoding.py:11-46), MoveEncoder.get_legal_actions (:231-243), move_to_index (:80-150).
 * fens: n NUL-terminated strings.  Outputs nullable:
 *   planes f32 [n,19,8,8]; mask u8 [n,4672]; nlegal i32 [n];
 *   moves u16 [n,256] (from | to<<6 | promo<<12, promo 1..4 = N,B,R,Q) in legal_moves order; idx i32 [n,256]. */
int m0_encode_fens(int hip_device, const char* const* fens, int n, float* planes, uint8_t* mask, int32_t* nlegal,
                   uint16_t* moves, int32_t* idx);
/* The network's own input image of the same positions: fp16 bits, NHWC [n][64 squares][32 channels] (19 used, square
 * n = row*8+col of the reference tensor), written by the device function the search's select kernel calls for every
 * leaf (csrc/tree.hip encode_nhwc) -- i.e. encode_board (encoding.py:11-46) as the timed path evaluates it. */
int m0_encode_fens_nhwc(int hip_device, const char* const* fens, int n, uint16_t* nhwc);
/* MoveEncoder.decode_move (encoding.py:174-229): policy index -> UCI (auto-queen, legal fallbacks); "0000" = null move.
 * uci_out: at least 6 bytes. */
int m0_decode_move_fen(int hip_device, const char* fen, int action_idx, char* uci_out);
/* ChessSSLAlgorithms.create_enhanced_ssl_targets (azchess/ssl_algorithms.py:519-543) on the device:
 * out f32 [n,17,8,8] = piece one-hot (13) | threat | pin | fork | control, tensor orientation. */
int m0_ssl_targets_fens(int hip_device, const char* const* fens, int n, float* out);
/* move_to_index for one (fen, uci): raises M0_ERR_INVALID for an illegal move (ValueError in the reference). */
int m0_move_to_index_fen(int hip_device, const char* fen, const char* uci, int32_t* idx);

/* ---- search + self-play (seam 2) ---- */
/* MCTSConfig (azchess/mcts.py:61-107) + the selfplay/draw keys of config.yaml the worker reads
 * (azchess/selfplay/internal.py:269-310, 348-381; azchess/draw.py). */
typedef struct m0_selfplay_cfg {
    /* mcts */
    int num_simulations;          /* selfplay.num_simulations overrides mcts.num_simulations (internal.py:291) */
    double cpuct, cpuct_start, cpuct_end;
    int cpuct_plies;              /* <=0 or missing start/end: constant cpuct */
    int use_c_base; double cpuct_c_base, cpuct_c_init;
    double dirichlet_alpha, dirichlet_frac;
    int dirichlet_plies;          /* <0: always */
    double selection_jitter, fpu_reduction, draw_penalty, virtual_loss;
    int legal_softmax, enable_entropy_noise, no_instant_backtrack, value_from_white;
    int inference_batch_size;     /* leaves collected per tree and step (<= 96 in the reference) */
    double playout_random_frac;
    /* selfplay */
    int max_game_len, min_resign_plies, opening_random_plies;
    double resign_threshold; int resign_window, resign_consecutive_bad; double resign_min_entropy, resign_value_margin;
    double temperature_start, temperature_end; int temperature_moves;
    int low_visit_threshold;
    /* draw adjudication (draw.py) */
    int draw_enabled, draw_min_plies, draw_window, draw_min_unique, draw_halfmove_cap, draw_material_threshold, draw_stalemate;
    /* engine */
    int concurrent_games;         /* trees resident on this GPU */
    int total_games;              /* games to play (<=0: unbounded, slots restart forever) */
    int first_game_index;         /* global index of this engine's first game (multi-GPU sharding) */
    int arena_nodes;              /* nodes per arena half and game (0 = default) */
    uint64_t seed;                /* cfg["seed"] (1234) */
    int virtual_loss_active;      /* 1 = apply mcts.py:889-890/922-923 as written (the reference never does) */
    int ssl_in_forward;           /* run the SSL heads in every leaf evaluation (BASELINE config 4) */
    int ssl_targets;              /* generate ssl_* target maps for every recorded ply (selfplay/internal.py:460-482) */
    int record_games;             /* keep s/pi/legal_mask per ply for m0_selfplay_poll */
    /* evaluation matches (azchess/arena.py:59-126), m0_arena_create only */
    int arena_mode;               /* set by m0_arena_create */
    double arena_temp;            /* move choice: softmax(log(visits+1e-8)/temp) for the first arena_temp_plies plies ... */
    int arena_temp_plies;         /* ... then (or with temp <= 1e-3) the most visited move, first maximum in move order */
    /* compatibility switches (reference behaviours the default engine deviates from; 0 = engine default) */
    int fresh_tree_per_move;      /* 1 = every move starts from a brand-new root (what the reference does with MCTS._tt_get
                                     patched out, the mode tests/golden/ref_worker_*.npz were played in); 0 = keep the played
                                     child's subtree */
    int tt_merge;                 /* 1 = transposition merging inside a search as mcts.py:919 + :1330-1346 (search graph is a DAG).
                                     Self-play: the table lives for one search.  Match engine (m0_arena_create*): one table per
                                     side for the WHOLE game, roots looked up in it, as the reference's per-side MCTS objects do
                                     (arena.py:157-158); arena_nodes must then hold all nodes a side creates in a game */
    int raw_legal_priors;         /* 1 = Node._expand_with_legal_priors (mcts.py:227-256): non-root priors = legal logits / their sum */
    int max_children;             /* MCTS._prune_children (mcts.py:806-826): keep the top-K children by prior; 0 = off */
    double min_child_prior;       /* ... after dropping children with prior < this; 0 = off */
    int root_reinfer;             /* 1 = re-evaluate a reused root as mcts.py:359-371 does (nn_cache of 10 000 positions) */
    /* per-game evaluation cache (csrc/tree.h EvalCache): a leaf whose position was evaluated before -- a transposition inside
     * the search, a position of a discarded subtree -- is expanded from the stored value + legal logits instead of going
     * through the network again.  Games are unchanged (the forward is bitwise batch invariant on the 320-wide path).  Active
     * only with legal_softmax = 1 and without tt_merge / raw_legal_priors, and never in a match engine (m0_arena_create*:
     * two networks alternate in one game slot and the key carries no network id -- the field is ignored there).  A hit must
     * also match the stored legal-move count and a checksum of the legal moves; otherwise it is served as a miss.  0 = off. */
    int eval_cache;
    int eval_cache_entries;       /* entries per game (rounded up to a power of two, 4-way sets); 0 = 16384 */
    /* 1 = a pass of >= 2048 rows on the 320-wide network is evaluated as a main part that is a whole number of rounds of
     * workgroups (a multiple of 1024 boards) plus a tail (< 1024 boards) on a second instance over the same weights, on its own
     * stream, at the same time: the tower is board-local, and the tail's workgroups run on the CUs the main launches' partial last
     * round would leave idle.  Results are unchanged (the forward is bitwise batch invariant).  Self-play engines only.
     * 2 = a pass of >= 4096 rows as two halves (the first a multiple of 1024 boards) on the two instances side by side: one
     * half's attention blocks then run beside the other half's power-bound convs instead of behind them (+1.3 % games/s
     * measured); per-launch kernel timings of the two halves overlap.  0 = off (default). */
    int tail_split;
} m0_selfplay_cfg;

typedef struct m0_selfplay m0_selfplay;

typedef struct m0_selfplay_stats {
    uint64_t steps, evals, sims, plies, games_finished, games_started;
    double ms_total, ms_net, ms_tree, ms_host;   /* accumulated wall (host) and device (HIP events) times */
    uint64_t arena_overflows;
    uint64_t ssl_dropped;         /* finished games emitted WITHOUT ssl_* targets because their staging buffers could not grow */
    uint64_t evals_cached;        /* leaf evaluations served by the evaluation cache (not counted in `evals`) */
    int active_games;
    uint64_t rows_tail;           /* of `evals`: rows evaluated by the tail instance (cfg.tail_split) */
} m0_selfplay_stats;

/* One finished game = one NPZ shard of the reference (selfplay/internal.py:628-651). Arrays stay valid
 * until m0_game_record_free. */
typedef struct m0_game_record {
    int game_index, moves, resigned, resigner /*0 none 1 W 2 B*/, draw, total_plies /*incl. opening*/;
    float result;                 /* z, White POV */
    float avg_policy_entropy, avg_sims;
    double secs;
    const float* s;               /* [T,19,8,8] */
    const float* pi;              /* [T,4672] */
    const float* z;               /* [T] */
    const uint8_t* legal_mask;    /* [T,4672] */
    const float* search_values;   /* [T] root_q per ply */
    const uint16_t* played;       /* [total_plies] moves incl. opening plies */
    const float* ssl;             /* [T,17,8,8] piece(13) threat pin fork control, or NULL */
    void* owner;
} m0_game_record;

m0_selfplay* m0_selfplay_create(m0_net* net, const m0_selfplay_cfg* cfg);
void m0_selfplay_destroy(m0_selfplay* sp);
/* Run `steps` search steps (select -> network -> expand/backup over all resident games), playing moves,
 * finishing and restarting games as searches complete. */
int m0_selfplay_step(m0_selfplay* sp, int steps);
int m0_selfplay_stats_get(m0_selfplay* sp, m0_selfplay_stats* out);
/* Pop one finished game; returns 1 if a record was written, 0 if none pending, <0 on error. */
int m0_selfplay_poll(m0_selfplay* sp, m0_game_record* out);
void m0_game_record_free(m0_game_record* rec);
/* 1 while games remain to be played or are in flight. */
int m0_selfplay_running(m0_selfplay* sp);
/* Opening book (selfplay/internal.py:34-69 load_opening_book / get_opening_position): every new game starts from one of
 * these positions, chosen with the game's own stream as random.choice(OPENING_BOOK) would; n = 0 clears the book
 * (games start from the initial position).  Call before the first step. */
int m0_selfplay_set_openings(m0_selfplay* sp, const char* const* fens, int n);
/* The self-play step split at the network, for an external evaluator behind the reference's infer_np seam (net may be
 * NULL): ext_select runs select for all resident games and returns the leaf planes f32 [rows,19,8,8]; ext_expand takes
 * logits f32 [rows,4672] and values f32 [rows], expands / backs up, and does the host part of the step (moves, game
 * ends, restarts) exactly as m0_selfplay_step does.  Parity tests play whole games against golden files this way. */
/* max_rows must be >= concurrent_games * (inference_batch_size + 1) (checked before anything runs). */
int m0_selfplay_ext_select(m0_selfplay* sp, int* rows, float* planes, int max_rows);
int m0_selfplay_ext_expand(m0_selfplay* sp, const float* logits, const float* values, int rows);
/* The network batch the last select (m0_selfplay_ext_select / m0_search_select) wrote on the device: fp16 bits
 * [rows][64][32], row order = the planes rows of that call.  This is what m0_selfplay_step feeds the network. */
int m0_selfplay_last_batch_nhwc(m0_selfplay* sp, uint16_t* nhwc, int max_rows, int* rows);

/* Evaluation match between two networks (azchess/arena.py:59-126 _arena_run_one_game, :305 play_match): game i is played
 * with net_a as White when i is even.  Every search of a game is evaluated by the network of the side to move; each
 * move starts a fresh tree, unless cfg->tt_merge asks for the reference's own structure: one transposition table per side
 * kept for the whole game (arena.py:157-158 keeps one MCTS object per side), roots looked up in it.  The game ends on board.is_game_over(claim_draw=True), on
 * cfg->max_game_len plies or on draw adjudication (draw.py); no resignation.  Step / poll / stats / destroy with the
 * m0_selfplay_* functions; a record's `played` holds the moves, `result` the outcome from White's point of view
 * (0 for unfinished or adjudicated games, as the reference scores them 1/2-1/2). */
m0_selfplay* m0_arena_create(m0_net* net_a, m0_net* net_b, const m0_selfplay_cfg* cfg);
/* The same match engine without networks, for external evaluators behind the reference's infer_np seam (golden tests replay
 * the reference's arena games this way): m0_arena_ext_select returns the leaves of network A's searches and of network B's
 * separately (planes f32 [rows,19,8,8] each; max_rows >= concurrent_games * (inference_batch_size + 1) for both buffers),
 * m0_arena_ext_expand takes the two evaluators' answers and finishes the step exactly as m0_selfplay_step does. */
m0_selfplay* m0_arena_create_ext(const m0_selfplay_cfg* cfg);
int m0_arena_ext_select(m0_selfplay* sp, int* rows_a, int* rows_b, float* planes_a, float* planes_b, int max_rows);
int m0_arena_ext_expand(m0_selfplay* sp, const float* logits_a, const float* values_a, int rows_a, const float* logits_b,
                        const float* values_b, int rows_b);
/* PGN output of arena games (arena.py:281-303 uses chess.pgn): standard algebraic notation, python-chess Board.san().
 * m0_san_legal_fen: the legal moves of `fen` in legal_moves order (moves u16[256]) with their SAN (san char[256][8],
 * NUL-padded).  m0_san_game: movetext "1. e4 e5 2. Nf3 ..." of a game from the start position (moves as in
 * m0_game_record.played); returns its length. */
/* arena.py:73-106 move choice over a visit list in move order; u = the uniform np.random.choice would draw. */
int m0_arena_choose_move(const int32_t* visits, int n, double temp, int ply, int temp_plies, double u);
int m0_san_legal_fen(const char* fen, uint16_t* moves, char* san, int* nlegal);
int m0_san_game(const uint16_t* moves, int n, char* out, int cap);
/* Board.fen() after pushing `n` legal moves (UCI) on the position `fen` (python-chess semantics: cleaned castling rights, the
 * en-passant square only when such a capture is legal); M0_ERR_INVALID for an illegal move.  Host function (no GPU): the PGN
 * opening-book reader (selfplay/internal.py:39-63) and FEN-addressed callers are built on it. */
int m0_fen_after(const char* fen, const char* const* ucis, int n, char* fen_out, int cap);

/* ---- split-step search (external evaluator / parity tests): net may be NULL ----
 * m0_search_begin: reset slot g to `fen` (history-less), sims simulations, optional Dirichlet.
 * m0_search_select: run select for all active slots; returns rows; planes f32 [rows,19,8,8] of the leaves.
 * m0_search_expand: feed logits f32 [rows,4672] and values f32 [rows]; expand + backup.
 * m0_search_result: visits/moves/indices/priors/q of the root children after the search. */
int m0_search_begin(m0_selfplay* sp, int g, const char* fen, int sims, int dirichlet, int game_uid);
int m0_search_select(m0_selfplay* sp, int* rows, float* planes, int max_rows);
int m0_search_expand(m0_selfplay* sp, const float* logits, const float* values, int rows);
int m0_search_result(m0_selfplay* sp, int g, int* nchild, int32_t* child_n, uint16_t* child_mv, int32_t* child_idx,
                     double* child_prior, double* child_q, double* root_q, int* root_n, int* finished);
/* play child slot `slot` of the finished search in g and keep its subtree (tree reuse across moves) */
int m0_search_advance(m0_selfplay* sp, int g, int slot, int sims, int dirichlet);

/* ---- host decision functions (selfplay/internal.py), exposed for parity tests ---- */
int m0_sample_move_index(const int32_t* visits, int n, double temperature, double u);
int m0_playout_cap(int sims, double frac, double u);
double m0_temperature_for(int fullmove_number, double t_start, double t_end, int t_moves);
/* position + move list -> flags: bit0 game_over, bit1 game_over(claim_draw), bit2 adjudicate_draw(cfg of sp),
 * bit3 checkmate, bit4 stalemate, bit5 insufficient, bit6 can_claim_fifty, bit7 is_repetition(3),
 * bit8 can_claim_threefold, bit9 fivefold, bit10 seventyfive ; result = game_result(board) */
int m0_rules_probe(const m0_selfplay_cfg* cfg, const char* fen, const char* const* ucis, int n, int* flags, float* result);

#ifdef __cplusplus
}
#endif
#endif /* M0_ENGINE_H */
