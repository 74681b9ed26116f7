// Device-resident search trees: one arena pair per concurrent game, SoA node arrays so a
// node's children (contiguous block) load coalesced, one child per lane.
// Mirrors azchess/mcts.py Node (120-133) / _select (851-925) / _expand (135-225) /
// _backpropagate (946-953) / _add_dirichlet (955-992).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "chess_core.h"

#define M0_MAX_DEPTH 256      // path length cap (nodes)
#define M0_HIST_CAP 192       // reversible-move history window (<= 150 by the 75-move rule)
#define M0_MAX_CHILDREN 256

struct TreeCfg {              // MCTSConfig fields the kernels read (mcts.py:61-107)
    double fpu_reduction;
    double draw_penalty;
    double virtual_loss;
    double selection_jitter;
    double cpuct;
    double cpuct_start, cpuct_end;
    int cpuct_plies;          // <=0: constant cpuct
    int use_c_base;
    double cpuct_c_base, cpuct_c_init;
    double dirichlet_alpha, dirichlet_frac;
    int legal_softmax;
    int enable_entropy_noise;
    int no_instant_backtrack;
    int virtual_loss_active;  // 1: apply the in-flight penalty as written (mcts.py:889-890, 922-923)
    int leaves_per_step;      // L = inference_batch_size per tree
    // compatibility switches (include/m0_engine.h, m0_selfplay_cfg)
    int tt_merge;             // mcts.py:919 + 1330-1346: children registered in a position table, select follows the table
    int raw_legal_priors;     // mcts.py:227-256 for non-root expansions
    int max_children;         // mcts.py:806-826
    double min_child_prior;
    int eval_cache;           // 1: leaf evaluations are looked up in / stored to the game's evaluation cache (EvalCache) and a leaf
                              //    reached twice in one pass shares one batch row
};

// Per-game evaluation cache: what the network said about a position (value + the logits of its legal moves, in generation
// order), keyed by everything the network input and the expansion depend on (pieces, turn, castling rights, legal
// en-passant square, the two move counters as the planes encode them).  A repeated position -- a transposition inside the
// search, a position whose node was dropped with a discarded subtree -- is expanded from the cache instead of costing a
// network evaluation.  Same role as the reference's position table, which never evaluates a transposed node twice
// (mcts.py:919), and its nn_cache (mcts.py:44-59); results are unchanged because the 320-wide forward is bitwise batch
// invariant (a fresh evaluation would return the very same numbers).  4-way set associative, least recently used way replaced.
#define M0_EC_MAXLEGAL 64     // positions with more legal moves are not cached
#define M0_EC_WORDS 66        // payload floats per entry: value, nlegal (as int bits), 64 logits
struct EvalCache {
    uint64_t* keys;           // [G][sets * 4], 0 = empty
    uint32_t* stamps;         // [G][sets * 4] last use (per-game clock)
    float* payload;           // [G][sets * 4][M0_EC_WORDS]
    float* hit_stage;         // [G][L + 1][M0_EC_WORDS] payload of this pass's hits (an insert of the same pass may evict the entry)
    int sets;                 // per game, a power of two; 0 = cache off
};

// Per-game control block (host writes between steps, kernels update counters).
struct GameDev {
    m0::Pos root_pos;
    int root;                 // node index of the root (arena-relative)
    int next;                 // bump allocator
    int arena;                // 0/1: which arena half holds the tree
    int active;               // searching
    int sims_done, sims_target;
    int need_dirichlet;       // apply Dirichlet noise at the next select (run(): mcts.py:374-376)
    int root_q_from_v;        // expanding a reused-but-unexpanded root sets root.q = v (mcts.py:412-413)
    int flip_root_v;          // value_from_white && black to move (mcts.py:1182-1188, root eval only)
    int root_fresh;           // root is a brand-new Node (mcts.py:344-358) rather than a reused child
    int hist_len;
    int nsamples;             // samples produced by the last select
    int overflow;             // arena exhausted at least once
    int finished;             // sims_done >= sims_target after the last expand
    int root_n;
    double root_q;
    double root_v;            // network value of the last root evaluation
    uint64_t seed_jitter, seed_noise, seed_dir;
    uint64_t ctr_jitter, ctr_noise, ctr_dir;
    uint64_t evals;           // network evaluations consumed by this game (counted by the engine)
    int net_id;               // arena: which network evaluates this game's current search (0 / 1); self-play: 0
    int reinfer;              // evaluate the (reused) root once more at the first select of this search (mcts.py:359-371)
    // match engine with compat.tt_merge (TreeDev::tt_sides == 2): arena half s and table s belong to side s for the WHOLE game
    uint64_t cache_hits;      // leaf evaluations served by the evaluation cache
    uint32_t cache_clock;
    int side_next[2];         // bump allocator of each side's half while the other side searches
    int root_found;           // advance_kernel: this search's root was found in the side's table (mcts.py:343, 359-371)
};

struct Sample {
    m0::Pos pos;              // leaf position
    int kind;                 // 0 none, 1 eval+backup, 2 root init (expand only), 3 terminal (already backed up),
                              // 4 root value only (re-evaluation of a reused root), 5 eval served by the evaluation cache,
                              // 6 the same leaf as an earlier sample of this pass (shares its batch row)
    uint64_t ckey;            // evaluation-cache key of the leaf (0: not cacheable / cache off)
    int leaf;
    int depth;                // path has depth+1 nodes
    int row;                  // network batch row
    int nlegal;               // legal moves of the leaf (list in TreeDev::leaf_moves), filled by select
};

struct TreeArrays {           // each [G][2*cap]
    double* prior;
    double* w;
    double* q;
    int* n;
    int* vl;
    int* cbase;
    int16_t* nch;             // -1 not expanded
    uint16_t* mv;
    uint16_t* midx;
    int cap;                  // nodes per arena half
};

struct RootResult {           // written when a search finishes
    int nchild;
    int root_n;
    double root_q;
    int child_node[M0_MAX_CHILDREN];
    int child_n[M0_MAX_CHILDREN];
    uint16_t child_mv[M0_MAX_CHILDREN];
    uint16_t child_idx[M0_MAX_CHILDREN];
    double child_prior[M0_MAX_CHILDREN];
    double child_q[M0_MAX_CHILDREN];
};

struct TreeDev {
    TreeArrays t;
    GameDev* games;           // [G]
    Sample* samples;          // [G][L+1]
    int* paths;               // [G][L+1][M0_MAX_DEPTH]
    int* epaths;              // tt_merge: [G][L+1][M0_MAX_DEPTH] the edge children chosen at each level (virtual-loss owners)
    uint64_t* tt_keys;        // tt_merge: [G][tt_cap] position keys, 0 = empty (open addressing, linear probing)
    int* tt_nodes;            // tt_merge: [G][tt_cap] node registered LAST under the key
    int tt_cap;               // entries per table, a power of two
    EvalCache ec;
    int search_nodes;         // head-room a per-side half (tt_sides == 2) must have before a search starts there: the nodes one
                              // search can create (~48 per simulation incl. margin); with less the half starts over first
    int tt_sides;             // tables per game: 1, or 2 in a match engine with compat.tt_merge (one per side, kept all game:
                              // the reference keeps one MCTS object, hence one table, per side -- arena.py:157-158)
    uint16_t* leaf_moves;     // [G][L+1][M0_MAX_CHILDREN] legal moves of each sampled leaf, generation order
    uint64_t* hist;           // [G][M0_HIST_CAP]
    RootResult* results;      // [G]
    int* row_counter;         // [2]: rows reserved for network 0 / network 1 (arena); row index = net_row_base*net + count
    int net_row_base;         // first batch row of network 1's region (self-play: unused)
    _Float16* x0;             // network input NHWC [rows][64][32]
    const float* logits;      // [rows][4672]
    const float* values;      // [rows]
    int G;
    int L;
};

hipError_t launch_select(const TreeDev& d, const TreeCfg& c, hipStream_t st);
hipError_t launch_expand(const TreeDev& d, const TreeCfg& c, hipStream_t st);
// child_slots: >= 0 keep that child's subtree; -1 fresh tree (and an empty table); tt_sides == 2 only: -2 next search of a
// game (root looked up in the side's table), -3 first search of a game (both tables cleared first)
hipError_t launch_advance(const TreeDev& d, const int* game_ids_dev, const int* child_slots_dev, int count, hipStream_t st);

// test hooks: position-wise encode / legal move / index kernels (encoding.py on device)
struct PosQuery {
    m0::Pos pos;
};
hipError_t launch_encode_positions(const m0::Pos* pos_dev, int n, float* planes_f32_dev /*[n][19][64]*/,
                                   _Float16* nhwc_dev /*[n][64][32] or null*/, uint8_t* mask_dev /*[n][4672]*/,
                                   int32_t* nlegal_dev, uint16_t* moves_dev /*[n][256]*/, int32_t* idx_dev /*[n][256]*/,
                                   hipStream_t st);

// SSL training targets for recorded positions: out f32 [n][17][64] (piece 13, threat, pin, fork, control)
hipError_t launch_ssl_targets(const m0::Pos* pos_dev, int n, float* out_dev, hipStream_t st);
