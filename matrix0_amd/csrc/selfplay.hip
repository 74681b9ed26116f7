// Host engine: drives the device trees and the network, plays the games.
// Mirrors the per-game loop of selfplay_worker (azchess/selfplay/internal.py:326-679) for many
// concurrent games: opening plies, draw adjudication, temperature, MCTS.run bookkeeping
// (mcts.py:318-512), move sampling, resign logic, result and record assembly.
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <deque>
#include <list>
#include <mutex>
#include <unordered_map>
#include <string>
#include <vector>
#include "../../include/m0_engine.h"
#include "capi_common.h"
#include "chess_core.h"
#include "host_rules.h"
#include "net.h"
#include "tree.h"

using namespace m0;

namespace {

inline double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct GameRecordOwner {
    std::vector<float> s, pi, z, search_values, ssl;
    std::vector<uint8_t> legal_mask;
    std::vector<uint16_t> played;
};

struct HostGame {
    bool in_use = false;
    int game_index = 0;
    Pos pos;
    RepWindow win;
    std::vector<Move> history;            // all moves incl. opening plies
    std::vector<Pos> rec_pos;             // position of every recorded ply (SSL targets)
    // records
    std::vector<float> states, pis, search_values;
    std::vector<int8_t> turns;
    std::vector<uint8_t> masks;
    std::vector<int> sims_used;
    int nstates = 0;
    double entropy_sum = 0.0;
    int entropy_count = 0;
    ResignState resign;
    HStream rng;                          // PURPOSE_GAME stream: opening plies, playout cap, move sampling
    double t0 = 0.0;
    int cur_sims = 0;
    bool a_is_white = true;               // arena
};

}  // namespace

struct m0_selfplay {
    m0_selfplay_cfg cfg;
    TreeCfg tc;
    m0_net* nethandle = nullptr;
    m0_net* nethandle_b = nullptr;
    Net* net = nullptr;
    Net* net_b = nullptr;                 // arena: the second network (games with an odd index play it as White)
    Net* net_tail = nullptr;              // cfg.tail_split: a view of `net` (same weights, own stream + workspace) for the partial last round
    hipStream_t stream_tail = nullptr;
    hipEvent_t ev_sel = nullptr, ev_tail = nullptr;
    bool half_split = false;              // cfg.tail_split == 2: two halves instead of main + tail
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int G = 0, L = 0, cap = 0, rows_max = 0;
    TreeDev d;
    std::vector<void*> allocs;
    std::vector<GameDev> hg;
    std::vector<HostGame> games;
    std::vector<RootResult> hres;
    float* logits_dev = nullptr;
    float* values_dev = nullptr;
    float* ssl_dev = nullptr;
    Pos* ssl_pos_dev = nullptr;            // staging of one finished game's positions / SSL target maps (ssl_targets)
    float* ssl_out_dev = nullptr;
    int ssl_cap = 0;
    int* ids_dev = nullptr;
    int* slots_dev = nullptr;
    std::deque<GameRecordOwner*> done_records;
    std::deque<m0_game_record> done_meta;
    m0_selfplay_stats stats;
    int next_game = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
    std::mutex mu;
    std::vector<Sample> hsamples;
    std::vector<int> prev_done;           // per slot: simulations already credited to stats.sims
    int last_rows = 0, last_rows_b = 0;
    int rows2[2] = {0, 0};                // rows of the last select per network
    std::vector<Pos> book;                // opening positions (m0_selfplay_set_openings)
    // LRUCache nn_cache of the reference (mcts.py:44-59, 303, 360-371): positions whose root was re-evaluated; 10 000 entries
    std::list<uint64_t> nn_lru;
    std::unordered_map<uint64_t, std::list<uint64_t>::iterator> nn_map;
    bool ext_pending = false;             // ext_select done, ext_expand outstanding
    bool counted = false;                 // registered in g_engines_with_net (forward gate)
};

namespace {

template <typename T>
T* dalloc(m0_selfplay* sp, size_t count) {
    void* p = nullptr;
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = 16;
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    (void)hipMemsetAsync(p, 0, bytes, sp->stream);          // on the engine's own stream: ordered before every kernel that uses it
    sp->allocs.push_back(p);
    return (T*)p;
}

void fill_tree_cfg(const m0_selfplay_cfg& c, TreeCfg& t) {
    t.fpu_reduction = c.fpu_reduction; t.draw_penalty = c.draw_penalty; t.virtual_loss = c.virtual_loss;
    t.selection_jitter = c.selection_jitter; t.cpuct = c.cpuct; t.cpuct_start = c.cpuct_start; t.cpuct_end = c.cpuct_end;
    t.cpuct_plies = c.cpuct_plies; t.use_c_base = c.use_c_base; t.cpuct_c_base = c.cpuct_c_base; t.cpuct_c_init = c.cpuct_c_init;
    t.dirichlet_alpha = c.dirichlet_alpha; t.dirichlet_frac = c.dirichlet_frac; t.legal_softmax = c.legal_softmax;
    t.enable_entropy_noise = c.enable_entropy_noise; t.no_instant_backtrack = c.no_instant_backtrack;
    t.virtual_loss_active = c.virtual_loss_active; t.leaves_per_step = c.inference_batch_size;
    t.tt_merge = c.tt_merge; t.raw_legal_priors = c.raw_legal_priors; t.max_children = c.max_children;
    t.min_child_prior = c.min_child_prior;
    // the cached payload is the LEGAL logits: only the legal-softmax expansion can be served from it
    t.eval_cache = (c.eval_cache && c.legal_softmax && !c.raw_legal_priors && !c.tt_merge) ? 1 : 0;
}

// mcts.py:359-371: a root that run() finds already in its table is evaluated again unless the position sits in nn_cache
// (which only this branch fills).  Returns true when the evaluation has to be made; updates the LRU either way.
bool nn_cache_miss(m0_selfplay* sp, uint64_t key) {
    auto it = sp->nn_map.find(key);
    if (it != sp->nn_map.end()) {
        sp->nn_lru.splice(sp->nn_lru.end(), sp->nn_lru, it->second);
        return false;
    }
    sp->nn_lru.push_back(key);
    sp->nn_map[key] = std::prev(sp->nn_lru.end());
    if (sp->nn_lru.size() > 10000) { sp->nn_map.erase(sp->nn_lru.front()); sp->nn_lru.pop_front(); }
    return true;
}

void seed_game_dev(GameDev& g, uint64_t base, int uid) {
    g.seed_jitter = derive_seed(base, uid, PURPOSE_JITTER);
    g.seed_noise = derive_seed(base, uid, PURPOSE_NOISE);
    g.seed_dir = derive_seed(base, uid, PURPOSE_DIRICHLET);
    g.ctr_jitter = g.ctr_noise = g.ctr_dir = 0;
}

int sync_games_d2h(m0_selfplay* sp) {
    if (hipMemcpyAsync(sp->hg.data(), sp->d.games, sizeof(GameDev) * sp->G, hipMemcpyDeviceToHost, sp->stream) != hipSuccess) return -1;
    return hipStreamSynchronize(sp->stream) == hipSuccess ? 0 : -1;
}
int sync_games_h2d(m0_selfplay* sp) {
    return hipMemcpyAsync(sp->d.games, sp->hg.data(), sizeof(GameDev) * sp->G, hipMemcpyHostToDevice, sp->stream) == hipSuccess ? 0 : -1;
}
void push_hist(m0_selfplay* sp, int slot, const RepWindow& w) {
    int n = (int)w.keys.size();
    const uint64_t* src = w.keys.data();
    if (n > M0_HIST_CAP) { src += n - M0_HIST_CAP; n = M0_HIST_CAP; }
    sp->hg[slot].hist_len = n;
    if (n > 0) (void)hipMemcpyAsync(sp->d.hist + (size_t)slot * M0_HIST_CAP, src, (size_t)n * 8, hipMemcpyHostToDevice, sp->stream);
}

// configure a search on slot for the host position (MCTS.run prologue, mcts.py:342-396)
void arm_search(m0_selfplay* sp, int slot, const Pos& pos, const RepWindow& win, int sims, bool dirichlet, bool fresh) {
    GameDev& g = sp->hg[slot];
    g.root_pos = pos;
    g.active = 1; g.sims_done = 0; g.sims_target = sims;
    g.need_dirichlet = dirichlet ? 1 : 0;
    g.root_q_from_v = 0;
    g.root_fresh = fresh ? 1 : 0;
    g.flip_root_v = (sp->cfg.value_from_white && pos.turn == BLACK) ? 1 : 0;
    g.finished = 0; g.nsamples = 0;
    g.reinfer = (sp->cfg.root_reinfer && !fresh && nn_cache_miss(sp, tkey(pos))) ? 1 : 0;
    sp->prev_done[slot] = 0;
    push_hist(sp, slot, win);
}

void finish_game(m0_selfplay* sp, int slot, bool resigned, int resigner, bool have_z, float z_in);
void begin_move(m0_selfplay* sp, int slot, int child_slot, std::vector<int>& adv_ids, std::vector<int>& adv_slots);

void start_game(m0_selfplay* sp, int slot, std::vector<int>& adv_ids, std::vector<int>& adv_slots) {
    HostGame& hgm = sp->games[slot];
    hgm = HostGame();
    hgm.in_use = true;
    hgm.game_index = sp->cfg.first_game_index + sp->next_game++;
    sp->stats.games_started++;
    parse_fen("rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1", hgm.pos);
    hgm.rng = HStream(derive_seed(sp->cfg.seed, hgm.game_index, PURPOSE_GAME));
    if (!sp->book.empty()) {               // get_opening_position (internal.py:65-69): random.choice(OPENING_BOOK)
        size_t k = (size_t)(hgm.rng.next() * (double)sp->book.size());
        if (k >= sp->book.size()) k = sp->book.size() - 1;
        hgm.pos = sp->book[k];
    }
    hgm.t0 = now_ms();
    hgm.a_is_white = (hgm.game_index % 2) == 0;               // arena.py:66
    seed_game_dev(sp->hg[slot], sp->cfg.seed, hgm.game_index);
    sp->hg[slot].evals = 0;
    // opening diversity: uniform random legal plies (internal.py:366-379; random.choice -> injected stream)
    for (int i = 0; i < sp->cfg.opening_random_plies; ++i) {
        if (is_game_over(hgm.pos, hgm.win, false)) break;
        Move mv[M0_MAX_MOVES];
        int n = gen_legal(hgm.pos, mv);
        if (n <= 0) break;
        int k = (int)(hgm.rng.next() * n);
        if (k >= n) k = n - 1;
        hgm.win.push(hgm.pos, mv[k]);
        make_move(hgm.pos, mv[k]);
        hgm.history.push_back(mv[k]);
    }
    begin_move(sp, slot, -1, adv_ids, adv_slots);
}

// top of the per-ply loop (internal.py:382-408): termination tests, then arm the search
void begin_move(m0_selfplay* sp, int slot, int child_slot, std::vector<int>& adv_ids, std::vector<int>& adv_slots) {
    HostGame& hgm = sp->games[slot];
    const m0_selfplay_cfg& c = sp->cfg;
    DrawCfg dc = draw_cfg_from(c);
    // self-play: internal.py:382-408; arena: `while not board.is_game_over(claim_draw=True) and moves < max_moves`
    // then the adjudication test (arena.py:68-72)
    if (is_game_over(hgm.pos, hgm.win, c.arena_mode != 0) || hgm.nstates >= c.max_game_len ||
        should_adjudicate_draw(hgm.pos, hgm.win, hgm.history, dc)) {
        finish_game(sp, slot, false, 0, false, 0.f);
        return;
    }
    if (c.arena_mode) {
        child_slot = -1;                                           // a fresh tree per move (see m0_arena_create)
        sp->hg[slot].net_id = ((hgm.pos.turn == WHITE) == hgm.a_is_white) ? 0 : 1;
    }
    // mcts.py:378-387 draws random.randint only when the cap is configured
    const bool cap_draws = c.playout_random_frac > 0.0 && c.num_simulations > 0;
    int sims = playout_cap(c.num_simulations, c.playout_random_frac, cap_draws ? hgm.rng.next() : 0.0);
    hgm.cur_sims = sims;
    if (c.fresh_tree_per_move || c.tt_merge) child_slot = -1;   // tt_merge: the table lives for one search (see m0_engine.h)
    // ... except in a match engine, where each side's table lives for the whole game: -3 = first search of the game (both
    // tables cleared), -2 = any later one (root looked up in the side's table by advance_kernel)
    if (c.arena_mode && c.tt_merge) child_slot = hgm.nstates == 0 ? -3 : -2;
    const bool dir = c.dirichlet_plies < 0 || hgm.nstates < c.dirichlet_plies;
    arm_search(sp, slot, hgm.pos, hgm.win, sims, dir, child_slot < 0);
    adv_ids.push_back(slot);
    adv_slots.push_back(child_slot);
}

void finish_game(m0_selfplay* sp, int slot, bool resigned, int resigner, bool have_z, float z_in) {
    HostGame& hgm = sp->games[slot];
    const m0_selfplay_cfg& c = sp->cfg;
    float z = z_in;
    if (!have_z) {
        // internal.py:587-599 computed per game (SURVEY B-8: the reference's stale-z reuse is not reproduced)
        if (is_game_over(hgm.pos, hgm.win, true)) z = game_result(hgm.pos);
        else if (c.arena_mode) z = 0.f;                              // arena.py:121-123: unfinished = 1/2-1/2
        else z = hgm.search_values.empty() ? 0.f : hgm.search_values.back();
    }
    sp->stats.games_finished++;
    if ((c.record_games && hgm.nstates > 0) || c.arena_mode) {
        GameRecordOwner* o = new GameRecordOwner();
        o->s.swap(hgm.states); o->pi.swap(hgm.pis); o->legal_mask.swap(hgm.masks);
        o->search_values = hgm.search_values;
        o->z.resize(hgm.nstates);
        for (int i = 0; i < hgm.nstates; ++i) o->z[i] = z * (float)hgm.turns[i];
        o->played.assign(hgm.history.begin(), hgm.history.end());
        if (c.ssl_targets && (int)hgm.rec_pos.size() == hgm.nstates) {
            // targets for all plies of the game in one launch on the engine stream
            const int T = hgm.nstates;
            if (T > sp->ssl_cap) {             // a game longer than the staging buffers (max_game_len <= 0): grow them
                Pos* np = dalloc<Pos>(sp, (size_t)T + 64);
                float* no = dalloc<float>(sp, ((size_t)T + 64) * 17 * 64);
                if (np && no) { sp->ssl_pos_dev = np; sp->ssl_out_dev = no; sp->ssl_cap = T + 64; }   // the old pair is freed with the engine
                else sp->stats.ssl_dropped++;
            }
            if (T <= sp->ssl_cap) {
                o->ssl.resize((size_t)T * 17 * 64);
                (void)hipMemcpyAsync(sp->ssl_pos_dev, hgm.rec_pos.data(), sizeof(Pos) * T, hipMemcpyHostToDevice, sp->stream);
                (void)launch_ssl_targets(sp->ssl_pos_dev, T, sp->ssl_out_dev, sp->stream);
                (void)hipMemcpyAsync(o->ssl.data(), sp->ssl_out_dev, (size_t)T * 17 * 64 * 4, hipMemcpyDeviceToHost, sp->stream);
                (void)hipStreamSynchronize(sp->stream);
            }
        }
        m0_game_record r;
        memset(&r, 0, sizeof(r));
        r.game_index = hgm.game_index; r.moves = hgm.nstates; r.resigned = resigned ? 1 : 0; r.resigner = resigner;
        r.draw = z == 0.0f ? 1 : 0; r.total_plies = (int)hgm.history.size(); r.result = z;
        r.avg_policy_entropy = (float)(hgm.entropy_sum / (double)(hgm.entropy_count > 0 ? hgm.entropy_count : 1));
        double ss = 0; for (int v : hgm.sims_used) ss += v;
        r.avg_sims = hgm.sims_used.empty() ? 0.f : (float)(ss / (double)hgm.sims_used.size());
        r.secs = (now_ms() - hgm.t0) / 1000.0;
        r.s = o->s.data(); r.pi = o->pi.data(); r.z = o->z.data(); r.legal_mask = o->legal_mask.data();
        r.search_values = o->search_values.data(); r.played = o->played.data(); r.owner = o;
        r.ssl = o->ssl.empty() ? nullptr : o->ssl.data();
        sp->done_meta.push_back(r);
    }
    hgm.in_use = false;
    sp->hg[slot].active = 0;
}

// MCTS.run epilogue + the rest of the per-ply loop body (mcts.py:431-507, internal.py:408-539)
void finish_search(m0_selfplay* sp, int slot, std::vector<int>& adv_ids, std::vector<int>& adv_slots) {
    HostGame& hgm = sp->games[slot];
    const m0_selfplay_cfg& c = sp->cfg;
    const RootResult& R = sp->hres[slot];
    const GameDev& g = sp->hg[slot];
    const int k = R.nchild;
    long total = 0;
    int maxv = 0;
    for (int i = 0; i < k; ++i) { total += R.child_n[i]; if (R.child_n[i] > maxv) maxv = R.child_n[i]; }
    if (k <= 0 || total <= 0) {        // mcts.py:435-463 raises RuntimeError: the game is dropped (no record), counted
        sp->stats.arena_overflows++;   // in stats.arena_overflows, which the worker logs
        hgm.nstates = 0;               // no rows -> finish_game emits no training record
        hgm.states.clear(); hgm.pis.clear(); hgm.masks.clear(); hgm.search_values.clear(); hgm.turns.clear();
        hgm.sims_used.clear(); hgm.rec_pos.clear();
        finish_game(sp, slot, false, 0, true, 0.f);
        return;
    }
    const double root_q = R.root_n > 0 ? R.root_q : g.root_v;
    // policy target (mcts.py:828-837)
    const size_t T = (size_t)hgm.nstates;
    if (c.record_games) {
        hgm.pis.resize((T + 1) * 4672, 0.f);
        float* pi = hgm.pis.data() + T * 4672;
        for (int i = 0; i < k; ++i) pi[R.child_idx[i]] = (float)((double)R.child_n[i] / (double)total);
        hgm.states.resize((T + 1) * 19 * 64);
        encode_planes_f32(hgm.pos, hgm.states.data() + T * 19 * 64);
        if (c.ssl_targets) hgm.rec_pos.push_back(hgm.pos);
        hgm.masks.resize((T + 1) * 4672, 0);
        uint8_t* mk = hgm.masks.data() + T * 4672;
        if (c.max_children > 0 || c.min_child_prior > 0.0) {          // pruned roots: the mask is still ALL legal moves
            Move lm[M0_MAX_MOVES];
            const int nl = gen_legal(hgm.pos, lm);
            for (int i = 0; i < nl; ++i) mk[move_to_index(hgm.pos, lm[i])] = 1;
        } else {
            for (int i = 0; i < k; ++i) mk[R.child_idx[i]] = 1;
        }
    }
    // entropy of pi (internal.py:430-436)
    {
        double ent = 0.0;
        for (int i = 0; i < k; ++i) {
            double p = (double)(float)((double)R.child_n[i] / (double)total);
            if (p < 1e-12) p = 1e-12;
            ent -= p * log(p);
        }
        ent -= (double)(4672 - k) * (1e-12 * log(1e-12));
        hgm.entropy_sum += ent; hgm.entropy_count++;
        hgm.resign.recent_entropies.push_back(ent);
        if ((int)hgm.resign.recent_entropies.size() > c.resign_window) hgm.resign.recent_entropies.erase(hgm.resign.recent_entropies.begin());
    }
    // temperature + move choice (internal.py:386-394, 418-427, 690-735)
    double temp = temperature_for(hgm.pos.fullmove, c.temperature_start, c.temperature_end, c.temperature_moves);
    if (c.low_visit_threshold > 0 && maxv < c.low_visit_threshold && temp < 0.8) temp = 0.8;
    std::vector<int32_t> visits(R.child_n, R.child_n + k);
    int pick;
    if (c.arena_mode) {       // arena.py:73-106 (the uniform is drawn only on the sampling branch, as np.random.choice is)
        const bool sampling = c.arena_temp > 1e-3 && hgm.nstates < c.arena_temp_plies;
        pick = arena_choose_move(visits.data(), k, c.arena_temp, hgm.nstates, c.arena_temp_plies, sampling ? hgm.rng.next() : 0.0);
    } else {
        pick = sample_move_index(visits.data(), k, temp, sample_move_draws(visits.data(), k, temp) ? hgm.rng.next() : 0.0);
    }
    const Move mv = R.child_mv[pick];
    hgm.search_values.push_back((float)root_q);
    hgm.turns.push_back(hgm.pos.turn == WHITE ? 1 : -1);
    hgm.sims_used.push_back(hgm.cur_sims);
    hgm.nstates++;
    sp->stats.plies++;
    // resign (internal.py:507-536)
    if (!c.arena_mode && resign_update(hgm.resign, root_q, hgm.nstates, c)) {
        const bool white = hgm.pos.turn == WHITE;
        finish_game(sp, slot, true, white ? 1 : 2, true, white ? -1.f : 1.f);
        return;
    }
    hgm.win.push(hgm.pos, mv);
    make_move(hgm.pos, mv);
    hgm.history.push_back(mv);
    begin_move(sp, slot, pick, adv_ids, adv_slots);
}

int run_select(m0_selfplay* sp, int* rows_out) {
    if (hipMemsetAsync(sp->d.row_counter, 0, 8, sp->stream) != hipSuccess) return -1;
    if (launch_select(sp->d, sp->tc, sp->stream) != hipSuccess) return -1;
    if (hipMemcpyAsync(sp->rows2, sp->d.row_counter, 8, hipMemcpyDeviceToHost, sp->stream) != hipSuccess) return -1;
    if (hipStreamSynchronize(sp->stream) != hipSuccess) return -1;
    *rows_out = sp->rows2[0];              // network 0's rows; network 1's (arena) start at d.net_row_base
    return 0;
}

int apply_advances(m0_selfplay* sp, std::vector<int>& ids, std::vector<int>& slots) {
    if (sync_games_h2d(sp) != 0) return -1;
    if (!ids.empty()) {
        (void)hipMemcpyAsync(sp->ids_dev, ids.data(), ids.size() * 4, hipMemcpyHostToDevice, sp->stream);
        (void)hipMemcpyAsync(sp->slots_dev, slots.data(), slots.size() * 4, hipMemcpyHostToDevice, sp->stream);
        if (launch_advance(sp->d, sp->ids_dev, sp->slots_dev, (int)ids.size(), sp->stream) != hipSuccess) return -1;
        if (hipStreamSynchronize(sp->stream) != hipSuccess) return -1;   // ids/slots vectors die with the caller
        if (sp->d.tt_sides == 2) {
            // per-side tables: only the device knows whether a root was found in its side's table.  A found root is evaluated
            // once more unless nn_cache holds the position (mcts.py:359-371; the cache belongs to the side's MCTS object).
            if (sync_games_d2h(sp) != 0) return -1;
            bool any = false;
            for (size_t k = 0; k < ids.size(); ++k) {
                if (slots[k] > -2) continue;
                GameDev& g = sp->hg[ids[k]];
                const uint64_t salt = (g.net_id & 1) ? 0x9E3779B97F4A7C15ull : 0ull;
                const uint64_t gsalt = (uint64_t)(uint32_t)sp->games[ids[k]].game_index * 0xD6E8FEB86659FD93ull;
                g.reinfer = (g.root_found && nn_cache_miss(sp, tkey(g.root_pos) ^ salt ^ gsalt)) ? 1 : 0;
                any = any || g.reinfer;
            }
            if (any && sync_games_h2d(sp) != 0) return -1;
        }
    }
    return 0;
}

int step_back(m0_selfplay* sp, int rows, double t0, std::string& err);

// Several engines on one GPU (engine.SelfplayPool).  By default their forwards simply overlap on the chip.  With
// M0_FORWARD_GATE=1 (read once) the forwards of engines that own a network on the same device take turns instead: what the
// second engine then adds is that its tree kernels and host work run while the other engine's forward owns the chip, and a
// conv launch's event-bracketed time stays a measurement of that kernel alone.  Measured: gated 1.098
// against 1.121 games/s with one engine in round 3 -- the gate is a measurement aid, not a speed-up, hence opt-in.  It is held
// from the first launch of a forward until its last kernel has finished and is skipped by a step without rows.
constexpr int M0_MAX_DEVICES = 16;
std::mutex g_forward_gate[M0_MAX_DEVICES];
std::atomic<int> g_engines_with_net[M0_MAX_DEVICES];
bool forward_gate_enabled() {
    static const bool on = [] { const char* v = getenv("M0_FORWARD_GATE"); return v && v[0] == '1'; }();
    return on;
}

struct ForwardGate {
    std::mutex* m = nullptr;
    hipStream_t st = nullptr;
    ForwardGate(int device, hipStream_t stream, bool has_rows) : st(stream) {
        if (has_rows && forward_gate_enabled() && (unsigned)device < (unsigned)M0_MAX_DEVICES &&
            g_engines_with_net[device].load(std::memory_order_relaxed) > 1) {
            m = &g_forward_gate[device];
            m->lock();
        }
    }
    void release() {
        if (m) { (void)hipStreamSynchronize(st); m->unlock(); m = nullptr; }
    }
    ~ForwardGate() { release(); }
};

int one_step(m0_selfplay* sp, std::string& err) {
    const double t0 = now_ms();
    (void)hipEventRecord(sp->ev0, sp->stream);
    int rows = 0;
    if (run_select(sp, &rows) != 0) { err = std::string("select failed: ") + hipGetErrorString(hipGetLastError()); return M0_ERR_HIP; }
    (void)hipEventRecord(sp->ev1, sp->stream);
    if (rows > sp->rows_max || sp->rows2[1] > sp->rows_max) { err = "row counter overflow"; return M0_ERR_STATE; }
    ForwardGate gate(sp->device, sp->stream, rows > 0 || sp->rows2[1] > 0);
    if (rows > 0) {
        if (!sp->net) { err = "m0_selfplay_step needs a network (use the split-step API without one)"; return M0_ERR_STATE; }
        // tail split: the rows beyond the last whole round of workgroups (1024 boards = 256 four-board tiles) go to the second
        // instance on its own stream, behind the select kernel and in front of the expand kernel by events; the main launches then
        // have no partial last round and the tail's workgroups fill CUs as they come free
        int main_rows = rows, tail_rows = 0;
        if (sp->net_tail && rows >= 2048 && (rows & 1023) != 0) { main_rows = rows & ~1023; tail_rows = rows - main_rows; }
        // tail_split = 2: the pass as two halves (the first one a whole number of rounds) on the two streams.  Two forwards side by
        // side keep the chip's power draw even -- one half's attention blocks (latency-bound, low power) fall beside the other
        // half's convs (power-bound) instead of running behind them at the clock they leave: +1.3 % games/s
        if (sp->net_tail && sp->half_split && rows >= 4096) { main_rows = ((rows / 2) + 1023) & ~1023; tail_rows = rows - main_rows; }
        float* ssl = sp->cfg.ssl_in_forward ? sp->ssl_dev : nullptr;
        m0_net_lock(sp->nethandle);          // an infer_np on the same backend from another thread waits here
        int rc = M0_OK;
        if (tail_rows > 0) (void)hipEventRecord(sp->ev_sel, sp->stream);          // the select kernel has written the batch
        // main part first: its launches start at once, the tail's are enqueued while they run
        rc = sp->net->forward(nullptr, sp->d.x0, main_rows, sp->logits_dev, sp->values_dev, ssl, sp->stream, err);
        if (rc == M0_OK && tail_rows > 0) {
            (void)hipStreamWaitEvent(sp->stream_tail, sp->ev_sel, 0);
            const size_t sslw = ssl ? (size_t)sp->net->ssl_channels_total() * 64 : 0;
            rc = sp->net_tail->forward(nullptr, sp->d.x0 + (size_t)main_rows * 64 * 32, tail_rows, sp->logits_dev + (size_t)main_rows * 4672,
                                       sp->values_dev + main_rows, ssl ? ssl + (size_t)main_rows * sslw : nullptr, sp->stream_tail, err);
            (void)hipEventRecord(sp->ev_tail, sp->stream_tail);
            (void)hipStreamWaitEvent(sp->stream, sp->ev_tail, 0);
        }
        m0_net_unlock(sp->nethandle);
        if (rc != M0_OK) return rc;
        sp->stats.rows_tail += (uint64_t)tail_rows;
    }
    if (sp->rows2[1] > 0) {                 // arena: the other network's leaves, in their own region of the batch
        if (!sp->net_b) { err = "rows for a second network without one"; return M0_ERR_STATE; }
        const size_t b = (size_t)sp->d.net_row_base;
        m0_net_lock(sp->nethandle_b);
        int rc = sp->net_b->forward(nullptr, sp->d.x0 + b * 64 * 32, sp->rows2[1], sp->logits_dev + b * 4672,
                                    sp->values_dev + b, nullptr, sp->stream, err);
        m0_net_unlock(sp->nethandle_b);
        if (rc != M0_OK) return rc;
        rows += sp->rows2[1];
    }
    gate.release();
    return step_back(sp, rows, t0, err);
}

// second half of a step: expand / backup on the device, then the host part (finished searches -> moves, game ends, restarts)
int step_back(m0_selfplay* sp, int rows, double t0, std::string& err) {
    (void)hipEventRecord(sp->ev2, sp->stream);
    if (launch_expand(sp->d, sp->tc, sp->stream) != hipSuccess) { err = "expand launch failed"; return M0_ERR_HIP; }
    (void)hipEventRecord(sp->ev3, sp->stream);
    if (sync_games_d2h(sp) != 0) { err = std::string("step failed: ") + hipGetErrorString(hipGetLastError()); return M0_ERR_HIP; }
    if (sp->net) sp->net->harvest_profile();
    if (sp->net_b) sp->net_b->harvest_profile();
    float ms_sel = 0, ms_net = 0, ms_exp = 0;
    (void)hipEventElapsedTime(&ms_sel, sp->ev0, sp->ev1);
    (void)hipEventElapsedTime(&ms_net, sp->ev1, sp->ev2);
    (void)hipEventElapsedTime(&ms_exp, sp->ev2, sp->ev3);
    sp->stats.ms_net += ms_net; sp->stats.ms_tree += ms_sel + ms_exp;
    sp->stats.steps++; sp->stats.evals += (uint64_t)rows;
    const double t1 = now_ms();
    // host: finished searches -> moves, game ends, restarts
    bool any = false;
    if (sp->tc.eval_cache) {
        uint64_t h = 0;
        for (int s = 0; s < sp->G; ++s) h += sp->hg[s].cache_hits;
        sp->stats.evals_cached = h;
    }
    for (int s = 0; s < sp->G; ++s) {
        if (!sp->hg[s].active) continue;
        const int delta = sp->hg[s].sims_done - sp->prev_done[s];
        if (delta > 0) { sp->stats.sims += (uint64_t)delta; sp->prev_done[s] = sp->hg[s].sims_done; }
        if (sp->hg[s].finished) any = true;
    }
    std::vector<int> ids, slots;
    if (any) {
        (void)hipMemcpy(sp->hres.data(), sp->d.results, sizeof(RootResult) * sp->G, hipMemcpyDeviceToHost);
        for (int s = 0; s < sp->G; ++s) {
            if (!(sp->hg[s].active && sp->hg[s].finished)) continue;
            if (sp->hg[s].overflow) sp->stats.arena_overflows++;
            finish_search(sp, s, ids, slots);
        }
        // refill free slots
        for (int s = 0; s < sp->G; ++s) {
            if (sp->games[s].in_use) continue;
            if (sp->cfg.total_games > 0 && sp->next_game >= sp->cfg.total_games) continue;
            start_game(sp, s, ids, slots);
        }
        if (apply_advances(sp, ids, slots) != 0) { err = "advance failed"; return M0_ERR_HIP; }
    }
    int act = 0;
    for (int s = 0; s < sp->G; ++s) act += sp->games[s].in_use ? 1 : 0;
    sp->stats.active_games = act;
    const double t2 = now_ms();
    sp->stats.ms_host += t2 - t1;
    sp->stats.ms_total += t2 - t0;
    return M0_OK;
}

}  // namespace


// Standard algebraic notation of a legal move (python-chess Board.san semantics: piece letter, minimal
// disambiguation by file, then rank, then both; 'x'; '=Q'; O-O / O-O-O; '+' / '#') -- PGN output of arena games
// (arena.py:281-303 writes them with chess.pgn).
static std::string san_of(const Pos& p, Move m, const Move* legal, int nlegal) {
    const int from = mv_from(m), to = mv_to(m), promo = mv_promo(m);
    const int pt = piece_type_at(p, from);
    std::string s;
    if (pt == KING && abs((to & 7) - (from & 7)) == 2) {
        s = (to & 7) > (from & 7) ? "O-O" : "O-O-O";
    } else {
        const bool capture = ((occ_of(p, p.turn ^ 1) >> to) & 1ull) || (pt == PAWN && (from & 7) != (to & 7));
        if (pt != PAWN) {
            s += "NBRQK"[pt - 1];
            bool any = false, same_file = false, same_rank = false;
            for (int i = 0; i < nlegal; ++i) {
                const Move o = legal[i];
                if (o == m || mv_to(o) != to || mv_from(o) == from || piece_type_at(p, mv_from(o)) != pt) continue;
                any = true;
                if ((mv_from(o) & 7) == (from & 7)) same_file = true;
                if ((mv_from(o) >> 3) == (from >> 3)) same_rank = true;
            }
            if (any) {
                if (!same_file) s += (char)('a' + (from & 7));
                else if (!same_rank) s += (char)('1' + (from >> 3));
                else { s += (char)('a' + (from & 7)); s += (char)('1' + (from >> 3)); }
            }
        } else if (capture) {
            s += (char)('a' + (from & 7));
        }
        if (capture) s += 'x';
        s += (char)('a' + (to & 7));
        s += (char)('1' + (to >> 3));
        if (promo) { s += '='; s += "NBRQ"[promo - 1]; }
    }
    Pos q = p;
    make_move(q, m);
    if (in_check(q)) s += any_legal(q) ? '+' : '#';
    return s;
}

extern "C" {

static m0_selfplay* selfplay_create_impl(m0_net* nh, m0_net* nh_b, const m0_selfplay_cfg* cfg, bool arena = false) {
    if (!cfg) { m0_set_error("cfg is null"); return nullptr; }
    if (cfg->concurrent_games <= 0 || cfg->inference_batch_size <= 0 || cfg->num_simulations <= 0) {
        m0_set_error("concurrent_games, inference_batch_size and num_simulations must be positive");
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { m0_set_error("no HIP device available (no CPU fallback)"); return nullptr; }
    m0_selfplay* sp = new m0_selfplay();
    sp->cfg = *cfg;
    fill_tree_cfg(*cfg, sp->tc);
    sp->nethandle = nh;
    sp->nethandle_b = nh_b;
    sp->net = m0_net_impl(nh);
    sp->net_b = m0_net_impl(nh_b);
    sp->cfg.arena_mode = (nh_b || arena) ? 1 : 0;
    // A match engine alternates two networks in one game slot and the cache key covers the position only (no network id):
    // side B's leaves would be expanded from side A's cached value and logits.  Off, whatever the caller asked for.
    if (sp->cfg.arena_mode) { sp->cfg.eval_cache = 0; sp->tc.eval_cache = 0; }
    sp->device = nh ? m0_net_device(nh) : 0;
    if (nh && (unsigned)sp->device < (unsigned)M0_MAX_DEVICES) { g_engines_with_net[sp->device].fetch_add(1); sp->counted = true; }
    (void)hipSetDevice(sp->device);
    if (nh) sp->stream = m0_net_stream(nh);
    else { (void)hipStreamCreateWithFlags(&sp->stream, hipStreamNonBlocking); sp->own_stream = true; }
    sp->G = cfg->concurrent_games;
    sp->L = cfg->inference_batch_size;
    // arena: reused subtree + one search of new children (218 max per expansion is far above the ~35 mean)
    long want = cfg->arena_nodes > 0 ? cfg->arena_nodes : (long)(cfg->num_simulations * 1.3 + 64) * 96;
    if (want < 4096) want = 4096;
    sp->cap = (int)want;
    sp->rows_max = (sp->G * (sp->L + 1) + 3) & ~3;      // L leaves + the re-evaluation of a reused root, per game
    const int nreg = sp->cfg.arena_mode ? 2 : 1;      // batch regions: one per network
    memset(&sp->stats, 0, sizeof(sp->stats));
    memset(&sp->d, 0, sizeof(sp->d));
    const size_t N = (size_t)sp->G * 2 * sp->cap;
    TreeArrays& t = sp->d.t;
    t.cap = sp->cap;
    t.prior = dalloc<double>(sp, N); t.w = dalloc<double>(sp, N); t.q = dalloc<double>(sp, N);
    t.n = dalloc<int>(sp, N); t.vl = dalloc<int>(sp, N); t.cbase = dalloc<int>(sp, N);
    t.nch = dalloc<int16_t>(sp, N); t.mv = dalloc<uint16_t>(sp, N); t.midx = dalloc<uint16_t>(sp, N);
    sp->d.games = dalloc<GameDev>(sp, sp->G);
    const size_t LS = (size_t)sp->L + 1;
    sp->d.samples = dalloc<Sample>(sp, (size_t)sp->G * LS);
    sp->d.paths = dalloc<int>(sp, (size_t)sp->G * LS * M0_MAX_DEPTH);
    sp->d.leaf_moves = dalloc<uint16_t>(sp, (size_t)sp->G * LS * M0_MAX_CHILDREN);
    sp->d.tt_sides = 1;
    if (cfg->tt_merge) {
        int tc = 1024;
        while (tc < 2 * sp->cap) tc <<= 1;
        sp->d.tt_cap = tc;
        // match engine: one table per side, kept for the whole game (the reference keeps one MCTS object per side, arena.py:157-158)
        sp->d.tt_sides = sp->cfg.arena_mode ? 2 : 1;
        // a side's half that cannot hold one more search starts over BEFORE that search (advance_kernel), not in the middle of it:
        // ~35 children per expansion on average, 48 with margin, never more than half of the half
        sp->d.search_nodes = (int)std::min<long>((long)cfg->num_simulations * 48 + 4 * M0_MAX_CHILDREN, (long)sp->cap / 2);
        sp->d.epaths = dalloc<int>(sp, (size_t)sp->G * LS * M0_MAX_DEPTH);
        sp->d.tt_keys = dalloc<uint64_t>(sp, (size_t)sp->G * sp->d.tt_sides * tc);
        sp->d.tt_nodes = dalloc<int>(sp, (size_t)sp->G * sp->d.tt_sides * tc);
        if (!sp->d.epaths || !sp->d.tt_keys || !sp->d.tt_nodes) {
            m0_set_error("hipMalloc failed for the position tables (tt_merge): lower concurrent_games or arena_nodes");
            m0_selfplay_destroy(sp);
            return nullptr;
        }
    }
    if (sp->tc.eval_cache) {
        int entries = cfg->eval_cache_entries > 0 ? cfg->eval_cache_entries : 16384;
        int sets = 64;
        while (sets * 4 < entries) sets <<= 1;
        EvalCache& ec = sp->d.ec;
        ec.sets = sets;
        ec.keys = dalloc<uint64_t>(sp, (size_t)sp->G * sets * 4);
        ec.stamps = dalloc<uint32_t>(sp, (size_t)sp->G * sets * 4);
        ec.payload = dalloc<float>(sp, (size_t)sp->G * sets * 4 * M0_EC_WORDS);
        ec.hit_stage = dalloc<float>(sp, (size_t)sp->G * LS * M0_EC_WORDS);
        if (!ec.keys || !ec.stamps || !ec.payload || !ec.hit_stage) {
            m0_set_error("hipMalloc failed for the evaluation cache: lower eval_cache_entries or concurrent_games");
            m0_selfplay_destroy(sp);
            return nullptr;
        }
    }
    sp->d.hist = dalloc<uint64_t>(sp, (size_t)sp->G * M0_HIST_CAP);
    sp->d.results = dalloc<RootResult>(sp, sp->G);
    sp->d.row_counter = dalloc<int>(sp, 4);
    sp->d.x0 = dalloc<_Float16>(sp, (size_t)(nreg * sp->rows_max + 4) * 64 * 32);
    sp->logits_dev = dalloc<float>(sp, (size_t)nreg * sp->rows_max * 4672);
    sp->values_dev = dalloc<float>(sp, (size_t)nreg * sp->rows_max + 4);
    sp->d.net_row_base = sp->rows_max;
    if (cfg->ssl_in_forward && sp->net && sp->net->ssl_channels_total() > 0)
        sp->ssl_dev = dalloc<float>(sp, (size_t)sp->rows_max * sp->net->ssl_channels_total() * 64);
    if (cfg->ssl_targets) {
        sp->ssl_cap = cfg->max_game_len > 0 ? cfg->max_game_len + 1 : 513;
        sp->ssl_pos_dev = dalloc<Pos>(sp, sp->ssl_cap);
        sp->ssl_out_dev = dalloc<float>(sp, (size_t)sp->ssl_cap * 17 * 64);
        if (!sp->ssl_pos_dev || !sp->ssl_out_dev) {
            m0_set_error("hipMalloc failed for the SSL target staging buffers");
            m0_selfplay_destroy(sp);
            return nullptr;
        }
    }
    sp->ids_dev = dalloc<int>(sp, sp->G);
    sp->slots_dev = dalloc<int>(sp, sp->G);
    sp->d.logits = sp->logits_dev; sp->d.values = sp->values_dev;
    sp->d.G = sp->G; sp->d.L = sp->L;
    bool ok = t.prior && t.w && t.q && t.n && t.vl && t.cbase && t.nch && t.mv && t.midx && sp->d.games && sp->d.samples &&
              sp->d.paths && sp->d.leaf_moves && sp->d.hist && sp->d.results && sp->d.row_counter && sp->d.x0 && sp->logits_dev && sp->values_dev &&
              sp->ids_dev && sp->slots_dev;
    if (!ok) {
        m0_set_error("hipMalloc failed for the search arenas (lower concurrent_games or arena_nodes)");
        m0_selfplay_destroy(sp);
        return nullptr;
    }
    sp->hg.assign(sp->G, GameDev());
    for (auto& g : sp->hg) memset(&g, 0, sizeof(GameDev));
    sp->games.assign(sp->G, HostGame());
    sp->hres.resize(sp->G);
    sp->hsamples.resize((size_t)sp->G * (sp->L + 1));
    sp->prev_done.assign(sp->G, 0);
    (void)hipEventCreate(&sp->ev0); (void)hipEventCreate(&sp->ev1); (void)hipEventCreate(&sp->ev2); (void)hipEventCreate(&sp->ev3);
    if (cfg->tail_split && sp->net && !sp->cfg.arena_mode && sp->net->cfg().channels > 256 && sp->net->cfg().channels <= 320 &&
        sp->rows_max >= 2048) {
        // M0_TAIL_CU_MASK=w0,...,w7 (measurement switch, like M0_NET_CU_MASK for the network's own stream): the second instance's
        // stream runs on those CUs only -- with complementary masks every launch of the two halves has a known share of the chip
        hipError_t tse;
        if (const char* mk = getenv("M0_TAIL_CU_MASK"); mk && *mk) {
            uint32_t words[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int nw = 0;
            for (const char* q = mk; *q && nw < 8; ++nw) {
                char* end = nullptr;
                words[nw] = (uint32_t)strtoul(q, &end, 16);
                if (end == q) break;
                q = (*end == ',') ? end + 1 : end;
            }
            tse = hipExtStreamCreateWithCUMask(&sp->stream_tail, 8, words);
        } else {
            tse = hipStreamCreateWithFlags(&sp->stream_tail, hipStreamNonBlocking);
        }
        if (tse == hipSuccess &&
            hipEventCreateWithFlags(&sp->ev_sel, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&sp->ev_tail, hipEventDisableTiming) == hipSuccess) {
            sp->net_tail = sp->net->shared_view(sp->stream_tail);
            std::string werr;                   // its workspace now, at its largest (a regrowth synchronises the device)
            sp->half_split = cfg->tail_split == 2;
            if (sp->net_tail->ensure_workspace(sp->half_split ? sp->rows_max / 2 + 1024 : 1023, werr) != M0_OK) {
                m0_set_error("tail split: " + werr);
                m0_selfplay_destroy(sp);
                return nullptr;
            }
        } else {
            m0_set_error("hipStreamCreate / hipEventCreate failed (tail split)");
            m0_selfplay_destroy(sp);
            return nullptr;
        }
    }
    if (sp->net) {
        std::string err;
        if (sp->net->ensure_workspace(sp->rows_max, err) != M0_OK) { m0_set_error(err); m0_selfplay_destroy(sp); return nullptr; }
        if (sp->net_b && sp->net_b->ensure_workspace(sp->rows_max, err) != M0_OK) { m0_set_error(err); m0_selfplay_destroy(sp); return nullptr; }
    }
    // the clears ride the engine's stream; wait once so that an allocation / clear failure surfaces here, not in the first step
    if (hipStreamSynchronize(sp->stream) != hipSuccess) {
        m0_set_error(std::string("clearing the search arenas failed: ") + hipGetErrorString(hipGetLastError()));
        m0_selfplay_destroy(sp);
        return nullptr;
    }
    return sp;
}

m0_selfplay* m0_selfplay_create(m0_net* nh, const m0_selfplay_cfg* cfg) { return selfplay_create_impl(nh, nullptr, cfg); }

m0_selfplay* m0_arena_create(m0_net* net_a, m0_net* net_b, const m0_selfplay_cfg* cfg) {
    if (!net_a || !net_b) { m0_set_error("m0_arena_create needs two networks"); return nullptr; }
    if (m0_net_device(net_a) != m0_net_device(net_b)) { m0_set_error("both networks must live on the same HIP device"); return nullptr; }
    if (cfg && (cfg->ssl_in_forward || cfg->ssl_targets)) { m0_set_error("arena games carry no SSL outputs"); return nullptr; }
    return selfplay_create_impl(net_a, net_b, cfg);
}

m0_selfplay* m0_arena_create_ext(const m0_selfplay_cfg* cfg) {
    if (cfg && (cfg->ssl_in_forward || cfg->ssl_targets)) { m0_set_error("arena games carry no SSL outputs"); return nullptr; }
    return selfplay_create_impl(nullptr, nullptr, cfg, true);
}

void m0_selfplay_destroy(m0_selfplay* sp) {
    if (!sp) return;
    if (sp->counted) g_engines_with_net[sp->device].fetch_sub(1);
    (void)hipSetDevice(sp->device);
    if (sp->stream) (void)hipStreamSynchronize(sp->stream);
    if (sp->stream_tail) (void)hipStreamSynchronize(sp->stream_tail);
    delete sp->net_tail;
    if (sp->ev_sel) (void)hipEventDestroy(sp->ev_sel);
    if (sp->ev_tail) (void)hipEventDestroy(sp->ev_tail);
    if (sp->stream_tail) (void)hipStreamDestroy(sp->stream_tail);
    for (void* p : sp->allocs) (void)hipFree(p);
    for (auto& r : sp->done_meta) delete (GameRecordOwner*)r.owner;
    if (sp->ev0) { (void)hipEventDestroy(sp->ev0); (void)hipEventDestroy(sp->ev1); (void)hipEventDestroy(sp->ev2); (void)hipEventDestroy(sp->ev3); }
    if (sp->own_stream && sp->stream) (void)hipStreamDestroy(sp->stream);
    delete sp;
}

// lazily start the first games
static int start_first_games(m0_selfplay* sp) {
    if (sp->stats.games_started != 0) return M0_OK;
    std::vector<int> ids, slots;
    if (sync_games_d2h(sp) != 0) { m0_set_error("device sync failed"); return M0_ERR_HIP; }
    for (int s = 0; s < sp->G; ++s) {
        if (sp->cfg.total_games > 0 && sp->next_game >= sp->cfg.total_games) break;
        start_game(sp, s, ids, slots);
    }
    if (apply_advances(sp, ids, slots) != 0) { m0_set_error("advance failed"); return M0_ERR_HIP; }
    int act = 0;
    for (int s = 0; s < sp->G; ++s) act += sp->games[s].in_use ? 1 : 0;
    sp->stats.active_games = act;
    return M0_OK;
}

int m0_selfplay_set_openings(m0_selfplay* sp, const char* const* fens, int n) {
    if (!sp || n < 0 || (n > 0 && !fens)) { m0_set_error("invalid argument"); return M0_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(sp->mu);
    if (sp->stats.games_started != 0) { m0_set_error("set the opening book before the first step"); return M0_ERR_STATE; }
    std::vector<Pos> book(n);
    for (int i = 0; i < n; ++i)
        if (!fens[i] || parse_fen(fens[i], book[i]) != 0) { m0_set_error(std::string("bad FEN at index ") + std::to_string(i)); return M0_ERR_INVALID; }
    sp->book.swap(book);
    return M0_OK;
}

// first half of a step for an external evaluator: select, then the leaves' planes on the host (region 0 = network A / the only
// network, region 1 = network B of a match engine, whose rows start at d.net_row_base on the device)
static int ext_select_impl(m0_selfplay* sp, int* rows_a, int* rows_b, float* planes_a, float* planes_b, int max_rows) {
    (void)hipSetDevice(sp->device);
    if (sp->ext_pending) { m0_set_error("m0_selfplay_ext_expand outstanding"); return M0_ERR_STATE; }
    // select applies virtual losses and reserves batch rows: refuse a buffer that cannot take the worst case BEFORE it runs
    // (an error after it would leave the engine waiting for an ext_expand the caller has no planes for)
    if (!planes_a || (sp->cfg.arena_mode && !planes_b) || max_rows < sp->G * (sp->L + 1)) {
        m0_set_error("planes buffer too small: concurrent_games * (inference_batch_size + 1) rows are required");
        return M0_ERR_INVALID;
    }
    int rc = start_first_games(sp);
    if (rc != M0_OK) return rc;
    int r = 0;
    if (run_select(sp, &r) != 0) { m0_set_error(std::string("select failed: ") + hipGetErrorString(hipGetLastError())); return M0_ERR_HIP; }
    const int rb = sp->cfg.arena_mode ? sp->rows2[1] : 0;
    if (r > sp->rows_max || rb > sp->rows_max) { m0_set_error("row counter overflow"); return M0_ERR_STATE; }
    *rows_a = r;
    if (rows_b) *rows_b = rb;
    sp->last_rows = r;
    sp->last_rows_b = rb;
    sp->ext_pending = true;
    if (r + rb > 0) {
        if (sync_games_d2h(sp) != 0) { m0_set_error("device sync failed"); return M0_ERR_HIP; }
        (void)hipMemcpy(sp->hsamples.data(), sp->d.samples, sizeof(Sample) * (size_t)sp->G * (sp->L + 1), hipMemcpyDeviceToHost);
        const int base = sp->d.net_row_base;
        for (int g = 0; g < sp->G; ++g) {
            if (!sp->hg[g].active) continue;
            for (int s = 0; s < sp->hg[g].nsamples; ++s) {
                const Sample& smp = sp->hsamples[(size_t)g * (sp->L + 1) + s];
                if (!(smp.kind == 1 || smp.kind == 2 || smp.kind == 4) || smp.row < 0) continue;
                if (smp.row < r) encode_planes_f32(smp.pos, planes_a + (size_t)smp.row * 19 * 64);
                else if (rb > 0 && smp.row >= base && smp.row < base + rb) encode_planes_f32(smp.pos, planes_b + (size_t)(smp.row - base) * 19 * 64);
            }
        }
    }
    return M0_OK;
}

static int ext_expand_impl(m0_selfplay* sp, const float* logits_a, const float* values_a, int rows_a, const float* logits_b,
                           const float* values_b, int rows_b) {
    (void)hipSetDevice(sp->device);
    if (!sp->ext_pending) { m0_set_error("no m0_selfplay_ext_select outstanding"); return M0_ERR_STATE; }
    if (rows_a != sp->last_rows || rows_b != sp->last_rows_b) { m0_set_error("rows does not match the last select"); return M0_ERR_INVALID; }
    if ((rows_a > 0 && (!logits_a || !values_a)) || (rows_b > 0 && (!logits_b || !values_b))) { m0_set_error("null argument"); return M0_ERR_INVALID; }
    if (rows_a > 0) {
        (void)hipMemcpyAsync(sp->logits_dev, logits_a, (size_t)rows_a * 4672 * 4, hipMemcpyHostToDevice, sp->stream);
        (void)hipMemcpyAsync(sp->values_dev, values_a, (size_t)rows_a * 4, hipMemcpyHostToDevice, sp->stream);
    }
    if (rows_b > 0) {
        const size_t base = (size_t)sp->d.net_row_base;
        (void)hipMemcpyAsync(sp->logits_dev + base * 4672, logits_b, (size_t)rows_b * 4672 * 4, hipMemcpyHostToDevice, sp->stream);
        (void)hipMemcpyAsync(sp->values_dev + base, values_b, (size_t)rows_b * 4, hipMemcpyHostToDevice, sp->stream);
    }
    sp->ext_pending = false;
    std::string err;
    (void)hipEventRecord(sp->ev0, sp->stream); (void)hipEventRecord(sp->ev1, sp->stream);
    int rc = step_back(sp, rows_a + rows_b, now_ms(), err);
    if (rc != M0_OK) m0_set_error(err);
    return rc;
}

int m0_selfplay_ext_select(m0_selfplay* sp, int* rows, float* planes, int max_rows) {
    if (!sp || !rows) { m0_set_error("null argument"); return M0_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(sp->mu);
    if (sp->cfg.arena_mode) { m0_set_error("a match engine has two evaluators: use m0_arena_ext_select"); return M0_ERR_STATE; }
    return ext_select_impl(sp, rows, nullptr, planes, nullptr, max_rows);
}

int m0_selfplay_ext_expand(m0_selfplay* sp, const float* logits, const float* values, int rows) {
    if (!sp) { m0_set_error("null argument"); return M0_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(sp->mu);
    if (sp->cfg.arena_mode) { m0_set_error("a match engine has two evaluators: use m0_arena_ext_expand"); return M0_ERR_STATE; }
    return ext_expand_impl(sp, logits, values, rows, nullptr, nullptr, 0);
}

int m0_arena_ext_select(m0_selfplay* sp, int* rows_a, int* rows_b, float* planes_a, float* planes_b, int max_rows) {
    if (!sp || !rows_a || !rows_b) { m0_set_error("null argument"); return M0_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(sp->mu);
    if (!sp->cfg.arena_mode) { m0_set_error("not a match engine"); return M0_ERR_STATE; }
    return ext_select_impl(sp, rows_a, rows_b, planes_a, planes_b, max_rows);
}

int m0_arena_ext_expand(m0_selfplay* sp, const float* logits_a, const float* values_a, int rows_a, const float* logits_b,
                        const float* values_b, int rows_b) {
    if (!sp) { m0_set_error("null argument"); return M0_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(sp->mu);
    if (!sp->cfg.arena_mode) { m0_set_error("not a match engine"); return M0_ERR_STATE; }
    return ext_expand_impl(sp, logits_a, values_a, rows_a, logits_b, values_b, rows_b);
}

int m0_selfplay_step(m0_selfplay* sp, int steps) {
    if (!sp) { m0_set_error("sp is null"); return M0_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(sp->mu);
    (void)hipSetDevice(sp->device);
    std::string err;
    if (sp->ext_pending) { m0_set_error("m0_selfplay_ext_expand outstanding"); return M0_ERR_STATE; }
    { int rc0 = start_first_games(sp); if (rc0 != M0_OK) return rc0; }
    for (int i = 0; i < steps; ++i) {
        if (sp->stats.active_games == 0 && sp->stats.steps > 0) break;
        int rc = one_step(sp, err);
        if (rc != M0_OK) { m0_set_error(err); return rc; }
    }
    return M0_OK;
}

int m0_selfplay_stats_get(m0_selfplay* sp, m0_selfplay_stats* out) {
    if (!sp || !out) { m0_set_error("null argument"); return M0_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(sp->mu);
    *out = sp->stats;
    return M0_OK;
}

int m0_selfplay_poll(m0_selfplay* sp, m0_game_record* out) {
    if (!sp || !out) { m0_set_error("null argument"); return M0_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(sp->mu);
    if (sp->done_meta.empty()) return 0;
    *out = sp->done_meta.front();
    sp->done_meta.pop_front();
    return 1;
}

void m0_game_record_free(m0_game_record* rec) {
    if (rec && rec->owner) { delete (GameRecordOwner*)rec->owner; rec->owner = nullptr; }
}

int m0_selfplay_running(m0_selfplay* sp) {
    if (!sp) return 0;
    std::lock_guard<std::mutex> lk(sp->mu);
    if (sp->stats.games_started == 0) return 1;
    if (sp->stats.active_games > 0) return 1;
    return (sp->cfg.total_games <= 0 || sp->next_game < sp->cfg.total_games) ? 1 : 0;
}

// ---------------- split-step search ----------------
int m0_search_begin(m0_selfplay* sp, int g, const char* fen, int sims, int dirichlet, int game_uid) {
    if (!sp || !fen || g < 0 || g >= sp->G || sims <= 0) { m0_set_error("invalid argument"); return M0_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(sp->mu);
    (void)hipSetDevice(sp->device);
    Pos p;
    if (parse_fen(fen, p) != 0) { m0_set_error("bad FEN"); return M0_ERR_INVALID; }
    if (sync_games_d2h(sp) != 0) { m0_set_error("device sync failed"); return M0_ERR_HIP; }
    HostGame& hgm = sp->games[g];
    hgm = HostGame();
    hgm.in_use = true; hgm.pos = p; hgm.game_index = game_uid;
    seed_game_dev(sp->hg[g], sp->cfg.seed, game_uid);
    sp->hg[g].evals = 0;
    arm_search(sp, g, p, hgm.win, sims, dirichlet != 0, true);
    std::vector<int> ids{g}, slots{-1};
    if (apply_advances(sp, ids, slots) != 0) { m0_set_error("advance failed"); return M0_ERR_HIP; }
    return M0_OK;
}

int m0_search_select(m0_selfplay* sp, int* rows, float* planes, int max_rows) {
    if (!sp || !rows) { m0_set_error("null argument"); return M0_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(sp->mu);
    (void)hipSetDevice(sp->device);
    int r = 0;
    if (run_select(sp, &r) != 0) { m0_set_error(std::string("select failed: ") + hipGetErrorString(hipGetLastError())); return M0_ERR_HIP; }
    *rows = r;
    sp->last_rows = r;
    if (planes && r > 0) {
        if (r > max_rows) { m0_set_error("planes buffer too small"); return M0_ERR_INVALID; }
        if (sync_games_d2h(sp) != 0) { m0_set_error("device sync failed"); return M0_ERR_HIP; }
        (void)hipMemcpy(sp->hsamples.data(), sp->d.samples, sizeof(Sample) * (size_t)sp->G * (sp->L + 1), hipMemcpyDeviceToHost);
        for (int g = 0; g < sp->G; ++g) {
            if (!sp->hg[g].active) continue;
            for (int s = 0; s < sp->hg[g].nsamples; ++s) {
                const Sample& smp = sp->hsamples[(size_t)g * (sp->L + 1) + s];
                if ((smp.kind == 1 || smp.kind == 2 || smp.kind == 4) && smp.row >= 0 && smp.row < r)
                    encode_planes_f32(smp.pos, planes + (size_t)smp.row * 19 * 64);
            }
        }
    }
    return M0_OK;
}

int m0_search_expand(m0_selfplay* sp, const float* logits, const float* values, int rows) {
    if (!sp) { m0_set_error("null argument"); return M0_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(sp->mu);
    (void)hipSetDevice(sp->device);
    if (rows != sp->last_rows || rows > sp->rows_max) { m0_set_error("rows does not match the last select"); return M0_ERR_INVALID; }
    if (rows > 0) {
        if (!logits || !values) { m0_set_error("null argument"); return M0_ERR_INVALID; }
        (void)hipMemcpyAsync(sp->logits_dev, logits, (size_t)rows * 4672 * 4, hipMemcpyHostToDevice, sp->stream);
        (void)hipMemcpyAsync(sp->values_dev, values, (size_t)rows * 4, hipMemcpyHostToDevice, sp->stream);
    }
    if (launch_expand(sp->d, sp->tc, sp->stream) != hipSuccess) { m0_set_error("expand launch failed"); return M0_ERR_HIP; }
    if (sync_games_d2h(sp) != 0) { m0_set_error(std::string("expand failed: ") + hipGetErrorString(hipGetLastError())); return M0_ERR_HIP; }
    sp->stats.evals += (uint64_t)rows;
    return M0_OK;
}

int m0_search_result(m0_selfplay* sp, int g, int* nchild, int32_t* child_n, uint16_t* child_mv, int32_t* child_idx,
                     double* child_prior, double* child_q, double* root_q, int* root_n, int* finished) {
    if (!sp || g < 0 || g >= sp->G) { m0_set_error("invalid argument"); return M0_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(sp->mu);
    (void)hipSetDevice(sp->device);
    if (finished) *finished = sp->hg[g].finished;
    if (!sp->hg[g].finished) { if (nchild) *nchild = 0; return M0_OK; }
    (void)hipMemcpy(&sp->hres[g], sp->d.results + g, sizeof(RootResult), hipMemcpyDeviceToHost);
    const RootResult& R = sp->hres[g];
    if (nchild) *nchild = R.nchild;
    for (int i = 0; i < R.nchild; ++i) {
        if (child_n) child_n[i] = R.child_n[i];
        if (child_mv) child_mv[i] = R.child_mv[i];
        if (child_idx) child_idx[i] = R.child_idx[i];
        if (child_prior) child_prior[i] = R.child_prior[i];
        if (child_q) child_q[i] = R.child_q[i];
    }
    if (root_q) *root_q = R.root_n > 0 ? R.root_q : sp->hg[g].root_v;
    if (root_n) *root_n = R.root_n;
    return M0_OK;
}

int m0_search_advance(m0_selfplay* sp, int g, int slot, int sims, int dirichlet) {
    if (!sp || g < 0 || g >= sp->G || sims <= 0) { m0_set_error("invalid argument"); return M0_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(sp->mu);
    (void)hipSetDevice(sp->device);
    // refresh the mirror first: an earlier advance changed root/next/arena on the device only
    if (sync_games_d2h(sp) != 0) { m0_set_error("device sync failed"); return M0_ERR_HIP; }
    if (!sp->hg[g].finished) { m0_set_error("search not finished"); return M0_ERR_STATE; }
    (void)hipMemcpy(&sp->hres[g], sp->d.results + g, sizeof(RootResult), hipMemcpyDeviceToHost);
    const RootResult& R = sp->hres[g];
    if (slot < 0 || slot >= R.nchild) { m0_set_error("child slot out of range"); return M0_ERR_INVALID; }
    HostGame& hgm = sp->games[g];
    const Move mv = R.child_mv[slot];
    hgm.win.push(hgm.pos, mv);
    make_move(hgm.pos, mv);
    hgm.history.push_back(mv);
    const bool fresh = sp->cfg.fresh_tree_per_move || sp->cfg.tt_merge;
    arm_search(sp, g, hgm.pos, hgm.win, sims, dirichlet != 0, fresh);
    std::vector<int> ids{g}, slots{fresh ? -1 : slot};
    if (apply_advances(sp, ids, slots) != 0) { m0_set_error("advance failed"); return M0_ERR_HIP; }
    return M0_OK;
}

// ---------------- encoding.py on the device ----------------
int m0_encode_fens(int hip_device, const char* const* fens, int n, float* planes, uint8_t* mask, int32_t* nlegal,
                   uint16_t* moves, int32_t* idx) {
    if (!fens || n <= 0) { m0_set_error("invalid argument"); return M0_ERR_INVALID; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { m0_set_error("no HIP device available (no CPU fallback)"); return M0_ERR_HIP; }
    if (hipSetDevice(hip_device) != hipSuccess) { m0_set_error("hipSetDevice failed"); return M0_ERR_HIP; }
    std::vector<Pos> hp(n);
    for (int i = 0; i < n; ++i)
        if (!fens[i] || parse_fen(fens[i], hp[i]) != 0) { m0_set_error(std::string("bad FEN at index ") + std::to_string(i)); return M0_ERR_INVALID; }
    Pos* dp = nullptr; float* dpl = nullptr; uint8_t* dm = nullptr; int32_t* dn = nullptr; uint16_t* dmv = nullptr; int32_t* di = nullptr;
    int rc = M0_OK;
    auto fail = [&](const char* what) { m0_set_error(what); rc = M0_ERR_HIP; };
    if (hipMalloc((void**)&dp, sizeof(Pos) * n) != hipSuccess) fail("hipMalloc failed");
    if (rc == M0_OK && planes && hipMalloc((void**)&dpl, (size_t)n * 19 * 64 * 4) != hipSuccess) fail("hipMalloc failed");
    if (rc == M0_OK && mask && hipMalloc((void**)&dm, (size_t)n * 4672) != hipSuccess) fail("hipMalloc failed");
    if (rc == M0_OK && nlegal && hipMalloc((void**)&dn, (size_t)n * 4) != hipSuccess) fail("hipMalloc failed");
    if (rc == M0_OK && moves && hipMalloc((void**)&dmv, (size_t)n * M0_MAX_MOVES * 2) != hipSuccess) fail("hipMalloc failed");
    if (rc == M0_OK && idx && hipMalloc((void**)&di, (size_t)n * M0_MAX_MOVES * 4) != hipSuccess) fail("hipMalloc failed");
    if (rc == M0_OK) {
        (void)hipMemcpy(dp, hp.data(), sizeof(Pos) * n, hipMemcpyHostToDevice);
        if (launch_encode_positions(dp, n, dpl, nullptr, dm, dn, dmv, di, nullptr) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
            fail("encode kernel failed");
    }
    if (rc == M0_OK) {
        if (planes) (void)hipMemcpy(planes, dpl, (size_t)n * 19 * 64 * 4, hipMemcpyDeviceToHost);
        if (mask) (void)hipMemcpy(mask, dm, (size_t)n * 4672, hipMemcpyDeviceToHost);
        if (nlegal) (void)hipMemcpy(nlegal, dn, (size_t)n * 4, hipMemcpyDeviceToHost);
        if (moves) (void)hipMemcpy(moves, dmv, (size_t)n * M0_MAX_MOVES * 2, hipMemcpyDeviceToHost);
        if (idx) (void)hipMemcpy(idx, di, (size_t)n * M0_MAX_MOVES * 4, hipMemcpyDeviceToHost);
    }
    if (dp) (void)hipFree(dp); if (dpl) (void)hipFree(dpl); if (dm) (void)hipFree(dm);
    if (dn) (void)hipFree(dn); if (dmv) (void)hipFree(dmv); if (di) (void)hipFree(di);
    return rc;
}

int m0_selfplay_last_batch_nhwc(m0_selfplay* sp, uint16_t* out, int max_rows, int* rows) {
    if (!sp || !out || !rows) { m0_set_error("null argument"); return M0_ERR_INVALID; }
    std::lock_guard<std::mutex> lk(sp->mu);
    (void)hipSetDevice(sp->device);
    const int r = sp->last_rows;
    if (r > max_rows) { m0_set_error("output buffer too small"); return M0_ERR_INVALID; }
    *rows = r;
    if (r > 0) {
        if (hipMemcpyAsync(out, sp->d.x0, (size_t)r * 64 * 32 * 2, hipMemcpyDeviceToHost, sp->stream) != hipSuccess ||
            hipStreamSynchronize(sp->stream) != hipSuccess) { m0_set_error("copy failed"); return M0_ERR_HIP; }
    }
    return M0_OK;
}

int m0_encode_fens_nhwc(int hip_device, const char* const* fens, int n, uint16_t* out) {
    if (!fens || n <= 0 || !out) { m0_set_error("invalid argument"); return M0_ERR_INVALID; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { m0_set_error("no HIP device available (no CPU fallback)"); return M0_ERR_HIP; }
    if (hipSetDevice(hip_device) != hipSuccess) { m0_set_error("hipSetDevice failed"); return M0_ERR_HIP; }
    std::vector<Pos> hp(n);
    for (int i = 0; i < n; ++i)
        if (!fens[i] || parse_fen(fens[i], hp[i]) != 0) { m0_set_error(std::string("bad FEN at index ") + std::to_string(i)); return M0_ERR_INVALID; }
    Pos* dp = nullptr; _Float16* dx = nullptr;
    int rc = M0_OK;
    if (hipMalloc((void**)&dp, sizeof(Pos) * n) != hipSuccess || hipMalloc((void**)&dx, (size_t)n * 64 * 32 * 2) != hipSuccess) {
        m0_set_error("hipMalloc failed"); rc = M0_ERR_HIP;
    } else {
        (void)hipMemcpy(dp, hp.data(), sizeof(Pos) * n, hipMemcpyHostToDevice);
        if (launch_encode_positions(dp, n, nullptr, dx, nullptr, nullptr, nullptr, nullptr, nullptr) != hipSuccess ||
            hipDeviceSynchronize() != hipSuccess) { m0_set_error("encode kernel failed"); rc = M0_ERR_HIP; }
        else (void)hipMemcpy(out, dx, (size_t)n * 64 * 32 * 2, hipMemcpyDeviceToHost);
    }
    if (dp) (void)hipFree(dp);
    if (dx) (void)hipFree(dx);
    return rc;
}

int m0_ssl_targets_fens(int hip_device, const char* const* fens, int n, float* out) {
    if (!fens || n <= 0 || !out) { m0_set_error("invalid argument"); return M0_ERR_INVALID; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { m0_set_error("no HIP device available (no CPU fallback)"); return M0_ERR_HIP; }
    if (hipSetDevice(hip_device) != hipSuccess) { m0_set_error("hipSetDevice failed"); return M0_ERR_HIP; }
    std::vector<Pos> hp(n);
    for (int i = 0; i < n; ++i)
        if (!fens[i] || parse_fen(fens[i], hp[i]) != 0) { m0_set_error(std::string("bad FEN at index ") + std::to_string(i)); return M0_ERR_INVALID; }
    Pos* dp = nullptr; float* dout = nullptr;
    int rc = M0_OK;
    if (hipMalloc((void**)&dp, sizeof(Pos) * n) != hipSuccess || hipMalloc((void**)&dout, (size_t)n * 17 * 64 * 4) != hipSuccess) {
        m0_set_error("hipMalloc failed"); rc = M0_ERR_HIP;
    } else {
        (void)hipMemcpy(dp, hp.data(), sizeof(Pos) * n, hipMemcpyHostToDevice);
        if (launch_ssl_targets(dp, n, dout, nullptr) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { m0_set_error("ssl kernel failed"); rc = M0_ERR_HIP; }
        else (void)hipMemcpy(out, dout, (size_t)n * 17 * 64 * 4, hipMemcpyDeviceToHost);
    }
    if (dp) (void)hipFree(dp);
    if (dout) (void)hipFree(dout);
    return rc;
}

int m0_move_to_index_fen(int hip_device, const char* fen, const char* uci, int32_t* out) {
    if (!fen || !uci || !out) { m0_set_error("null argument"); return M0_ERR_INVALID; }
    const Move want = parse_uci(uci);
    if (want == 0xFFFF) { m0_set_error(std::string("Illegal move: ") + uci); return M0_ERR_INVALID; }
    std::vector<uint16_t> mv(M0_MAX_MOVES);
    std::vector<int32_t> id(M0_MAX_MOVES);
    int32_t n = 0;
    const char* fens[1] = {fen};
    int rc = m0_encode_fens(hip_device, fens, 1, nullptr, nullptr, &n, mv.data(), id.data());
    if (rc != M0_OK) return rc;
    for (int i = 0; i < n; ++i)
        if (mv[i] == want) { *out = id[i]; return M0_OK; }
    m0_set_error(std::string("Illegal move: ") + uci);     // encoding.py:120-121 raises ValueError
    return M0_ERR_INVALID;
}

int m0_decode_move_fen(int hip_device, const char* fen, int action_idx, char* uci_out) {
    if (!fen || !uci_out) { m0_set_error("null argument"); return M0_ERR_INVALID; }
    if (action_idx < 0 || action_idx >= M0_POLICY_SIZE) { m0_set_error("action_idx out of range"); return M0_ERR_INVALID; }
    Pos p;
    if (parse_fen(fen, p) != 0) { m0_set_error("bad FEN"); return M0_ERR_INVALID; }
    std::vector<uint16_t> mv(M0_MAX_MOVES);
    int32_t n = 0;
    const char* fens[1] = {fen};
    int rc = m0_encode_fens(hip_device, fens, 1, nullptr, nullptr, &n, mv.data(), nullptr);   // legal moves from the device
    if (rc != M0_OK) return rc;
    static const int RAY[8][2] = {{1, 0}, {-1, 0}, {0, 1}, {0, -1}, {1, 1}, {1, -1}, {-1, 1}, {-1, -1}};
    static const int KN[8][2] = {{-2, -1}, {-2, 1}, {-1, -2}, {-1, 2}, {1, -2}, {1, 2}, {2, -1}, {2, 1}};
    const int from = action_idx / 73, off = action_idx % 73;
    const int fr = from >> 3, ff = from & 7;
    int dr, df, steps = 1, promo = 0;
    bool under = false;
    if (off < 56) { dr = RAY[off / 7][0]; df = RAY[off / 7][1]; steps = off % 7 + 1; }
    else if (off < 64) { dr = KN[off - 56][0]; df = KN[off - 56][1]; }
    else {
        const int u = off - 64, d = u % 3;
        static const int DW[3][2] = {{1, 0}, {1, -1}, {1, 1}}, DB[3][2] = {{-1, 0}, {-1, 1}, {-1, -1}};
        dr = p.turn == WHITE ? DW[d][0] : DB[d][0]; df = p.turn == WHITE ? DW[d][1] : DB[d][1];
        promo = u / 3 + 1; under = true;
    }
    const int tr = fr + dr * steps, tf = ff + df * steps;
    int to = -1;
    Move want = 0xFFFF;
    if (tr >= 0 && tr < 8 && tf >= 0 && tf < 8) {
        to = tr * 8 + tf;
        if (!under && piece_type_at(p, from) == PAWN && (p.occ[0] | p.occ[1]) & bit(from) && (tr == 0 || tr == 7)) promo = 4;
        want = mk_move(from, to, promo);
    }
    Move pick = 0xFFFF;
    for (int i = 0; i < n && pick == 0xFFFF; ++i) if (mv[i] == want) pick = mv[i];
    if (to >= 0) for (int i = 0; i < n && pick == 0xFFFF; ++i) if (mv_from(mv[i]) == from && mv_to(mv[i]) == to) pick = mv[i];
    if (to < 0) for (int i = 0; i < n && pick == 0xFFFF; ++i) if (mv_from(mv[i]) == from && mv_to(mv[i]) == 0) pick = mv[i];   // null move target a1 (python Move.null().to_square == 0)
    for (int i = 0; i < n && pick == 0xFFFF; ++i) if (mv_from(mv[i]) == from) pick = mv[i];
    if (pick == 0xFFFF) { strcpy(uci_out, "0000"); return M0_OK; }
    const int f = mv_from(pick), t = mv_to(pick), pr = mv_promo(pick);
    uci_out[0] = (char)('a' + (f & 7)); uci_out[1] = (char)('1' + (f >> 3));
    uci_out[2] = (char)('a' + (t & 7)); uci_out[3] = (char)('1' + (t >> 3));
    uci_out[4] = pr ? " nbrq"[pr] : '\0'; uci_out[5] = '\0';
    return M0_OK;
}

// ---------------- host decision functions ----------------
int m0_sample_move_index(const int32_t* visits, int n, double temperature, double u) {
    if (!visits || n <= 0) return -1;
    return sample_move_index(visits, n, temperature, u);
}
int m0_playout_cap(int sims, double frac, double u) { return playout_cap(sims, frac, u); }
double m0_temperature_for(int fullmove_number, double t_start, double t_end, int t_moves) {
    return temperature_for(fullmove_number, t_start, t_end, t_moves);
}
int m0_rules_probe(const m0_selfplay_cfg* cfg, const char* fen, const char* const* ucis, int n, int* flags, float* result) {
    if (!cfg || !fen || !flags) { m0_set_error("null argument"); return M0_ERR_INVALID; }
    Pos p;
    if (parse_fen(fen, p) != 0) { m0_set_error("bad FEN"); return M0_ERR_INVALID; }
    RepWindow w;
    std::vector<Move> hist;
    for (int i = 0; i < n; ++i) {
        Move m = parse_uci(ucis[i]);
        Move mv[M0_MAX_MOVES];
        int k = gen_legal(p, mv);
        bool ok = false;
        for (int j = 0; j < k; ++j) if (mv[j] == m) ok = true;
        if (!ok) { m0_set_error(std::string("Illegal move: ") + ucis[i]); return M0_ERR_INVALID; }
        w.push(p, m);
        make_move(p, m);
        hist.push_back(m);
    }
    DrawCfg dc = draw_cfg_from(*cfg);
    int f = 0;
    if (is_game_over(p, w, false)) f |= 1;
    if (is_game_over(p, w, true)) f |= 2;
    if (should_adjudicate_draw(p, w, hist, dc)) f |= 4;
    const bool anyl = any_legal(p), chk = in_check(p);
    if (!anyl && chk) f |= 8;
    if (!anyl && !chk) f |= 16;
    if (is_insufficient(p)) f |= 32;
    if (can_claim_fifty(p)) f |= 64;
    if (w.is_repetition(p, 3)) f |= 128;
    if (w.can_claim_threefold(p)) f |= 256;
    if (w.is_repetition(p, 5)) f |= 512;
    if (p.halfmove >= 150 && anyl) f |= 1024;
    *flags = f;
    if (result) *result = game_result(p);
    return M0_OK;
}

int m0_arena_choose_move(const int32_t* visits, int n, double temp, int ply, int temp_plies, double u) {
    if (!visits || n <= 0) { m0_set_error("empty visit list"); return M0_ERR_INVALID; }
    return arena_choose_move(visits, n, temp, ply, temp_plies, u);
}

int m0_san_legal_fen(const char* fen, uint16_t* moves, char* san, int* nlegal) {
    if (!fen || !moves || !san || !nlegal) { m0_set_error("null argument"); return M0_ERR_INVALID; }
    Pos p;
    if (parse_fen(fen, p) != 0) { m0_set_error("bad FEN"); return M0_ERR_INVALID; }
    Move mv[M0_MAX_MOVES];
    const int k = gen_legal(p, mv);
    for (int i = 0; i < k; ++i) {
        moves[i] = mv[i];
        const std::string t = san_of(p, mv[i], mv, k);
        memset(san + 8 * i, 0, 8);
        memcpy(san + 8 * i, t.c_str(), t.size() < 8 ? t.size() : 7);
    }
    *nlegal = k;
    return M0_OK;
}

// Board.fen() of python-chess (en_passant="legal": the ep square only when an en-passant capture is legal; cleaned castling rights)
static std::string fen_of(const Pos& p) {
    std::string s;
    for (int r = 7; r >= 0; --r) {
        int e = 0;
        for (int f = 0; f < 8; ++f) {
            const int sq = r * 8 + f;
            const uint64_t b = bit(sq);
            if (!((p.occ[0] | p.occ[1]) & b)) { ++e; continue; }
            if (e) { s += (char)('0' + e); e = 0; }
            const int t = piece_type_at(p, sq);
            s += ((p.occ[WHITE] & b) ? "PNBRQK" : "pnbrqk")[t];
        }
        if (e) s += (char)('0' + e);
        if (r) s += '/';
    }
    s += p.turn == WHITE ? " w " : " b ";
    const int cr = clean_cr(p);
    std::string c;
    if (cr & CR_WK) c += 'K';
    if (cr & CR_WQ) c += 'Q';
    if (cr & CR_BK) c += 'k';
    if (cr & CR_BQ) c += 'q';
    s += c.empty() ? "-" : c;
    s += ' ';
    if (p.ep >= 0 && has_legal_ep(p)) { s += (char)('a' + (p.ep & 7)); s += (char)('1' + (p.ep >> 3)); }
    else s += '-';
    s += ' ' + std::to_string(p.halfmove) + ' ' + std::to_string(p.fullmove);
    return s;
}

int m0_fen_after(const char* fen, const char* const* ucis, int n, char* fen_out, int cap) {
    if (!fen || !fen_out || cap <= 0 || (n > 0 && !ucis)) { m0_set_error("null argument"); return M0_ERR_INVALID; }
    Pos p;
    if (parse_fen(fen, p) != 0) { m0_set_error("bad FEN"); return M0_ERR_INVALID; }
    for (int i = 0; i < n; ++i) {
        const Move m = ucis[i] ? parse_uci(ucis[i]) : (Move)0xFFFF;
        Move mv[M0_MAX_MOVES];
        const int k = gen_legal(p, mv);
        bool ok = false;
        for (int j = 0; j < k; ++j) if (mv[j] == m) ok = true;
        if (!ok) { m0_set_error(std::string("Illegal move: ") + (ucis[i] ? ucis[i] : "(null)")); return M0_ERR_INVALID; }
        make_move(p, m);
    }
    const std::string f = fen_of(p);
    if ((int)f.size() + 1 > cap) { m0_set_error("output buffer too small"); return M0_ERR_INVALID; }
    memcpy(fen_out, f.c_str(), f.size() + 1);
    return M0_OK;
}

int m0_san_game(const uint16_t* moves, int n, char* out, int cap) {
    if ((!moves && n > 0) || !out || cap <= 0) { m0_set_error("null argument"); return M0_ERR_INVALID; }
    Pos p;
    parse_fen("rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1", p);
    std::string text;
    for (int i = 0; i < n; ++i) {
        Move mv[M0_MAX_MOVES];
        const int k = gen_legal(p, mv);
        bool ok = false;
        for (int j = 0; j < k; ++j) if (mv[j] == moves[i]) ok = true;
        if (!ok) { m0_set_error("illegal move in game"); return M0_ERR_INVALID; }
        if (p.turn == WHITE) text += std::to_string(p.fullmove) + ". ";
        text += san_of(p, moves[i], mv, k);
        text += ' ';
        make_move(p, moves[i]);
    }
    if ((int)text.size() + 1 > cap) { m0_set_error("output buffer too small"); return M0_ERR_INVALID; }
    memcpy(out, text.c_str(), text.size() + 1);
    return (int)text.size();
}

}  // extern "C"
