// Host-side game logic of the self-play loop (no device code): repetition window and claimable
// draws (python-chess semantics), draw adjudication (azchess/draw.py:8-84), temperature, move
// sampling, resign rule, playout cap, result (azchess/selfplay/internal.py:386-394, 507-536,
// 587-599, 690-750; azchess/mcts.py:378-387), counter-based random streams.
#pragma once
#include <math.h>
#include <stdint.h>
#include <set>
#include <vector>
#include "../../include/m0_engine.h"
#include "chess_core.h"

namespace m0 {

enum { PURPOSE_JITTER = 1, PURPOSE_NOISE = 2, PURPOSE_DIRICHLET = 3, PURPOSE_GAME = 4 };

inline uint64_t derive_seed(uint64_t base, int game, int purpose) {
    const uint64_t G = 0x9E3779B97F4A7C15ull;
    return mix64(mix64(base + G * (uint64_t)(game + 1)) ^ ((uint64_t)purpose * 0xD6E8FEB86659FD93ull));
}

struct HStream {
    uint64_t seed = 0, ctr = 0;
    HStream() {}
    explicit HStream(uint64_t s) : seed(s), ctr(0) {}
    double next() {
        const uint64_t G = 0x9E3779B97F4A7C15ull;
        double u = (double)(mix64(seed + (ctr + 1) * G) >> 11) * (1.0 / 9007199254740992.0);
        ++ctr;
        return u;
    }
};

// Positions since the last irreversible move (python-chess is_repetition / can_claim_threefold_repetition
// walk the move stack back to the first irreversible move).
struct RepWindow {
    std::vector<uint64_t> keys;       // keys of the positions BEFORE each trailing reversible move
    void push(const Pos& before, Move m) {
        if (irreversible(before, m)) keys.clear();
        else keys.push_back(tkey(before));
    }
    int count(uint64_t k) const {
        int c = 0;
        for (uint64_t x : keys) c += x == k ? 1 : 0;
        return c;
    }
    bool is_repetition(const Pos& cur, int n) const { return 1 + count(tkey(cur)) >= n; }
    bool can_claim_threefold(const Pos& cur) const {
        const uint64_t k = tkey(cur);
        if (1 + count(k) >= 3) return true;
        Move mv[M0_MAX_MOVES];
        const int n = gen_legal(cur, mv);
        for (int i = 0; i < n; ++i) {
            Pos q = cur;
            make_move(q, mv[i]);
            const uint64_t k2 = tkey(q);
            if (count(k2) + (k2 == k ? 1 : 0) >= 2) return true;
        }
        return false;
    }
};

inline bool is_fifty(const Pos& p) { return p.halfmove >= 100 && any_legal(p); }
inline bool can_claim_fifty(const Pos& p) {
    if (is_fifty(p)) return true;
    if (p.halfmove >= 99) {
        Move mv[M0_MAX_MOVES];
        const int n = gen_legal(p, mv);
        for (int i = 0; i < n; ++i) {
            const int from = mv_from(mv[i]), to = mv_to(mv[i]);
            const bool zeroing = piece_type_at(p, from) == PAWN || (occ_of(p, p.turn ^ 1) & bit(to));
            if (!zeroing) {
                Pos q = p;
                make_move(q, mv[i]);
                if (is_fifty(q)) return true;
            }
        }
    }
    return false;
}

// Board.is_game_over(claim_draw) -> outcome() order
inline bool is_game_over(const Pos& p, const RepWindow& w, bool claim_draw) {
    const bool anyl = any_legal(p);
    if (!anyl) return true;                       // checkmate or stalemate
    if (is_insufficient(p)) return true;
    if (claim_draw) {
        if (can_claim_fifty(p)) return true;
        if (w.can_claim_threefold(p)) return true;
    }
    if (p.halfmove >= 150) return true;           // seventy-five moves (legal moves exist)
    if (w.is_repetition(p, 5)) return true;
    return false;
}

// game_result (internal.py:738-750): checkmate -> loser is the side to move; every other end is 0
inline float game_result(const Pos& p) {
    if (!any_legal(p) && in_check(p)) return p.turn == WHITE ? -1.0f : 1.0f;
    return 0.0f;
}

struct DrawCfg {
    bool enabled, stalemate_draw;
    int min_plies, window, min_unique, halfmove_cap, material_threshold;
};
inline DrawCfg draw_cfg_from(const m0_selfplay_cfg& c) {
    DrawCfg d;
    d.enabled = c.draw_enabled != 0; d.stalemate_draw = c.draw_stalemate != 0;
    d.min_plies = c.draw_min_plies; d.window = c.draw_window; d.min_unique = c.draw_min_unique;
    d.halfmove_cap = c.draw_halfmove_cap; d.material_threshold = c.draw_material_threshold;
    return d;
}

inline bool should_adjudicate_draw(const Pos& p, const RepWindow& w, const std::vector<Move>& moves, const DrawCfg& c) {
    if (is_insufficient(p)) return true;
    if (can_claim_fifty(p)) return true;
    if (w.is_repetition(p, 3) || w.can_claim_threefold(p)) return true;
    if (c.stalemate_draw && !any_legal(p) && !in_check(p)) return true;
    if (!c.enabled) return false;
    if ((int)moves.size() < c.min_plies) return false;
    if (c.window > 0 && c.min_unique > 0 && (int)moves.size() >= c.window) {
        std::set<Move> uniq(moves.end() - c.window, moves.end());
        if ((int)uniq.size() < c.min_unique) return true;
    }
    if (c.halfmove_cap && p.halfmove >= c.halfmove_cap) return true;
    if (c.material_threshold > 0) {
        int mat = 0;
        mat += popc(p.bb[PAWN]) + 3 * popc(p.bb[KNIGHT]) + 3 * popc(p.bb[BISHOP]) + 5 * popc(p.bb[ROOK]) + 9 * popc(p.bb[QUEEN]);
        if (mat <= c.material_threshold) return true;
    }
    return false;
}

// mcts.py:378-387 with random.randint(low, high) = low + floor(u * (high - low + 1))
inline int playout_cap(int sims, double frac, double u) {
    if (frac > 0.0 && sims > 0) {
        int low = (int)fmax(1.0, sims * (1.0 - frac));
        int high = (int)fmax((double)low, sims * (1.0 + frac));
        int k = (int)(u * (double)(high - low + 1));
        if (k > high - low) k = high - low;
        return low + k;
    }
    return sims;
}

inline double temperature_for(int fullmove, double t_start, double t_end, int t_moves) {
    if (t_moves <= 0) return t_end;
    int m = fullmove < 0 ? 0 : (fullmove > t_moves ? t_moves : fullmove);
    double t = (double)m / (double)(t_moves > 1 ? t_moves : 1);
    return t_start + (t_end - t_start) * t;
}

// numpy float32 add.reduce: pairwise with an 8-way unrolled base case (blocks of <= 128)
inline float np_sum_f32(const float* a, int n) {
    if (n < 8) {
        float r = 0.f;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    if (n <= 128) {
        float r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_sum_f32(a, n2) + np_sum_f32(a + n2, n - n2);
}

// sample_move_from_counts (internal.py:690-735); np.random.choice(p=...) -> inverse CDF on u
inline int sample_move_index(const int32_t* visits, int n, double temperature, double u) {
    bool allzero = true;
    for (int i = 0; i < n; ++i) if (visits[i] != 0) allzero = false;
    auto uniform_pick = [&]() { int k = (int)(u * n); return k >= n ? n - 1 : k; };
    if (allzero) return uniform_pick();
    if (temperature < 1e-3) {
        int best = 0;
        for (int i = 1; i < n; ++i) if (visits[i] > visits[best]) best = i;
        return best;
    }
    std::vector<float> d(n);
    const float ex = (float)(1.0 / temperature);
    for (int i = 0; i < n; ++i) d[i] = powf((float)visits[i], ex);
    const float s = np_sum_f32(d.data(), n);
    if (!(s > 0.f) || isnan(s)) return uniform_pick();
    double cum = 0.0;
    std::vector<double> cdf(n);
    bool anynan = false;
    for (int i = 0; i < n; ++i) { d[i] = d[i] / s; anynan = anynan || isnan(d[i]); cum += (double)d[i]; cdf[i] = cum; }
    if (anynan) return uniform_pick();            // internal.py:727-732: overflowed powers (inf / inf)
    const double last = cdf[n - 1];
    for (int i = 0; i < n; ++i)
        if (cdf[i] / last > u) return i;
    return n - 1;
}
// whether sample_move_from_counts consumes a random draw: every branch but the arg-max one (internal.py:706-708)
inline bool sample_move_draws(const int32_t* visits, int n, double temperature) {
    bool allzero = true;
    for (int i = 0; i < n; ++i) if (visits[i] != 0) allzero = false;
    return allzero || !(temperature < 1e-3);
}

// arena move choice (arena.py:73-86, 106): float32 logits = log(visits + 1e-8) / max(temp, 1e-3), softmax, then
// np.random.choice(p=probs) = inverse CDF on u; temp <= 1e-3 or ply >= temp_plies: most visited, first maximum
inline int arena_choose_move(const int32_t* visits, int n, double temp, int ply, int temp_plies, double u) {
    int best = 0;
    for (int i = 1; i < n; ++i) if (visits[i] > visits[best]) best = i;
    if (!(temp > 1e-3 && ply < temp_plies)) return best;
    std::vector<float> lg(n), pr(n);
    const float t = (float)(temp > 1e-3 ? temp : 1e-3);
    float mx = -INFINITY;
    for (int i = 0; i < n; ++i) { lg[i] = logf((float)visits[i] + 1e-8f) / t; if (lg[i] > mx) mx = lg[i]; }
    for (int i = 0; i < n; ++i) pr[i] = expf(lg[i] - mx);
    const float s = np_sum_f32(pr.data(), n);
    if (!(s > 0.f) || !std::isfinite(s)) return best;
    double cum = 0.0;
    std::vector<double> cdf(n);
    for (int i = 0; i < n; ++i) { pr[i] = pr[i] / s; cum += (double)pr[i]; cdf[i] = cum; }
    const double last = cdf[n - 1];
    for (int i = 0; i < n; ++i)
        if (cdf[i] / last > u) return i;
    return n - 1;
}

struct ResignState {
    int consec_bad = 0;
    std::vector<double> recent_values, recent_entropies;
};
// internal.py:507-536
inline bool resign_update(ResignState& st, double v, int n_states, const m0_selfplay_cfg& c) {
    if (!(c.resign_threshold > -1.0 && n_states >= c.min_resign_plies)) return false;
    st.recent_values.push_back(v);
    if ((int)st.recent_values.size() > c.resign_window) st.recent_values.erase(st.recent_values.begin());
    if (v < c.resign_threshold) st.consec_bad++; else st.consec_bad = 0;
    const int need = c.resign_window / 2 > 2 ? c.resign_window / 2 : 2;
    bool stable_bad = false, low_unc = false;
    if ((int)st.recent_values.size() >= need) {
        double s = 0; for (double x : st.recent_values) s += x;
        stable_bad = s / (double)st.recent_values.size() < c.resign_threshold + c.resign_value_margin;
    }
    if ((int)st.recent_entropies.size() >= need) {
        double s = 0; for (double x : st.recent_entropies) s += x;
        low_unc = s / (double)st.recent_entropies.size() < c.resign_min_entropy;
    }
    return st.consec_bad >= c.resign_consecutive_bad && (stable_bad || low_unc);
}

}  // namespace m0
