// TEST INFRASTRUCTURE: host (g++) build of matrix0_amd/csrc/chess_core.h so the bitboard core the
// device kernels use can be cross-checked against the oracle on the CPU (perft, move order,
// indices, planes) without spending GPU time.  Not part of libm0engine.so.
#include <string.h>
#include "../../matrix0_amd/csrc/chess_core.h"
using namespace m0;

static uint64_t perft(const Pos& p, int d) {
    Move mv[M0_MAX_MOVES];
    int n = gen_legal(p, mv);
    if (d <= 1) return d == 1 ? (uint64_t)n : 1;
    uint64_t t = 0;
    for (int i = 0; i < n; ++i) { Pos q = p; make_move(q, mv[i]); t += perft(q, d - 1); }
    return t;
}

extern "C" {
uint64_t hc_perft(const char* fen, int depth) { Pos p; if (parse_fen(fen, p)) return ~0ull; return perft(p, depth); }

// legal moves in order: out_mv[i] = from | to<<8 | promo<<16 (promo 0 or python-chess type 2..5), out_idx = policy index
int hc_legal(const char* fen, int32_t* out_mv, int32_t* out_idx) {
    Pos p; if (parse_fen(fen, p)) return -1;
    Move mv[M0_MAX_MOVES];
    int n = gen_legal(p, mv);
    for (int i = 0; i < n; ++i) {
        int pr = mv_promo(mv[i]);
        out_mv[i] = mv_from(mv[i]) | (mv_to(mv[i]) << 8) | ((pr ? pr + 1 : 0) << 16);
        out_idx[i] = move_to_index(p, mv[i]);
    }
    return n;
}
// the device's two-phase generator (pseudo-legal list in order, then per-move legality filter) run sequentially
int hc_legal_two_phase(const char* fen, int32_t* out_mv) {
    Pos p; if (parse_fen(fen, p)) return -1;
    Move pm[M0_MAX_MOVES];
    int np = gen_pseudo(p, pm), n = 0;
    for (int i = 0; i < np; ++i)
        if (legal_after(p, pm[i])) { int pr = mv_promo(pm[i]); out_mv[n++] = mv_from(pm[i]) | (mv_to(pm[i]) << 8) | ((pr ? pr + 1 : 0) << 16); }
    return n;
}
int hc_encode(const char* fen, float* out) { Pos p; if (parse_fen(fen, p)) return -1; encode_planes_f32(p, out); return 0; }

// play uci moves from fen; report state: out[0]=in_check out[1]=insufficient out[2]=clean_cr out[3]=has_legal_ep
// out[4]=halfmove out[5]=fullmove out[6]=ep out[7]=turn ; keys[i] = tkey before move i (n+1 entries), irr[i]
int hc_play(const char* fen, const char* const* ucis, int n, int32_t* out, uint64_t* keys, int32_t* irr, float* planes) {
    Pos p; if (parse_fen(fen, p)) return -1;
    for (int i = 0; i < n; ++i) {
        Move m = parse_uci(ucis[i]);
        Move mv[M0_MAX_MOVES];
        int k = gen_legal(p, mv);
        bool ok = false;
        for (int j = 0; j < k; ++j) if (mv[j] == m) ok = true;
        if (!ok) return -(i + 2);
        if (keys) keys[i] = tkey(p);
        if (irr) irr[i] = irreversible(p, m) ? 1 : 0;
        make_move(p, m);
    }
    if (keys) keys[n] = tkey(p);
    out[0] = in_check(p); out[1] = is_insufficient(p); out[2] = clean_cr(p); out[3] = has_legal_ep(p);
    out[4] = p.halfmove; out[5] = p.fullmove; out[6] = p.ep; out[7] = p.turn;
    if (planes) encode_planes_f32(p, planes);
    return 0;
}
}
