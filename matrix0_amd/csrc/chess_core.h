// Bitboard chess core shared by the device kernels (tree walk / expand / encode) and the
// host game loop.  Header-only, __host__ __device__.
//
// Semantics follow python-chess as the reference uses it: cleaned castling
// rights, ep square set after every double push, legal-move generation ORDER
// (Board.generate_legal_moves), insufficient-material rule, transposition key contents.
// Encoding follows azchess/encoding.py: encode_board (11-46), move_to_index (80-150).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define M0_HD __host__ __device__ inline
#else
#define M0_HD inline
#endif

namespace m0 {

enum { PAWN = 0, KNIGHT = 1, BISHOP = 2, ROOK = 3, QUEEN = 4, KING = 5 };
enum { BLACK = 0, WHITE = 1 };

// castling-right bits (raw; cleaning applied on read)
enum { CR_WK = 1, CR_WQ = 2, CR_BK = 4, CR_BQ = 8 };

struct Pos {
    uint64_t bb[6];    // by piece type, both colours
    uint64_t occ[2];   // [BLACK], [WHITE]
    uint8_t turn;      // WHITE = 1
    uint8_t cr;        // raw castling rights
    int8_t ep;         // -1 or square
    uint8_t pad;
    uint16_t halfmove;
    uint16_t fullmove;
};

typedef uint16_t Move;  // from | to<<6 | promo<<12 ; promo: 0 none, 1 N, 2 B, 3 R, 4 Q
#define M0_MAX_MOVES 256

M0_HD Move mk_move(int from, int to, int promo) { return (Move)(from | (to << 6) | (promo << 12)); }
M0_HD int mv_from(Move m) { return m & 63; }
M0_HD int mv_to(Move m) { return (m >> 6) & 63; }
M0_HD int mv_promo(Move m) { return (m >> 12) & 7; }

M0_HD int msb(uint64_t b) { return 63 - __builtin_clzll(b); }
M0_HD int lsb(uint64_t b) { return __builtin_ctzll(b); }
M0_HD int popc(uint64_t b) { return __builtin_popcountll(b); }
M0_HD uint64_t bit(int s) { return 1ull << s; }

constexpr uint64_t FILE_A = 0x0101010101010101ull;
constexpr uint64_t FILE_H = 0x8080808080808080ull;
constexpr uint64_t RANK_1 = 0xFFull;
constexpr uint64_t RANK_8 = 0xFFull << 56;
constexpr uint64_t DARK_SQ = 0xAA55AA55AA55AA55ull;

M0_HD uint64_t knight_att(int s) {
    uint64_t b = bit(s);
    uint64_t l1 = (b >> 1) & ~FILE_H, l2 = (b >> 2) & ~(FILE_H | (FILE_H >> 1));
    uint64_t r1 = (b << 1) & ~FILE_A, r2 = (b << 2) & ~(FILE_A | (FILE_A << 1));
    uint64_t h1 = l1 | r1, h2 = l2 | r2;
    return (h1 << 16) | (h1 >> 16) | (h2 << 8) | (h2 >> 8);
}
M0_HD uint64_t king_att(int s) {
    uint64_t b = bit(s);
    uint64_t a = ((b << 1) & ~FILE_A) | ((b >> 1) & ~FILE_H);
    uint64_t row = a | b;
    return a | (row << 8) | (row >> 8);
}
// squares attacked BY a pawn of colour c standing on s
M0_HD uint64_t pawn_att(int s, int c) {
    uint64_t b = bit(s);
    if (c == WHITE) return ((b << 7) & ~FILE_H) | ((b << 9) & ~FILE_A);
    return ((b >> 7) & ~FILE_A) | ((b >> 9) & ~FILE_H);
}
// Sliding attacks by "hyperbola quintessence": along one line (mask m without the slider's square, slider bit r) the
// attacked squares are ((o - 2r) ^ rev(rev(o) - 2 rev(r))) & m with o = occupancy & m and rev = 64-bit bit reversal --
// a dozen 64-bit operations per line instead of a loop over up to seven squares per direction (the tree kernels spend
// most of their instructions in these: legality test of every pseudo-legal move, check tests, slider targets).  Same
// sets as stepping along the rays up to and including the first blocker.
M0_HD uint64_t rev64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brevll(x);
#else
    x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    return __builtin_bswap64(x);
#endif
}
M0_HD uint64_t line_att(uint64_t occ, uint64_t r, uint64_t m) {
    const uint64_t o = occ & m;
    const uint64_t fwd = o - 2 * r;
    const uint64_t bwd = rev64(rev64(o) - 2 * rev64(r));
    return (fwd ^ bwd) & m;
}
M0_HD uint64_t diag_mask(int s) {          // a1-h8 direction through s
    const int d = 8 * (s & 7) - (s & 56);
    const int nort = -d & (d >> 31), sout = d & (-d >> 31);
    return (0x8040201008040201ull >> sout) << nort;
}
M0_HD uint64_t anti_mask(int s) {          // h1-a8 direction through s
    const int d = 56 - 8 * (s & 7) - (s & 56);
    const int nort = -d & (d >> 31), sout = d & (-d >> 31);
    return (0x0102040810204080ull >> sout) << nort;
}
M0_HD uint64_t bishop_att(int s, uint64_t occ) {
    const uint64_t r = bit(s);
    return line_att(occ, r, diag_mask(s) ^ r) | line_att(occ, r, anti_mask(s) ^ r);
}
M0_HD uint64_t rook_att(int s, uint64_t occ) {
    const uint64_t r = bit(s);
    return line_att(occ, r, (FILE_A << (s & 7)) ^ r) | line_att(occ, r, (RANK_1 << (s & 56)) ^ r);
}

M0_HD uint64_t occ_all(const Pos& p) { return p.occ[0] | p.occ[1]; }
// Colour- / type-indexed access without run-time array indexing: with p.occ[c] / p.bb[t] and a run-time index hipcc
// keeps the whole position in scratch memory on the device (every access a memory round trip in the tree kernels).
M0_HD uint64_t occ_of(const Pos& p, int c) { return c ? p.occ[1] : p.occ[0]; }
M0_HD void occ_and(Pos& p, int c, uint64_t m) { p.occ[0] &= c ? ~0ull : m; p.occ[1] &= c ? m : ~0ull; }
M0_HD void occ_or(Pos& p, int c, uint64_t m) { p.occ[0] |= c ? 0ull : m; p.occ[1] |= c ? m : 0ull; }
M0_HD void bb_and(Pos& p, int t, uint64_t m) {
    p.bb[0] &= t == 0 ? m : ~0ull; p.bb[1] &= t == 1 ? m : ~0ull; p.bb[2] &= t == 2 ? m : ~0ull;
    p.bb[3] &= t == 3 ? m : ~0ull; p.bb[4] &= t == 4 ? m : ~0ull; p.bb[5] &= t == 5 ? m : ~0ull;
}
M0_HD void bb_or(Pos& p, int t, uint64_t m) {
    p.bb[0] |= t == 0 ? m : 0ull; p.bb[1] |= t == 1 ? m : 0ull; p.bb[2] |= t == 2 ? m : 0ull;
    p.bb[3] |= t == 3 ? m : 0ull; p.bb[4] |= t == 4 ? m : 0ull; p.bb[5] |= t == 5 ? m : 0ull;
}

M0_HD bool attacked(const Pos& p, int s, int by) {
    const uint64_t them = occ_of(p, by);
    if (pawn_att(s, by ^ 1) & p.bb[PAWN] & them) return true;
    if (knight_att(s) & p.bb[KNIGHT] & them) return true;
    if (king_att(s) & p.bb[KING] & them) return true;
    const uint64_t o = occ_all(p);
    if (bishop_att(s, o) & (p.bb[BISHOP] | p.bb[QUEEN]) & them) return true;
    if (rook_att(s, o) & (p.bb[ROOK] | p.bb[QUEEN]) & them) return true;
    return false;
}

M0_HD int king_sq(const Pos& p, int c) {
    uint64_t k = p.bb[KING] & occ_of(p, c);
    return k ? msb(k) : -1;
}
M0_HD bool in_check(const Pos& p) {
    int k = king_sq(p, p.turn);
    return k >= 0 && attacked(p, k, p.turn ^ 1);
}
M0_HD int piece_type_at(const Pos& p, int s) {
    uint64_t b = bit(s);
    // (branch-free, constant indices)
    int r = -1;
    r = (p.bb[5] & b) ? 5 : r; r = (p.bb[4] & b) ? 4 : r; r = (p.bb[3] & b) ? 3 : r;
    r = (p.bb[2] & b) ? 2 : r; r = (p.bb[1] & b) ? 1 : r; r = (p.bb[0] & b) ? 0 : r;
    return r;
}

// clean_castling_rights(): right needs its rook on the corner and the king on e1/e8
M0_HD int clean_cr(const Pos& p) {
    int out = 0;
    const uint64_t wr = p.bb[ROOK] & p.occ[WHITE], br = p.bb[ROOK] & p.occ[BLACK];
    if (p.bb[KING] & p.occ[WHITE] & bit(4)) {
        if ((p.cr & CR_WK) && (wr & bit(7))) out |= CR_WK;
        if ((p.cr & CR_WQ) && (wr & bit(0))) out |= CR_WQ;
    }
    if (p.bb[KING] & p.occ[BLACK] & bit(60)) {
        if ((p.cr & CR_BK) && (br & bit(63))) out |= CR_BK;
        if ((p.cr & CR_BQ) && (br & bit(56))) out |= CR_BQ;
    }
    return out;
}

M0_HD int cr_mask_for_square(int s) {
    switch (s) {
        case 0: return CR_WQ; case 7: return CR_WK; case 56: return CR_BQ; case 63: return CR_BK;
        default: return 0;
    }
}

M0_HD void make_move(Pos& p, Move m) {
    const int from = mv_from(m), to = mv_to(m), promo = mv_promo(m);
    const int us = p.turn, them = us ^ 1;
    const uint64_t fb = bit(from), tb = bit(to);
    const int type = piece_type_at(p, from);
    const bool capture = (occ_of(p, them) & tb) != 0;
    const bool zeroing = type == PAWN || capture;
    const int ep_old = p.ep;
    p.ep = -1;
    p.cr &= ~(cr_mask_for_square(from) | cr_mask_for_square(to));
    if (type == KING) p.cr &= us == WHITE ? ~(CR_WK | CR_WQ) : ~(CR_BK | CR_BQ);
    if (capture) {
        p.bb[0] &= ~tb; p.bb[1] &= ~tb; p.bb[2] &= ~tb; p.bb[3] &= ~tb; p.bb[4] &= ~tb; p.bb[5] &= ~tb;
        occ_and(p, them, ~tb);
    }
    bb_and(p, type, ~fb);
    occ_and(p, us, ~fb);
    if (type == PAWN) {
        const int diff = to - from;
        if (diff == 16 && (from >> 3) == 1) p.ep = (int8_t)(from + 8);
        else if (diff == -16 && (from >> 3) == 6) p.ep = (int8_t)(from - 8);
        else if (to == ep_old && !capture && (diff == 7 || diff == 9 || diff == -7 || diff == -9)) {
            const uint64_t cb = bit(to + (us == WHITE ? -8 : 8));
            p.bb[PAWN] &= ~cb;
            occ_and(p, them, ~cb);
        }
    }
    if (type == KING && ((to & 7) - (from & 7) == 2 || (to & 7) - (from & 7) == -2)) {
        const int r = from & 56;
        uint64_t rf, rt;
        if ((to & 7) == 6) { rf = bit(r + 7); rt = bit(r + 5); } else { rf = bit(r); rt = bit(r + 3); }
        p.bb[ROOK] = (p.bb[ROOK] & ~rf) | rt;
        occ_and(p, us, ~rf); occ_or(p, us, rt);
    }
    const int placed = promo ? promo : type;   // promo codes 1..4 == KNIGHT..QUEEN
    bb_or(p, placed, tb);
    occ_or(p, us, tb);
    p.halfmove = zeroing ? 0 : (uint16_t)(p.halfmove + 1);
    if (us == BLACK) p.fullmove = (uint16_t)(p.fullmove + 1);
    p.turn = (uint8_t)them;
}

M0_HD bool legal_after(const Pos& p, Move m) {
    Pos q = p;
    make_move(q, m);
    int k = king_sq(q, p.turn);
    return k < 0 || !attacked(q, k, p.turn ^ 1);
}

template <bool LEGAL>
M0_HD int emit(const Pos& p, Move* out, int n, int from, int to, int promo) {
    Move m = mk_move(from, to, promo);
    if (!LEGAL || legal_after(p, m)) out[n++] = m;
    return n;
}
template <bool LEGAL>
M0_HD int emit_promos(const Pos& p, Move* out, int n, int from, int to) {
    if ((to >> 3) == 0 || (to >> 3) == 7) {
        n = emit<LEGAL>(p, out, n, from, to, 4); n = emit<LEGAL>(p, out, n, from, to, 3);
        n = emit<LEGAL>(p, out, n, from, to, 2); n = emit<LEGAL>(p, out, n, from, to, 1);
        return n;
    }
    return emit<LEGAL>(p, out, n, from, to, 0);
}
M0_HD uint64_t piece_targets(const Pos& p, int from, int type) {
    const uint64_t o = occ_all(p);
    uint64_t a;
    switch (type) {
        case KNIGHT: a = knight_att(from); break;
        case BISHOP: a = bishop_att(from, o); break;
        case ROOK: a = rook_att(from, o); break;
        case QUEEN: a = bishop_att(from, o) | rook_att(from, o); break;
        default: a = king_att(from); break;
    }
    return a & ~occ_of(p, p.turn);
}
template <bool LEGAL>
M0_HD int emit_piece(const Pos& p, Move* out, int n, int from) {
    uint64_t t = piece_targets(p, from, piece_type_at(p, from));
    while (t) { int to = msb(t); t &= ~bit(to); n = emit<LEGAL>(p, out, n, from, to, 0); }
    return n;
}

// Moves in python-chess generation order (see oracle/chess_oracle.c header).  LEGAL = true: the legal moves.
// LEGAL = false: the pseudo-legal superset in the same order (castling entries already fully checked); filtering it
// with legal_after() move by move -- which the device does one move per lane -- gives exactly the LEGAL = true list.
template <bool LEGAL>
M0_HD int gen_moves(const Pos& p, Move* out) {
    int n = 0;
    const int us = p.turn, them = us ^ 1;
    const uint64_t own = occ_of(p, us), o = occ_all(p);
    const int ksq = king_sq(p, us);
    const bool chk = ksq >= 0 && attacked(p, ksq, them);
    if (chk) n = emit_piece<LEGAL>(p, out, n, ksq);
    uint64_t pcs = own & ~p.bb[PAWN];
    if (chk && ksq >= 0) pcs &= ~bit(ksq);
    while (pcs) { int s = msb(pcs); pcs &= ~bit(s); n = emit_piece<LEGAL>(p, out, n, s); }
    if (!chk && ksq >= 0) {
        const int cr = clean_cr(p);
        const int base = us == WHITE ? 0 : 56;
        if (ksq == base + 4) {
            const int kbit = us == WHITE ? CR_WK : CR_BK, qbit = us == WHITE ? CR_WQ : CR_BQ;
            if ((cr & kbit) && !(o & (bit(base + 5) | bit(base + 6))) && !attacked(p, base + 5, them) &&
                !attacked(p, base + 6, them))
                out[n++] = mk_move(ksq, base + 6, 0);
            if ((cr & qbit) && !(o & (bit(base + 1) | bit(base + 2) | bit(base + 3))) && !attacked(p, base + 3, them) &&
                !attacked(p, base + 2, them))
                out[n++] = mk_move(ksq, base + 2, 0);
        }
    }
    const uint64_t pawns = p.bb[PAWN] & own;
    uint64_t c = pawns;
    while (c) {
        int s = msb(c); c &= ~bit(s);
        uint64_t t = pawn_att(s, us) & occ_of(p, them);
        while (t) { int to = msb(t); t &= ~bit(to); n = emit_promos<LEGAL>(p, out, n, s, to); }
    }
    uint64_t single = (us == WHITE ? pawns << 8 : pawns >> 8) & ~o;
    uint64_t dbl = (us == WHITE ? single << 8 : single >> 8) & ~o & (us == WHITE ? (RANK_1 << 24) : (RANK_1 << 32));
    while (single) {
        int to = msb(single); single &= ~bit(to);
        n = emit_promos<LEGAL>(p, out, n, to + (us == WHITE ? -8 : 8), to);
    }
    while (dbl) {
        int to = msb(dbl); dbl &= ~bit(to);
        n = emit<LEGAL>(p, out, n, to + (us == WHITE ? -16 : 16), to, 0);
    }
    if (p.ep >= 0 && !(o & bit(p.ep))) {
        uint64_t cap = pawns & pawn_att(p.ep, them) & (us == WHITE ? (RANK_1 << 32) : (RANK_1 << 24));
        while (cap) { int s = msb(cap); cap &= ~bit(s); n = emit<LEGAL>(p, out, n, s, p.ep, 0); }
    }
    return n;
}
M0_HD int gen_legal(const Pos& p, Move* out) { return gen_moves<true>(p, out); }
M0_HD int gen_pseudo(const Pos& p, Move* out) { return gen_moves<false>(p, out); }

M0_HD bool has_legal_ep(const Pos& p) {
    if (p.ep < 0 || (occ_all(p) & bit(p.ep))) return false;
    const int us = p.turn;
    uint64_t cap = p.bb[PAWN] & p.occ[us] & pawn_att(p.ep, us ^ 1) & (us == WHITE ? (RANK_1 << 32) : (RANK_1 << 24));
    while (cap) {
        int s = msb(cap); cap &= ~bit(s);
        if (legal_after(p, mk_move(s, p.ep, 0))) return true;
    }
    return false;
}

M0_HD bool any_legal(const Pos& p) {
    Move mv[M0_MAX_MOVES];
    return gen_legal(p, mv) > 0;
}

M0_HD bool insufficient_side(const Pos& p, int c) {
    const uint64_t own = occ_of(p, c);
    if (own & (p.bb[PAWN] | p.bb[ROOK] | p.bb[QUEEN])) return false;
    if (own & p.bb[KNIGHT])
        return popc(own) <= 2 && !(occ_of(p, c ^ 1) & ~p.bb[KING] & ~p.bb[QUEEN]);
    if (own & p.bb[BISHOP]) {
        const bool same = !(p.bb[BISHOP] & DARK_SQ) || !(p.bb[BISHOP] & ~DARK_SQ);
        return same && !p.bb[PAWN] && !p.bb[KNIGHT];
    }
    return true;
}
M0_HD bool is_insufficient(const Pos& p) { return insufficient_side(p, WHITE) && insufficient_side(p, BLACK); }

// 64-bit key over the _transposition_key() contents (piece placement, turn, cleaned castling
// rights, legal ep square).  Equality of keys stands in for tuple equality (2^-64 collisions).
M0_HD uint64_t mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}
M0_HD uint64_t tkey(const Pos& p) {
    uint64_t h = 0x9E3779B97F4A7C15ull;
    for (int t = 0; t < 6; ++t) h = mix64(h ^ p.bb[t]) + 0x9E3779B97F4A7C15ull * (uint64_t)(t + 1);
    h = mix64(h ^ p.occ[WHITE]);
    h = mix64(h ^ p.occ[BLACK] ^ 0x5851F42D4C957F2Dull);
    uint64_t tail = (uint64_t)p.turn | ((uint64_t)clean_cr(p) << 1) | ((uint64_t)(has_legal_ep(p) ? (p.ep + 1) : 0) << 8);
    return mix64(h ^ (tail * 0xD6E8FEB86659FD93ull));
}

// is_irreversible(move) on the position BEFORE the move
M0_HD bool irreversible(const Pos& before, Move m) {
    const int from = mv_from(m), to = mv_to(m);
    const int type = piece_type_at(before, from);
    if (type == PAWN || (occ_of(before, before.turn ^ 1) & bit(to))) return true;
    const int cr0 = clean_cr(before);
    int cr1 = cr0 & ~(cr_mask_for_square(from) | cr_mask_for_square(to));
    if (type == KING) cr1 &= before.turn == WHITE ? ~(CR_WK | CR_WQ) : ~(CR_BK | CR_BQ);
    if (cr1 != cr0) return true;
    return has_legal_ep(before);
}

// ---- azchess/encoding.py ----
// move_to_index (encoding.py:80-150) for a legal move; -1 if unmappable
M0_HD int move_to_index(const Pos& p, Move m) {
    const int from = mv_from(m), to = mv_to(m), promo = mv_promo(m);
    const int dr = (to >> 3) - (from >> 3), df = (to & 7) - (from & 7);
    const int adr = dr < 0 ? -dr : dr, adf = df < 0 ? -df : df;
    if ((adr == 2 && adf == 1) || (adr == 1 && adf == 2)) {
        // KNIGHT_DELTAS order (-2,-1)(-2,1)(-1,-2)(-1,2)(1,-2)(1,2)(2,-1)(2,1)
        int k;
        if (dr == -2) k = df == -1 ? 0 : 1;
        else if (dr == -1) k = df == -2 ? 2 : 3;
        else if (dr == 1) k = df == -2 ? 4 : 5;
        else k = df == -1 ? 6 : 7;
        return from * 73 + 56 + k;
    }
    if (promo >= 1 && promo <= 3) {   // N,B,R under-promotions; directions side-relative
        int dir = -1;
        if (p.turn == WHITE) { if (dr == 1) dir = df == 0 ? 0 : (df == -1 ? 1 : (df == 1 ? 2 : -1)); }
        else { if (dr == -1) dir = df == 0 ? 0 : (df == 1 ? 1 : (df == -1 ? 2 : -1)); }
        if (dir >= 0) return from * 73 + 64 + (promo - 1) * 3 + dir;
    }
    if (dr == 0 || df == 0 || adr == adf) {
        const int step = adr > adf ? adr : adf;
        const int sdr = dr == 0 ? 0 : (dr > 0 ? 1 : -1), sdf = df == 0 ? 0 : (df > 0 ? 1 : -1);
        // RAY_DIRS order N S E W NE NW SE SW as (dr,df)
        int d;
        if (sdf == 0) d = sdr > 0 ? 0 : 1;
        else if (sdr == 0) d = sdf > 0 ? 2 : 3;
        else if (sdr > 0) d = sdf > 0 ? 4 : 5;
        else d = sdf > 0 ? 6 : 7;
        if (step >= 1 && step <= 7) return from * 73 + d * 7 + (step - 1);
    }
    return -1;
}

// plane constants 12..18 of encode_board (encoding.py:23-33); 17/18 are float32 of a double quotient
M0_HD void plane_consts(const Pos& p, float* c7) {
    const int cr = clean_cr(p);
    c7[0] = p.turn == WHITE ? 1.f : 0.f;
    c7[1] = (cr & CR_WK) ? 1.f : 0.f;
    c7[2] = (cr & CR_WQ) ? 1.f : 0.f;
    c7[3] = (cr & CR_BK) ? 1.f : 0.f;
    c7[4] = (cr & CR_BQ) ? 1.f : 0.f;
    c7[5] = (float)((double)(p.halfmove < 99 ? p.halfmove : 99) / 99.0);
    c7[6] = (float)((double)(p.fullmove < 199 ? p.fullmove : 199) / 199.0);
}
// plane index 0..11 of the piece on square s, or -1
M0_HD int piece_plane(const Pos& p, int s) {
    const uint64_t b = bit(s);
    if (!((p.occ[0] | p.occ[1]) & b)) return -1;
    const int t = piece_type_at(p, s);
    return (p.occ[WHITE] & b) ? t : 6 + t;
}
// f32 [19][8][8], plane[7-rank][file]
M0_HD void encode_planes_f32(const Pos& p, float* out) {
    for (int i = 0; i < 19 * 64; ++i) out[i] = 0.f;
    for (int s = 0; s < 64; ++s) {
        int pl = piece_plane(p, s);
        if (pl >= 0) out[pl * 64 + (7 - (s >> 3)) * 8 + (s & 7)] = 1.f;
    }
    float c[7];
    plane_consts(p, c);
    for (int i = 0; i < 7; ++i)
        for (int s = 0; s < 64; ++s) out[(12 + i) * 64 + s] = c[i];
}

inline int parse_fen(const char* fen, Pos& p) {
    for (int t = 0; t < 6; ++t) p.bb[t] = 0;
    p.occ[0] = p.occ[1] = 0;
    p.turn = WHITE; p.cr = 0; p.ep = -1; p.pad = 0; p.halfmove = 0; p.fullmove = 1;
    int r = 7, f = 0;
    const char* c = fen;
    while (*c && *c != ' ') {
        if (*c == '/') { r--; f = 0; }
        else if (*c >= '1' && *c <= '8') f += *c - '0';
        else {
            const char* syms = "PNBRQKpnbrqk";
            int idx = -1;
            for (int i = 0; i < 12; ++i) if (syms[i] == *c) idx = i;
            if (idx < 0 || r < 0 || f > 7) return -1;
            p.bb[idx % 6] |= bit(r * 8 + f);
            p.occ[idx < 6 ? WHITE : BLACK] |= bit(r * 8 + f);
            f++;
        }
        c++;
    }
    if (*c == ' ') c++;
    if (*c == 'w') p.turn = WHITE; else if (*c == 'b') p.turn = BLACK; else return -1;
    c++;
    if (*c == ' ') c++;
    while (*c && *c != ' ') {
        if (*c == 'K') p.cr |= CR_WK; else if (*c == 'Q') p.cr |= CR_WQ;
        else if (*c == 'k') p.cr |= CR_BK; else if (*c == 'q') p.cr |= CR_BQ;
        c++;
    }
    if (*c == ' ') c++;
    if (*c && *c != '-' && c[1]) { p.ep = (int8_t)((c[1] - '1') * 8 + (c[0] - 'a')); c += 2; }
    else if (*c) c++;
    if (*c == ' ') c++;
    if (*c) { int v = 0; while (*c >= '0' && *c <= '9') { v = v * 10 + (*c - '0'); c++; } p.halfmove = (uint16_t)(v > 65535 ? 65535 : v); if (*c == ' ') c++; }
    if (*c) { int v = 0; while (*c >= '0' && *c <= '9') { v = v * 10 + (*c - '0'); c++; } p.fullmove = (uint16_t)(v > 65535 ? 65535 : v); }
    // rights without a rook on the corner never come back (a rook arriving there clears the bit)
    const uint64_t wr = p.bb[ROOK] & p.occ[WHITE], br = p.bb[ROOK] & p.occ[BLACK];
    if (!(wr & bit(7))) p.cr &= ~CR_WK;
    if (!(wr & bit(0))) p.cr &= ~CR_WQ;
    if (!(br & bit(63))) p.cr &= ~CR_BK;
    if (!(br & bit(56))) p.cr &= ~CR_BQ;
    return 0;
}

inline Move parse_uci(const char* u) {
    if (!u || !u[0] || !u[1] || !u[2] || !u[3]) return 0xFFFF;
    int from = (u[0] - 'a') + 8 * (u[1] - '1'), to = (u[2] - 'a') + 8 * (u[3] - '1');
    int promo = 0;
    if (u[4] == 'n') promo = 1; else if (u[4] == 'b') promo = 2; else if (u[4] == 'r') promo = 3; else if (u[4] == 'q') promo = 4;
    if (from < 0 || from > 63 || to < 0 || to > 63) return 0xFFFF;
    return mk_move(from, to, promo);
}

}  // namespace m0
