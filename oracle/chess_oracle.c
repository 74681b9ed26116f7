/* ORACLE (test infrastructure, not product): plain-C chess rules used as the CPU
 * checker for the device bitboard core.  Deliberately a different algorithm
 * (8x8 mailbox + make/unmake legality) so that agreement is meaningful.
 *
 * The reference delegates all rules to the third-party python-chess
 * (requirements.txt:4 `python-chess[syzygy]>=1.9.0`, not vendored, not installed
 * here), so this file restates python-chess's published algorithm for the calls
 * the reference makes:
 *   legal_moves (and its generation ORDER), is_legal, push, clean castling rights,
 *   has_{king,queen}side_castling_rights, _transposition_key, is_checkmate,
 *   is_stalemate, is_insufficient_material, is_seventyfive_moves,
 *   is_fivefold_repetition / is_repetition(n), can_claim_fifty_moves,
 *   can_claim_threefold_repetition, is_game_over, result(claim_draw=True).
 * Pinning: perft known answers + the reference's own fixtures
 * (data/tactical/tactical_metadata.json legal-move counts, tests/test_encoding.py,
 * tests/test_board_tensor.py, azchess/validate_moves.py) -- tests/test_oracle_chess.py.
 * Legal-move ORDER and key contents are pinned by no reference test ("parity unpinned").
 *
 * Move order restated (python-chess Board.generate_legal_moves):
 *   in check:  king moves (to-square descending) first, then the list below without the king
 *   otherwise: non-pawn pieces by from-square descending, each to-square descending;
 *              castling (h-side then a-side); pawn captures by from descending, to descending
 *              (promotions Q,R,B,N); single pushes by to descending (promotions Q,R,B,N);
 *              double pushes by to descending; en passant by capturer descending.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#define EMPTY 0
/* piece = color*6 + type, type 1..6 = P N B R Q K ; white 1..6, black 7..12 */
#define PT(p) (((p) - 1) % 6 + 1)
#define IS_WHITE(p) ((p) >= 1 && (p) <= 6)
#define IS_BLACK(p) ((p) >= 7)
#define COLOR_OF(p) (IS_WHITE(p) ? 1 : 0) /* python-chess: WHITE = True */

typedef struct {
    int8_t sq[64];      /* a1 = 0 ... h8 = 63 */
    int8_t turn;        /* 1 white, 0 black */
    uint64_t castling;  /* bitboard of rook squares with castling rights (raw, un-cleaned) */
    int8_t ep;          /* -1 or ep square (set after every double push) */
    int32_t halfmove;
    int32_t fullmove;
} OPos;

typedef struct {
    uint8_t from, to, promo; /* promo: 0 or piece type 2..5 */
} OMove;

#define MAX_MOVES 256

static int file_of(int s) { return s & 7; }
static int rank_of(int s) { return s >> 3; }

static const int KN_D[8][2] = {{-2, -1}, {-2, 1}, {-1, -2}, {-1, 2}, {1, -2}, {1, 2}, {2, -1}, {2, 1}};
static const int KG_D[8][2] = {{-1, -1}, {-1, 0}, {-1, 1}, {0, -1}, {0, 1}, {1, -1}, {1, 0}, {1, 1}};
static const int BI_D[4][2] = {{-1, -1}, {-1, 1}, {1, -1}, {1, 1}};
static const int RO_D[4][2] = {{-1, 0}, {1, 0}, {0, -1}, {0, 1}};

/* is square s attacked by side `by` (1 white / 0 black)? */
static int attacked(const OPos* p, int s, int by) {
    int r = rank_of(s), f = file_of(s), i, k;
    /* pawns */
    int pr = by ? r - 1 : r + 1;
    if (pr >= 0 && pr < 8) {
        for (k = -1; k <= 1; k += 2) {
            int pf = f + k;
            if (pf >= 0 && pf < 8) {
                int pc = p->sq[pr * 8 + pf];
                if (pc && COLOR_OF(pc) == by && PT(pc) == 1) return 1;
            }
        }
    }
    for (i = 0; i < 8; ++i) {
        int rr = r + KN_D[i][0], ff = f + KN_D[i][1];
        if (rr >= 0 && rr < 8 && ff >= 0 && ff < 8) {
            int pc = p->sq[rr * 8 + ff];
            if (pc && COLOR_OF(pc) == by && PT(pc) == 2) return 1;
        }
        rr = r + KG_D[i][0]; ff = f + KG_D[i][1];
        if (rr >= 0 && rr < 8 && ff >= 0 && ff < 8) {
            int pc = p->sq[rr * 8 + ff];
            if (pc && COLOR_OF(pc) == by && PT(pc) == 6) return 1;
        }
    }
    for (i = 0; i < 4; ++i) {
        int rr = r + BI_D[i][0], ff = f + BI_D[i][1];
        while (rr >= 0 && rr < 8 && ff >= 0 && ff < 8) {
            int pc = p->sq[rr * 8 + ff];
            if (pc) { if (COLOR_OF(pc) == by && (PT(pc) == 3 || PT(pc) == 5)) return 1; break; }
            rr += BI_D[i][0]; ff += BI_D[i][1];
        }
        rr = r + RO_D[i][0]; ff = f + RO_D[i][1];
        while (rr >= 0 && rr < 8 && ff >= 0 && ff < 8) {
            int pc = p->sq[rr * 8 + ff];
            if (pc) { if (COLOR_OF(pc) == by && (PT(pc) == 4 || PT(pc) == 5)) return 1; break; }
            rr += RO_D[i][0]; ff += RO_D[i][1];
        }
    }
    return 0;
}

static int king_sq(const OPos* p, int color) {
    int s, want = color ? 6 : 12, found = -1;
    for (s = 0; s < 64; ++s) if (p->sq[s] == want) found = s; /* msb, as python-chess */
    return found;
}

int o_in_check(const OPos* p) {
    int k = king_sq(p, p->turn);
    return k >= 0 && attacked(p, k, !p->turn);
}

/* clean_castling_rights(): rights whose rook stands on its corner and whose king stands on e1/e8 */
uint64_t o_clean_castling(const OPos* p) {
    uint64_t out = 0;
    if (p->sq[4] == 6) {
        if ((p->castling & (1ull << 0)) && p->sq[0] == 4) out |= 1ull << 0;
        if ((p->castling & (1ull << 7)) && p->sq[7] == 4) out |= 1ull << 7;
    }
    if (p->sq[60] == 12) {
        if ((p->castling & (1ull << 56)) && p->sq[56] == 10) out |= 1ull << 56;
        if ((p->castling & (1ull << 63)) && p->sq[63] == 10) out |= 1ull << 63;
    }
    return out;
}

int o_has_kingside(const OPos* p, int color) { return (o_clean_castling(p) >> (color ? 7 : 63)) & 1; }
int o_has_queenside(const OPos* p, int color) { return (o_clean_castling(p) >> (color ? 0 : 56)) & 1; }

void o_make(OPos* p, OMove m) {
    int pc = p->sq[m.from];
    int type = PT(pc);
    int captured = p->sq[m.to];
    int zeroing = (type == 1) || captured != EMPTY;
    int ep_old = p->ep;
    p->ep = -1;
    /* castling rights: from/to squares lose rights; king move loses the whole back rank */
    p->castling &= ~(1ull << m.from) & ~(1ull << m.to);
    if (type == 6) p->castling &= p->turn ? ~0xFFull : ~(0xFFull << 56);
    p->sq[m.from] = EMPTY;
    if (type == 1) {
        int diff = (int)m.to - (int)m.from;
        if (diff == 16 && rank_of(m.from) == 1) p->ep = m.from + 8;
        else if (diff == -16 && rank_of(m.from) == 6) p->ep = m.from - 8;
        else if (m.to == ep_old && (diff == 7 || diff == 9 || diff == -7 || diff == -9) && captured == EMPTY) {
            int cap_sq = m.to + (p->turn ? -8 : 8);
            p->sq[cap_sq] = EMPTY;
        }
    }
    if (type == 6 && abs(file_of(m.to) - file_of(m.from)) == 2) { /* castling as king e->g / e->c */
        int r = rank_of(m.from) * 8;
        if (file_of(m.to) == 6) { p->sq[r + 5] = p->sq[r + 7]; p->sq[r + 7] = EMPTY; }
        else { p->sq[r + 3] = p->sq[r + 0]; p->sq[r + 0] = EMPTY; }
    }
    if (m.promo) p->sq[m.to] = (int8_t)((p->turn ? 0 : 6) + m.promo);
    else p->sq[m.to] = (int8_t)pc;
    if (zeroing) p->halfmove = 0; else p->halfmove += 1;
    if (!p->turn) p->fullmove += 1;
    p->turn = !p->turn;
}

static int legal_after(const OPos* p, OMove m) {
    OPos q = *p;
    int us = p->turn;
    o_make(&q, m);
    int k = king_sq(&q, us);
    return k < 0 || !attacked(&q, k, !us);
}

static int add(const OPos* p, OMove* out, int n, int from, int to, int promo) {
    OMove m; m.from = (uint8_t)from; m.to = (uint8_t)to; m.promo = (uint8_t)promo;
    if (legal_after(p, m)) out[n++] = m;
    return n;
}

static int add_promos(const OPos* p, OMove* out, int n, int from, int to) {
    if (rank_of(to) == 0 || rank_of(to) == 7) {
        n = add(p, out, n, from, to, 5); n = add(p, out, n, from, to, 4);
        n = add(p, out, n, from, to, 3); n = add(p, out, n, from, to, 2);
        return n;
    }
    return add(p, out, n, from, to, 0);
}

/* piece (non-pawn) destinations in descending square order */
static int piece_moves(const OPos* p, OMove* out, int n, int from) {
    int pc = p->sq[from], type = PT(pc), us = p->turn, to;
    uint64_t targets = 0;
    int r = rank_of(from), f = file_of(from), i;
    if (type == 2 || type == 6) {
        const int(*D)[2] = type == 2 ? KN_D : KG_D;
        for (i = 0; i < 8; ++i) {
            int rr = r + D[i][0], ff = f + D[i][1];
            if (rr >= 0 && rr < 8 && ff >= 0 && ff < 8) {
                int t = p->sq[rr * 8 + ff];
                if (!t || COLOR_OF(t) != us) targets |= 1ull << (rr * 8 + ff);
            }
        }
    } else {
        if (type == 3 || type == 5)
            for (i = 0; i < 4; ++i) {
                int rr = r + BI_D[i][0], ff = f + BI_D[i][1];
                while (rr >= 0 && rr < 8 && ff >= 0 && ff < 8) {
                    int t = p->sq[rr * 8 + ff];
                    if (!t || COLOR_OF(t) != us) targets |= 1ull << (rr * 8 + ff);
                    if (t) break;
                    rr += BI_D[i][0]; ff += BI_D[i][1];
                }
            }
        if (type == 4 || type == 5)
            for (i = 0; i < 4; ++i) {
                int rr = r + RO_D[i][0], ff = f + RO_D[i][1];
                while (rr >= 0 && rr < 8 && ff >= 0 && ff < 8) {
                    int t = p->sq[rr * 8 + ff];
                    if (!t || COLOR_OF(t) != us) targets |= 1ull << (rr * 8 + ff);
                    if (t) break;
                    rr += RO_D[i][0]; ff += RO_D[i][1];
                }
            }
    }
    for (to = 63; to >= 0; --to)
        if (targets & (1ull << to)) n = add(p, out, n, from, to, 0);
    return n;
}

int o_gen_legal(const OPos* p, OMove* out) {
    int n = 0, us = p->turn, s, in_check = o_in_check(p);
    int ksq = king_sq(p, us);
    if (in_check && ksq >= 0) n = piece_moves(p, out, n, ksq);
    for (s = 63; s >= 0; --s) {
        int pc = p->sq[s];
        if (!pc || COLOR_OF(pc) != us || PT(pc) == 1) continue;
        if (in_check && s == ksq) continue;
        n = piece_moves(p, out, n, s);
    }
    /* castling: h-side then a-side; king not in check, path empty, king path not attacked */
    if (!in_check && ksq >= 0) {
        uint64_t cr = o_clean_castling(p);
        int base = us ? 0 : 56;
        if (ksq == base + 4) {
            if ((cr >> (base + 7)) & 1) {
                if (!p->sq[base + 5] && !p->sq[base + 6] && !attacked(p, base + 5, !us) && !attacked(p, base + 6, !us)) {
                    OMove m = {(uint8_t)ksq, (uint8_t)(base + 6), 0};
                    out[n++] = m;
                }
            }
            if ((cr >> base) & 1) {
                if (!p->sq[base + 1] && !p->sq[base + 2] && !p->sq[base + 3] && !attacked(p, base + 3, !us) &&
                    !attacked(p, base + 2, !us)) {
                    OMove m = {(uint8_t)ksq, (uint8_t)(base + 2), 0};
                    out[n++] = m;
                }
            }
        }
    }
    /* pawn captures */
    for (s = 63; s >= 0; --s) {
        int pc = p->sq[s], k;
        if (!pc || COLOR_OF(pc) != us || PT(pc) != 1) continue;
        int r = rank_of(s) + (us ? 1 : -1);
        if (r < 0 || r > 7) continue;
        for (k = 1; k >= -1; k -= 2) { /* higher to-square first */
            int f = file_of(s) + k;
            if (f < 0 || f > 7) continue;
            int to = r * 8 + f, t = p->sq[to];
            if (t && COLOR_OF(t) != us) n = add_promos(p, out, n, s, to);
        }
    }
    /* single pushes by to-square descending */
    for (s = 63; s >= 0; --s) {
        int from = s + (us ? -8 : 8);
        if (from < 0 || from > 63 || p->sq[s]) continue;
        int pc = p->sq[from];
        if (pc && COLOR_OF(pc) == us && PT(pc) == 1) n = add_promos(p, out, n, from, s);
    }
    /* double pushes by to-square descending */
    for (s = 63; s >= 0; --s) {
        if (rank_of(s) != (us ? 3 : 4) || p->sq[s]) continue;
        int mid = s + (us ? -8 : 8), from = s + (us ? -16 : 16);
        int pc = p->sq[from];
        if (!p->sq[mid] && pc && COLOR_OF(pc) == us && PT(pc) == 1) n = add(p, out, n, from, s, 0);
    }
    /* en passant by capturer descending */
    if (p->ep >= 0 && !p->sq[p->ep]) {
        int er = rank_of(p->ep), k;
        int cr = er + (us ? -1 : 1);
        if (cr == (us ? 4 : 3)) {
            for (k = 1; k >= -1; k -= 2) {
                int f = file_of(p->ep) + k;
                if (f < 0 || f > 7) continue;
                int from = cr * 8 + f, pc = p->sq[from];
                if (pc && COLOR_OF(pc) == us && PT(pc) == 1) n = add(p, out, n, from, p->ep, 0);
            }
        }
    }
    return n;
}

int o_parse_fen(const char* fen, OPos* p) {
    memset(p, 0, sizeof(*p));
    p->ep = -1; p->halfmove = 0; p->fullmove = 1; p->turn = 1;
    int r = 7, f = 0;
    const char* c = fen;
    while (*c && *c != ' ') {
        if (*c == '/') { r--; f = 0; }
        else if (*c >= '1' && *c <= '8') f += *c - '0';
        else {
            const char* syms = "PNBRQKpnbrqk";
            const char* q = strchr(syms, *c);
            if (!q || r < 0 || f > 7) return -1;
            p->sq[r * 8 + f] = (int8_t)(q - syms + 1);
            f++;
        }
        c++;
    }
    if (*c == ' ') c++;
    if (*c == 'w') p->turn = 1; else if (*c == 'b') p->turn = 0; else return -1;
    c++;
    if (*c == ' ') c++;
    while (*c && *c != ' ') {
        if (*c == 'K') p->castling |= 1ull << 7;
        else if (*c == 'Q') p->castling |= 1ull << 0;
        else if (*c == 'k') p->castling |= 1ull << 63;
        else if (*c == 'q') p->castling |= 1ull << 56;
        c++;
    }
    if (*c == ' ') c++;
    if (*c && *c != '-' && c[1]) { p->ep = (int8_t)((c[1] - '1') * 8 + (c[0] - 'a')); c += 2; }
    else if (*c) c++;
    if (*c == ' ') c++;
    if (*c) { p->halfmove = atoi(c); while (*c && *c != ' ') c++; if (*c == ' ') c++; }
    if (*c) p->fullmove = atoi(c);
    /* python-chess set_castling_fen keeps only rights that have a rook on the square */
    {
        uint64_t keep = 0;
        if ((p->castling >> 7 & 1) && p->sq[7] == 4) keep |= 1ull << 7;
        if ((p->castling >> 0 & 1) && p->sq[0] == 4) keep |= 1ull << 0;
        if ((p->castling >> 63 & 1) && p->sq[63] == 10) keep |= 1ull << 63;
        if ((p->castling >> 56 & 1) && p->sq[56] == 10) keep |= 1ull << 56;
        p->castling = keep;
    }
    return 0;
}

/* has_legal_en_passant(): an ep capture that is fully legal exists */
int o_has_legal_ep(const OPos* p) {
    if (p->ep < 0 || p->sq[p->ep]) return 0;
    int us = p->turn, er = rank_of(p->ep), cr = er + (us ? -1 : 1), k;
    if (cr != (us ? 4 : 3)) return 0;
    for (k = -1; k <= 1; k += 2) {
        int f = file_of(p->ep) + k;
        if (f < 0 || f > 7) continue;
        int from = cr * 8 + f, pc = p->sq[from];
        if (pc && COLOR_OF(pc) == us && PT(pc) == 1) {
            OMove m = {(uint8_t)from, (uint8_t)p->ep, 0};
            if (legal_after(p, m)) return 1;
        }
    }
    return 0;
}

/* _transposition_key(): piece placement, turn, cleaned castling rights, legal ep square.
 * Serialised into 68 bytes so that equality == python tuple equality. */
void o_tkey(const OPos* p, uint8_t* out68) {
    memcpy(out68, p->sq, 64);
    out68[64] = (uint8_t)p->turn;
    uint64_t cr = o_clean_castling(p);
    out68[65] = (uint8_t)(((cr >> 7) & 1) | (((cr >> 0) & 1) << 1) | (((cr >> 63) & 1) << 2) | (((cr >> 56) & 1) << 3));
    out68[66] = (uint8_t)(o_has_legal_ep(p) ? p->ep : 255);
    out68[67] = 0;
}

int o_insufficient_side(const OPos* p, int color) {
    int s, own_n = 0, own_b = 0, own_cnt = 0, opp_other = 0;
    int pawns = 0, knights = 0, light_b = 0, dark_b = 0;
    for (s = 0; s < 64; ++s) {
        int pc = p->sq[s];
        if (!pc) continue;
        int t = PT(pc), c = COLOR_OF(pc);
        if (t == 1) pawns++;
        if (t == 2) knights++;
        if (t == 3) { if ((rank_of(s) + file_of(s)) & 1) light_b++; else dark_b++; }
        if (c == color) {
            own_cnt++;
            if (t == 1 || t == 4 || t == 5) return 0;
            if (t == 2) own_n++;
            if (t == 3) own_b++;
        } else {
            if (t != 6 && t != 5) opp_other++;
        }
    }
    if (own_n) return own_cnt <= 2 && !opp_other;
    if (own_b) {
        int same_color = (dark_b == 0) || (light_b == 0);
        return same_color && !pawns && !knights;
    }
    return 1;
}

int o_is_insufficient(const OPos* p) { return o_insufficient_side(p, 1) && o_insufficient_side(p, 0); }

int o_any_legal(const OPos* p) {
    OMove mv[MAX_MOVES];
    return o_gen_legal(p, mv) > 0;
}

int o_is_checkmate(const OPos* p) { return o_in_check(p) && !o_any_legal(p); }
int o_is_stalemate(const OPos* p) { return !o_in_check(p) && !o_any_legal(p); }

/* ---- game with history (move stack) ---- */
typedef struct {
    OPos cur;
    int n;              /* plies in stack */
    int cap;
    OPos* stack;        /* positions before each move */
    OMove* moves;
} OGame;

OGame* o_game_new(const char* fen) {
    OGame* g = (OGame*)calloc(1, sizeof(OGame));
    if (o_parse_fen(fen, &g->cur) != 0) { free(g); return NULL; }
    g->cap = 64;
    g->stack = (OPos*)malloc(sizeof(OPos) * g->cap);
    g->moves = (OMove*)malloc(sizeof(OMove) * g->cap);
    return g;
}
void o_game_free(OGame* g) { if (g) { free(g->stack); free(g->moves); free(g); } }
OGame* o_game_copy(const OGame* s) {
    OGame* g = (OGame*)malloc(sizeof(OGame));
    *g = *s;
    g->stack = (OPos*)malloc(sizeof(OPos) * g->cap);
    g->moves = (OMove*)malloc(sizeof(OMove) * g->cap);
    memcpy(g->stack, s->stack, sizeof(OPos) * s->n);
    memcpy(g->moves, s->moves, sizeof(OMove) * s->n);
    return g;
}
void o_game_push(OGame* g, OMove m) {
    if (g->n == g->cap) {
        g->cap *= 2;
        g->stack = (OPos*)realloc(g->stack, sizeof(OPos) * g->cap);
        g->moves = (OMove*)realloc(g->moves, sizeof(OMove) * g->cap);
    }
    g->stack[g->n] = g->cur; g->moves[g->n] = m; g->n++;
    o_make(&g->cur, m);
}
void o_game_pop(OGame* g) { if (g->n > 0) { g->n--; g->cur = g->stack[g->n]; } }
const OPos* o_game_pos(const OGame* g) { return &g->cur; }
int o_game_len(const OGame* g) { return g->n; }

/* is_irreversible(move) evaluated on the position BEFORE the move */
static int irreversible(const OPos* before, OMove m) {
    int pc = before->sq[m.from];
    int zeroing = PT(pc) == 1 || before->sq[m.to] != EMPTY;
    if (zeroing) return 1;
    OPos after = *before;
    uint64_t cr0 = o_clean_castling(before);
    o_make(&after, m);
    /* python-chess: _reduces_castling_rights(move) or has_legal_en_passant() */
    OPos tmp = *before;
    tmp.castling &= ~(1ull << m.from) & ~(1ull << m.to);
    if (PT(pc) == 6) tmp.castling &= before->turn ? ~0xFFull : ~(0xFFull << 56);
    uint64_t cr1 = 0;
    {   /* clean rights of `before` restricted to what survives the move */
        cr1 = cr0 & tmp.castling;
        if (PT(pc) == 6) cr1 &= before->turn ? ~0xFFull : ~(0xFFull << 56);
    }
    if (cr1 != cr0) return 1;
    if (o_has_legal_ep(before)) return 1;
    (void)after;
    return 0;
}

int o_game_is_repetition(const OGame* g, int count) {
    uint8_t key[68], k2[68];
    o_tkey(&g->cur, key);
    int i = g->n;
    int c = count;
    while (1) {
        if (c <= 1) return 1;
        if (i < c - 1) break;      /* len(move_stack) < count - 1 */
        i--;
        if (i < 0) break;
        if (irreversible(&g->stack[i], g->moves[i])) break;
        o_tkey(&g->stack[i], k2);
        if (memcmp(key, k2, 68) == 0) c--;
    }
    return c <= 1;
}

int o_game_can_claim_threefold(OGame* g) {
    uint8_t key[68];
    o_tkey(&g->cur, key);
    /* collect keys back to the last irreversible move */
    int cap = g->n + 1, nk = 0, i;
    uint8_t* keys = (uint8_t*)malloc((size_t)cap * 68);
    memcpy(keys, key, 68); nk = 1;
    for (i = g->n - 1; i >= 0; --i) {
        if (irreversible(&g->stack[i], g->moves[i])) break;
        o_tkey(&g->stack[i], keys + (size_t)nk * 68); nk++;
    }
    int cnt = 0, res = 0;
    for (i = 0; i < nk; ++i) if (memcmp(keys + (size_t)i * 68, key, 68) == 0) cnt++;
    if (cnt >= 3) res = 1;
    if (!res) {
        OMove mv[MAX_MOVES];
        int n = o_gen_legal(&g->cur, mv), j;
        for (j = 0; j < n && !res; ++j) {
            OPos q = g->cur;
            uint8_t k2[68];
            o_make(&q, mv[j]);
            o_tkey(&q, k2);
            int c2 = 0;
            for (i = 0; i < nk; ++i) if (memcmp(keys + (size_t)i * 68, k2, 68) == 0) c2++;
            if (c2 >= 2) res = 1;
        }
    }
    free(keys);
    return res;
}

int o_is_fifty(const OPos* p) { return p->halfmove >= 100 && o_any_legal(p); }
int o_is_seventyfive(const OPos* p) { return p->halfmove >= 150 && o_any_legal(p); }

int o_can_claim_fifty(const OPos* p) {
    if (o_is_fifty(p)) return 1;
    if (p->halfmove >= 99) {
        OMove mv[MAX_MOVES];
        int n = o_gen_legal(p, mv), j;
        for (j = 0; j < n; ++j) {
            int pc = p->sq[mv[j].from];
            int zeroing = PT(pc) == 1 || p->sq[mv[j].to] != EMPTY;
            if (!zeroing) {
                OPos q = *p;
                o_make(&q, mv[j]);
                if (o_is_fifty(&q)) return 1;
            }
        }
    }
    return 0;
}

/* outcome(): 0 none, 1 checkmate, 2 insufficient, 3 stalemate, 4 fifty-claim, 5 threefold-claim,
 * 6 seventyfive, 7 fivefold */
int o_game_outcome(OGame* g, int claim_draw) {
    const OPos* p = &g->cur;
    if (o_is_checkmate(p)) return 1;
    if (o_is_insufficient(p)) return 2;
    if (!o_any_legal(p)) return 3;
    if (claim_draw) {
        if (o_can_claim_fifty(p)) return 4;
        if (o_game_can_claim_threefold(g)) return 5;
    }
    if (o_is_seventyfive(p)) return 6;
    if (o_game_is_repetition(g, 5)) return 7;
    return 0;
}

/* perft for pinning against published known answers */
uint64_t o_perft(const OPos* p, int depth) {
    OMove mv[MAX_MOVES];
    int n = o_gen_legal(p, mv), i;
    if (depth <= 1) return depth == 1 ? (uint64_t)n : 1;
    uint64_t t = 0;
    for (i = 0; i < n; ++i) { OPos q = *p; o_make(&q, mv[i]); t += o_perft(&q, depth - 1); }
    return t;
}

/* ---- azchess/encoding.py restated ---- */
/* encode_board (encoding.py:11-46): f32 [19][8][8], plane[7-rank][file] */
void o_encode_board(const OPos* p, float* out) {
    int s, i;
    memset(out, 0, sizeof(float) * 19 * 64);
    for (s = 0; s < 64; ++s) {
        int pc = p->sq[s];
        if (!pc) continue;
        int plane = IS_WHITE(pc) ? (PT(pc) - 1) : (6 + PT(pc) - 1);
        out[plane * 64 + (7 - rank_of(s)) * 8 + file_of(s)] = 1.0f;
    }
    float c[7];
    c[0] = p->turn ? 1.0f : 0.0f;
    c[1] = o_has_kingside(p, 1) ? 1.0f : 0.0f;
    c[2] = o_has_queenside(p, 1) ? 1.0f : 0.0f;
    c[3] = o_has_kingside(p, 0) ? 1.0f : 0.0f;
    c[4] = o_has_queenside(p, 0) ? 1.0f : 0.0f;
    c[5] = (float)((double)(p->halfmove < 99 ? p->halfmove : 99) / 99.0);
    c[6] = (float)((double)(p->fullmove < 199 ? p->fullmove : 199) / 199.0);
    for (i = 0; i < 7; ++i)
        for (s = 0; s < 64; ++s) out[(12 + i) * 64 + s] = c[i];
}

/* move_to_index (encoding.py:80-150) for a move already known legal; -1 if unmappable */
int o_move_to_index(const OPos* p, OMove m) {
    static const int RAY[8][2] = {{1, 0}, {-1, 0}, {0, 1}, {0, -1}, {1, 1}, {1, -1}, {-1, 1}, {-1, -1}};
    int fr = rank_of(m.from), ff = file_of(m.from), tr = rank_of(m.to), tf = file_of(m.to);
    int dr = tr - fr, df = tf - ff, i;
    for (i = 0; i < 8; ++i)
        if (KN_D[i][0] == dr && KN_D[i][1] == df) return m.from * 73 + 56 + i;
    if (m.promo == 2 || m.promo == 3 || m.promo == 4) {
        int dirs_w[3][2] = {{1, 0}, {1, -1}, {1, 1}}, dirs_b[3][2] = {{-1, 0}, {-1, 1}, {-1, -1}};
        for (i = 0; i < 3; ++i) {
            int a = p->turn ? dirs_w[i][0] : dirs_b[i][0], b = p->turn ? dirs_w[i][1] : dirs_b[i][1];
            if (a == dr && b == df) return m.from * 73 + 64 + (m.promo - 2) * 3 + i;
        }
    }
    if (dr == 0 || df == 0 || abs(dr) == abs(df)) {
        int step = abs(dr) > abs(df) ? abs(dr) : abs(df);
        int sdr = dr == 0 ? 0 : (dr > 0 ? 1 : -1), sdf = df == 0 ? 0 : (df > 0 ? 1 : -1);
        if (step >= 1 && step <= 7)
            for (i = 0; i < 8; ++i)
                if (RAY[i][0] == sdr && RAY[i][1] == sdf) return m.from * 73 + i * 7 + (step - 1);
    }
    return -1;
}

/* get_legal_actions (encoding.py:231-243): returns number of legal moves */
int o_legal_mask(const OPos* p, uint8_t* mask4672) {
    OMove mv[MAX_MOVES];
    int n = o_gen_legal(p, mv), i;
    memset(mask4672, 0, 4672);
    for (i = 0; i < n; ++i) {
        int idx = o_move_to_index(p, mv[i]);
        if (idx >= 0) mask4672[idx] = 1;
    }
    return n;
}

/* legal moves with indices, in generation order */
int o_legal_moves_idx(const OPos* p, OMove* mv, int32_t* idx) {
    int n = o_gen_legal(p, mv), i;
    for (i = 0; i < n; ++i) idx[i] = o_move_to_index(p, mv[i]);
    return n;
}

size_t o_sizeof_pos(void) { return sizeof(OPos); }
