// Tree-search kernels: one wavefront per game tree.
//   select_kernel   L descents per game and step: children scored one per lane from coalesced SoA
//                   loads (PUCT + FPU + virtual loss + jitter in fp64, mcts.py:851-925), wave
//                   arg-max with first-max tie-break, leaf board by make_move, terminal test,
//                   immediate terminal backup, 19-plane encode written straight into the network
//                   input (lane = square).
//   expand_kernel   after the network: legal moves, policy indices, legal-only softmax, entropy
//                   noise, renormalise (mcts.py:135-225), child block allocation, backup
//                   (mcts.py:946-953), virtual-loss release, root result extraction.
//   advance_kernel  re-root on the played move and compact the kept subtree into the other arena.
//   encode_positions_kernel  position-wise encoding.py functions (parity tests / boundary helpers).
#include <hip/hip_runtime.h>
#include <math.h>
#include "tree.h"
#include "kernel_common.h"   // DeviceOnce

using namespace m0;

#define GOLDEN64 0x9E3779B97F4A7C15ull

__device__ __forceinline__ double u01(uint64_t seed, uint64_t k) {
    return (double)(mix64(seed + (k + 1) * GOLDEN64) >> 11) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ double normal_at(uint64_t seed, uint64_t k) {   // consumes uniforms k, k+1
    double u1 = u01(seed, k), u2 = u01(seed, k + 1);
    if (u1 < 1e-300) u1 = 1e-300;
    return sqrt(-2.0 * log(u1)) * cos(2.0 * 3.141592653589793 * u2);
}
__device__ double gamma_draw(uint64_t seed, uint64_t& ctr, double a) {     // Marsaglia-Tsang
    double boost = 1.0;
    if (a < 1.0) {
        double u = u01(seed, ctr++);
        boost = pow(u, 1.0 / a);
        a += 1.0;
    }
    const double d = a - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (int it = 0; it < 1000; ++it) {
        double x = normal_at(seed, ctr); ctr += 2;
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        double u = u01(seed, ctr++);
        if (u < 1e-300) u = 1e-300;
        if (log(u) < 0.5 * x * x + d - d * v + d * log(v)) return d * v * boost;
    }
    return d * boost;
}

__device__ __forceinline__ double cpuct_at(const TreeCfg& c, int ply) {    // mcts.py:927-944
    if (c.use_c_base) {
        double N = fmax(1.0, (double)(ply + 1));
        return c.cpuct_c_init + log((N + c.cpuct_c_base) / c.cpuct_c_base);
    }
    if (c.cpuct_plies <= 0) return c.cpuct;
    int p = ply < 0 ? 0 : (ply > c.cpuct_plies ? c.cpuct_plies : ply);
    double t = (double)p / (double)c.cpuct_plies;
    return c.cpuct_start + (c.cpuct_end - c.cpuct_start) * t;
}

struct Arena {
    double* prior; double* w; double* q; int* n; int* vl; int* cbase; int16_t* nch; uint16_t* mv; uint16_t* midx;
};
__device__ __forceinline__ Arena arena_of(const TreeArrays& t, int g, int half) {
    size_t b = ((size_t)g * 2 + half) * (size_t)t.cap;
    Arena a;
    a.prior = t.prior + b; a.w = t.w + b; a.q = t.q + b; a.n = t.n + b; a.vl = t.vl + b;
    a.cbase = t.cbase + b; a.nch = t.nch + b; a.mv = t.mv + b; a.midx = t.midx + b;
    return a;
}

// Exclusive prefix sum over the wave's lanes (lane 0 first) of three packed 10-bit counters; `total` = wave sum.
__device__ __forceinline__ uint32_t wave_excl_scan3(uint32_t v, int lane, uint32_t& total) {
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(incl, off);
        if (lane >= off) incl += u;
    }
    total = __shfl(incl, 63);
    return incl - v;
}

// Pseudo-legal moves in python-chess generation order (chess_core.h gen_moves<false>), one SQUARE per lane:
// lane L works square 63 - L, so ascending lanes are python-chess's scan_reversed order.  Per category (piece moves by
// from-square; pawn captures by from-square; single pushes, double pushes by to-square; en passant by from-square)
// every lane counts its moves, three packed prefix sums give its output offsets, and it writes its own moves; the
// check-evasion king moves (first) and castling (after the pieces) are uniform work.  gen_moves<false> on every lane was
// 33 k cycles per leaf (the pseudo-legal list built 64 times over, sequentially).
__device__ int gen_pseudo_wave(const Pos& p, Move* out, int lane) {
    const int us = p.turn, them = us ^ 1;
    const uint64_t own = occ_of(p, us), theirs = occ_of(p, them), o = own | theirs;
    const int ksq = king_sq(p, us);
    const bool chk = ksq >= 0 && attacked(p, ksq, them);
    const int s = 63 - lane;
    const uint64_t sb = bit(s);
    const uint64_t pawns = p.bb[PAWN] & own;
    // --- check evasion: the king's moves come first and the king leaves the piece scan
    int n0 = 0;
    if (chk) {
        uint64_t t = king_att(ksq) & ~own;
        n0 = popc(t);
        if (lane == 0) { int j = 0; while (t) { const int to = msb(t); t &= ~bit(to); out[j++] = mk_move(ksq, to, 0); } }
    }
    // --- category A: non-pawn pieces
    uint64_t tA = 0;
    if ((own & ~p.bb[PAWN] & sb) && !(chk && s == ksq)) tA = piece_targets(p, s, piece_type_at(p, s));
    const uint32_t cA = (uint32_t)popc(tA);
    // --- castling (uniform), after the pieces
    Move cz[2];
    int ncz = 0;
    if (!chk && ksq >= 0) {
        const int cr = clean_cr(p);
        const int base = us == WHITE ? 0 : 56;
        if (ksq == base + 4) {
            const int kbit = us == WHITE ? CR_WK : CR_BK, qbit = us == WHITE ? CR_WQ : CR_BQ;
            if ((cr & kbit) && !(o & (bit(base + 5) | bit(base + 6))) && !attacked(p, base + 5, them) &&
                !attacked(p, base + 6, them))
                cz[ncz++] = mk_move(ksq, base + 6, 0);
            if ((cr & qbit) && !(o & (bit(base + 1) | bit(base + 2) | bit(base + 3))) && !attacked(p, base + 3, them) &&
                !attacked(p, base + 2, them))
                cz[ncz++] = mk_move(ksq, base + 2, 0);
        }
    }
    // --- category C: pawn captures (from-square s), promotions expand to 4
    const uint64_t tC = (pawns & sb) ? (pawn_att(s, us) & theirs) : 0;
    const uint64_t promo_rank = RANK_1 | RANK_8;
    const uint32_t cC = (uint32_t)(popc(tC & ~promo_rank) + 4 * popc(tC & promo_rank));
    // --- category D / E: single and double pushes (to-square s)
    const uint64_t single = (us == WHITE ? pawns << 8 : pawns >> 8) & ~o;
    const uint64_t dbl = (us == WHITE ? single << 8 : single >> 8) & ~o & (us == WHITE ? (RANK_1 << 24) : (RANK_1 << 32));
    const uint32_t cD = (single & sb) ? ((sb & promo_rank) ? 4u : 1u) : 0u;
    const uint32_t cE = (dbl & sb) ? 1u : 0u;
    // --- category F: en passant (from-square s)
    uint64_t epc = 0;
    if (p.ep >= 0 && !(o & bit(p.ep))) epc = pawns & pawn_att(p.ep, them) & (us == WHITE ? (RANK_1 << 32) : (RANK_1 << 24));
    const uint32_t cF = (epc & sb) ? 1u : 0u;
    // --- offsets
    uint32_t tot1, tot2;
    const uint32_t ex1 = wave_excl_scan3(cA | (cC << 10) | (cD << 20), lane, tot1);
    const uint32_t ex2 = wave_excl_scan3(cE | (cF << 10), lane, tot2);
    const int nA = tot1 & 1023, nC = (tot1 >> 10) & 1023, nD = (tot1 >> 20) & 1023, nE = tot2 & 1023, nF = (tot2 >> 10) & 1023;
    const int baseA = n0, baseZ = baseA + nA, baseC = baseZ + ncz, baseD = baseC + nC, baseE = baseD + nD, baseF = baseE + nE;
    {
        int j = baseA + (int)(ex1 & 1023);
        uint64_t t = tA;
        while (t) { const int to = msb(t); t &= ~bit(to); out[j++] = mk_move(s, to, 0); }
    }
    if (lane == 0) for (int i = 0; i < ncz; ++i) out[baseZ + i] = cz[i];
    {
        int j = baseC + (int)((ex1 >> 10) & 1023);
        uint64_t t = tC;
        while (t) {
            const int to = msb(t); t &= ~bit(to);
            if (bit(to) & promo_rank) { out[j++] = mk_move(s, to, 4); out[j++] = mk_move(s, to, 3); out[j++] = mk_move(s, to, 2); out[j++] = mk_move(s, to, 1); }
            else out[j++] = mk_move(s, to, 0);
        }
    }
    if (cD) {
        int j = baseD + (int)((ex1 >> 20) & 1023);
        const int from = s + (us == WHITE ? -8 : 8);
        if (cD == 4) { out[j++] = mk_move(from, s, 4); out[j++] = mk_move(from, s, 3); out[j++] = mk_move(from, s, 2); out[j++] = mk_move(from, s, 1); }
        else out[j] = mk_move(from, s, 0);
    }
    if (cE) out[baseE + (int)(ex2 & 1023)] = mk_move(s + (us == WHITE ? -16 : 16), s, 0);
    if (cF) out[baseF + (int)((ex2 >> 10) & 1023)] = mk_move(s, p.ep, 0);
    return baseF + nF;
}

// Legal moves in generation order: the pseudo-legal list (one square per lane, above), then the legality test
// (make + king-attack) spread one move per lane and an order-preserving ballot compaction.  Same list as gen_legal()
// (tests/test_chess_core_host.py pins gen_legal; tests/test_encoding_gpu.py compares this one on 10 000 positions).
__device__ int gen_legal_wave(const Pos& p, Move* out_lds, Move* tmp_lds, int lane) {
    const int np = gen_pseudo_wave(p, tmp_lds, lane);
    __syncthreads();
    int base = 0;
    for (int k0 = 0; k0 < np; k0 += 64) {
        const int i = k0 + lane;
        const Move m = i < np ? tmp_lds[i] : (Move)0;
        const bool ok = i < np && legal_after(p, m);
        const unsigned long long mask = __ballot(ok);
        if (ok) out_lds[base + __popcll(mask & ((1ull << lane) - 1ull))] = m;
        base += __popcll(mask);
    }
    __syncthreads();
    return base;
}

// lane = tensor square n (row-major, row 0 = rank 8): 32 fp16 channels (19 used)
__device__ void encode_nhwc(const Pos& p, _Float16* dst /*[64][32]*/, int lane) {
    const int s = (7 - (lane >> 3)) * 8 + (lane & 7);
    float c7[7];
    plane_consts(p, c7);
    const int pl = piece_plane(p, s);
    __attribute__((aligned(16))) _Float16 h[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) h[i] = (_Float16)0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) h[i] = (_Float16)(pl == i ? 1.f : 0.f);
#pragma unroll
    for (int i = 0; i < 7; ++i) h[12 + i] = (_Float16)c7[i];
    uint4* o = reinterpret_cast<uint4*>(dst + lane * 32);
    const uint4* hv = reinterpret_cast<const uint4*>(h);
    o[0] = hv[0]; o[1] = hv[1]; o[2] = hv[2]; o[3] = hv[3];
}

// ---- position table of the tt_merge mode (MCTS._tt_get / _tt_put / _register_children_in_tt, mcts.py:1231-1346):
// open addressing with linear probing over a per-game region; key 0 = empty; an entry holds the node registered LAST
// under its key (the reference's dict assignment overwrites).  Cleared when a search starts from a fresh root.
__device__ __forceinline__ uint64_t tt_key_of(const Pos& p) { const uint64_t k = tkey(p); return k ? k : 1ull; }
__device__ __forceinline__ int tt_home(uint64_t key, int cap) { return (int)((key ^ (key >> 29)) & (uint64_t)(cap - 1)); }
// whole-wave lookup of a wave-uniform key: 64 slots per probe round; -1 = not registered
__device__ int tt_lookup(const uint64_t* keys, const int* nodes, int cap, uint64_t key, int lane) {
    const int mask = cap - 1, home = tt_home(key, cap);
    for (int it = 0; it < cap; it += 64) {
        const uint64_t k = keys[(home + it + lane) & mask];
        const unsigned long long hit = __ballot(k == key), emp = __ballot(k == 0ull);
        const int fh = hit ? __builtin_ctzll(hit) : 64, fe = emp ? __builtin_ctzll(emp) : 64;
        if (fh < fe) return nodes[(home + it + fh) & mask];
        if (fe < 64) return -1;
    }
    return -1;
}
// per-lane insert-or-overwrite (the active lanes of one call hold distinct keys)
__device__ void tt_insert(uint64_t* keys, int* nodes, int cap, uint64_t key, int node) {
    const int mask = cap - 1;
    int s = tt_home(key, cap);
    for (int it = 0; it < cap; ++it) {
        const unsigned long long prev = atomicCAS(reinterpret_cast<unsigned long long*>(keys + s), 0ull, (unsigned long long)key);
        if (prev == 0ull || prev == (unsigned long long)key) { nodes[s] = node; return; }
        s = (s + 1) & mask;
    }
}

// numpy float32 add.reduce over a[0..n): pairwise with an 8-way unrolled base case (blocks of <= 128), n <= 256
__device__ float np_sum_f32_dev(const float* a, int n) {
    auto block = [&](const float* b, int m) -> float {
        if (m < 8) { float r = 0.f; for (int i = 0; i < m; ++i) r += b[i]; return r; }
        float r[8];
        for (int j = 0; j < 8; ++j) r[j] = b[j];
        int i;
        for (i = 8; i < m - (m % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += b[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < m; ++i) res += b[i];
        return res;
    };
    if (n <= 128) return block(a, n);
    int n2 = n / 2;
    n2 -= n2 % 8;
    return block(a, n2) + block(a + n2, n - n2);
}

// MCTS._backpropagate (mcts.py:946-953): the leaf gets +v, its parent -v, ...  The nodes of a path are distinct, so
// every level is independent: lane d updates level d (all levels' loads in flight at once instead of a chain of
// dependent read-modify-writes by one lane); the arithmetic per node is that of the sequential loop (negation is exact).
// (Ln / Lq / ncached: select_kernel's LDS copies of n and q for the nodes below ncached get the very values that go to the arena)
__device__ void backprop(const Arena& A, const int* path, int depth, double value, int lane, bool may_repeat = false,
                         int* Ln = nullptr, double* Lq = nullptr, int ncached = 0) {   // whole-wave caller
    const double v = fmax(-1.0, fmin(1.0, value));
    if (may_repeat) {
        // tt_merge: a path can pass through the same node twice (a position repeated along the line); the reference's
        // sequential loop then updates it twice, in path order from the leaf up -- do exactly that when it happens
        bool dup = false;
        for (int d = lane; d <= depth; d += 64) {
            const int nd = path[d];
            for (int e = 0; e < d; ++e) dup = dup || path[e] == nd;
        }
        if (__any(dup)) {
            if (lane == 0) {
                double vv = v;
                for (int d = depth; d >= 0; --d) {
                    const int nd = path[d];
                    const int nn = A.n[nd] + 1;
                    const double ww = A.w[nd] + vv;
                    A.n[nd] = nn; A.w[nd] = ww; A.q[nd] = ww / (double)nn;
                    vv = -vv;
                }
            }
            return;
        }
    }
    for (int d = lane; d <= depth; d += 64) {
        const int nd = path[d];
        const int nn = A.n[nd] + 1;
        const double ww = A.w[nd] + (((depth - d) & 1) ? -v : v);
        const double qq = ww / (double)nn;
        A.n[nd] = nn; A.w[nd] = ww; A.q[nd] = qq;
        if (nd < ncached) { Ln[nd] = nn; Lq[nd] = qq; }
    }
}

__device__ void apply_dirichlet(const Arena& A, int root, GameDev* gd, const TreeCfg& c, double* sg, int lane) {
    const int k = A.nch[root];
    if (k <= 0 || c.dirichlet_frac <= 0.0) return;
    const int cb = A.cbase[root];
    uint64_t ctr = gd->ctr_dir;
    double sum = 0.0;
    for (int i = 0; i < k; ++i) {                 // sequential draws, uniform across lanes
        double gmm = gamma_draw(gd->seed_dir, ctr, c.dirichlet_alpha);
        if (lane == 0) sg[i] = gmm;
        sum += gmm;
    }
    __syncthreads();
    for (int i = lane; i < k; i += 64) {
        double nv = A.prior[cb + i] * (1.0 - c.dirichlet_frac) + (sg[i] / sum) * c.dirichlet_frac;
        A.prior[cb + i] = fmax(1e-8, fmin(1.0 - 1e-8, nv));
    }
    if (lane == 0) gd->ctr_dir = ctr;
    __syncthreads();
}

// What an evaluation-cache entry stands for besides its 64-bit key: the legal-move count (low 8 bits; cacheable positions have
// at most M0_EC_MAXLEGAL = 64 moves, one per lane) and a 24-bit checksum of the legal moves in order.  A hit whose signature
// differs is a key collision and is treated as a miss: the payload's logits are stored in legal-move order, so serving them
// to another move list would expand the node with wrong priors without any other symptom.
__device__ __forceinline__ int legal_sig(const Move* mv, int n, int lane) {
    uint32_t h = lane < n ? ((uint32_t)mv[lane] + 1u) * 0x9E3779B1u + (uint32_t)lane * 0x85EBCA6Bu : 0u;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 13;
    for (int o = 32; o > 0; o >>= 1) h ^= (uint32_t)__shfl_xor((int)h, o);
    return n | (int)(h & 0xFFFFFF00u);
}

// table of game g (of its side gd->arena in a match engine with per-side tables)
__device__ __forceinline__ size_t tt_table_of(const TreeDev& d, int g, const GameDev* gd) {
    return ((size_t)g * d.tt_sides + (d.tt_sides == 2 ? gd->arena : 0)) * (size_t)d.tt_cap;
}

// The top of a game's tree lives in LDS for the duration of a select launch (north star: "tree walk over LDS-resident node
// arrays"): after every played move the kept subtree is compacted BREADTH-FIRST (advance_kernel), so the nodes with the lowest
// indices are the root, its children, their children ... -- the levels every one of the pass's 96 descents walks through.  The
// first M0_SEL_CACHE nodes' select fields (prior, q, n, in-flight count, child block, child count, move: 32 B per node) are
// copied once per launch; the children scan reads them from LDS instead of paying a global round trip per level, and the two
// things a select launch writes to such nodes (in-flight counts, the statistics of a terminal leaf's path) are written through.
// Deeper nodes -- and everything in the table modes, whose arenas are not compacted -- are read from the arenas in HBM as before.
#ifndef M0_SEL_CACHE_N
#define M0_SEL_CACHE_N 2048
#endif
constexpr int M0_SEL_CACHE = M0_SEL_CACHE_N;
constexpr int M0_SEL_CACHE_BYTES = M0_SEL_CACHE * 32;

__global__ __launch_bounds__(64) void select_kernel(TreeDev d, TreeCfg c) {
    extern __shared__ __attribute__((aligned(16))) char sel_cache[];
    double* const L_prior = reinterpret_cast<double*>(sel_cache);
    double* const L_q = L_prior + M0_SEL_CACHE;
    int* const L_n = reinterpret_cast<int*>(L_q + M0_SEL_CACHE);
    int* const L_vl = L_n + M0_SEL_CACHE;
    int* const L_cb = L_vl + M0_SEL_CACHE;
    int16_t* const L_nch = reinterpret_cast<int16_t*>(L_cb + M0_SEL_CACHE);
    uint16_t* const L_mv = reinterpret_cast<uint16_t*>(L_nch + M0_SEL_CACHE);
    __shared__ uint64_t pkey[M0_MAX_DEPTH];
    __shared__ uint8_t pirr[M0_MAX_DEPTH];
    __shared__ Move smoves[M0_MAX_MOVES];
    __shared__ Move spseudo[M0_MAX_MOVES];
    __shared__ double sg[M0_MAX_CHILDREN];
    const int g = blockIdx.x, lane = threadIdx.x;
    GameDev* gd = &d.games[g];
    if (!gd->active) { if (lane == 0) gd->nsamples = 0; return; }
    uint16_t* LM = d.leaf_moves + (size_t)g * (d.L + 1) * M0_MAX_CHILDREN;
    const Arena A = arena_of(d.t, g, gd->arena);
    const int root = gd->root;
    Sample* S = d.samples + (size_t)g * (d.L + 1);
    int* P = d.paths + (size_t)g * (d.L + 1) * M0_MAX_DEPTH;
    int* EP = c.tt_merge ? d.epaths + (size_t)g * (d.L + 1) * M0_MAX_DEPTH : nullptr;
    const uint64_t* TK = c.tt_merge ? d.tt_keys + tt_table_of(d, g, gd) : nullptr;
    const int* TN = c.tt_merge ? d.tt_nodes + tt_table_of(d, g, gd) : nullptr;
    // mcts.py:359-371: a root taken over from the previous search is evaluated once more (value only)
    auto emit_reinfer = [&](int at) {
        int row = 0;
        if (lane == 0) row = atomicAdd(d.row_counter + gd->net_id, 1) + gd->net_id * d.net_row_base;
        row = __shfl(row, 0);
        if (lane == 0) {
            Sample s; s.pos = gd->root_pos; s.kind = 4; s.leaf = root; s.depth = 0; s.row = row; s.nlegal = 0; s.ckey = 0;
            S[at] = s; gd->reinfer = 0;
        }
        encode_nhwc(gd->root_pos, d.x0 + (size_t)row * 64 * 32, lane);
    };
    const bool reinfer = gd->reinfer != 0;

    if (A.nch[root] < 0) {                       // root not expanded: one network evaluation, no simulation
        if (lane == 0 && !gd->root_fresh) {
            // reused but never expanded child: run() applies Dirichlet BEFORE expanding it (a no-op on a
            // childless node, mcts.py:374-376 vs 398-413) and then sets root.q = v
            gd->need_dirichlet = 0;
            gd->root_q_from_v = 1;
        }
        int row = 0;
        if (lane == 0) row = atomicAdd(d.row_counter + gd->net_id, 1) + gd->net_id * d.net_row_base;
        row = __shfl(row, 0);
        const Pos rp = gd->root_pos;
        const int nl = gen_legal_wave(rp, smoves, spseudo, lane);
        for (int i = lane; i < nl; i += 64) LM[i] = smoves[i];
        if (lane == 0) {
            Sample s; s.pos = rp; s.kind = 2; s.leaf = root; s.depth = 0; s.row = row; s.nlegal = nl; s.ckey = 0;
            S[0] = s; P[0] = root; gd->nsamples = reinfer ? 2 : 1;
        }
        encode_nhwc(rp, d.x0 + (size_t)row * 64 * 32, lane);
        if (reinfer) emit_reinfer(1);
        return;
    }
    if (gd->need_dirichlet) {
        apply_dirichlet(A, root, gd, c, sg, lane);
        if (lane == 0) gd->need_dirichlet = 0;
    }
    int nleaf = gd->sims_target - gd->sims_done;
    if (nleaf > d.L) nleaf = d.L;
    if (nleaf < 0) nleaf = 0;
    // the top of the tree -> LDS (after the Dirichlet noise, which rewrites the root's priors)
    int ncached = 0;
    if (!c.tt_merge && nleaf > 0) {
        ncached = gd->next < M0_SEL_CACHE ? gd->next : M0_SEL_CACHE;
        for (int i = lane; i < ncached; i += 64) {
            L_prior[i] = A.prior[i]; L_q[i] = A.q[i]; L_n[i] = A.n[i]; L_vl[i] = A.vl[i]; L_cb[i] = A.cbase[i];
            L_nch[i] = A.nch[i]; L_mv[i] = A.mv[i];
        }
        __syncthreads();
    }
    uint64_t ctrj = gd->ctr_jitter;
    const uint64_t seedj = gd->seed_jitter;
    const double jit = c.selection_jitter > 0.0 ? c.selection_jitter : 0.001;
    const int hist_len = gd->hist_len;
    const uint64_t* H = d.hist + (size_t)g * M0_HIST_CAP;

    for (int s = 0; s < nleaf; ++s) {
        Pos pos = gd->root_pos;
        int* path = P + (size_t)s * M0_MAX_DEPTH;
        int* epath = c.tt_merge ? EP + (size_t)s * M0_MAX_DEPTH : nullptr;
        int node = root, depth = 0;
        int prev_from = -1, prev_to = -1;
        if (lane == 0) { path[0] = root; if (epath) epath[0] = root; }
        // One round of dependent loads per level: the children scan also fetches every candidate's own node fields
        // (children ARE nodes), and the winner's are broadcast -- the next level starts without loading its node.
        const bool rh = node < ncached;
        int nc = rh ? (int)L_nch[node] : (int)A.nch[node], cb = rh ? L_cb[node] : A.cbase[node], nn = rh ? L_n[node] : A.n[node];
        double nq = rh ? L_q[node] : A.q[node];
        while (true) {
            if (nc <= 0 || depth >= M0_MAX_DEPTH - 1) break;
            const double sq = sqrt((double)(nn > 1 ? nn : 1));
            const double eff = cpuct_at(c, depth);
            double best = -1e9;
            int bi = -1;
            int b_nch = 0, b_cb = 0, b_n = 0, b_vl = 0;
            double b_q = 0.0;
            Move b_mv = 0;
            for (int i = lane; i < nc; i += 64) {
                const int ci = cb + i;
                const bool hit = ci < ncached;                  // the children of a node are one block: all lanes agree but at the edge
                const int cn = hit ? L_n[ci] : A.n[ci];
                const double cq = hit ? L_q[ci] : A.q[ci];
                const int c_nch = hit ? (int)L_nch[ci] : (int)A.nch[ci], c_cb = hit ? L_cb[ci] : A.cbase[ci];
                const Move m = hit ? (Move)L_mv[ci] : A.mv[ci];
                const double cprior = hit ? L_prior[ci] : A.prior[ci];
                const double qq = cn == 0 ? nq - c.fpu_reduction : cq;
                const double u = eff * cprior * (sq / (1.0 + (double)cn));
                double sc = qq + u;
                if (c.no_instant_backtrack && depth >= 1) {
                    if (mv_from(m) == prev_to && mv_to(m) == prev_from) sc -= 0.01;
                }
                const int cvl = c.virtual_loss_active ? (hit ? L_vl[ci] : A.vl[ci]) : 0;
                if (c.virtual_loss_active && c.virtual_loss > 0.0) sc -= (double)cvl * c.virtual_loss;
                sc += (u01(seedj, ctrj + (uint64_t)i) - 0.5) * jit;
                if (sc > best) { best = sc; bi = i; b_nch = c_nch; b_cb = c_cb; b_n = cn; b_q = cq; b_mv = m; b_vl = cvl; }
            }
            for (int off = 32; off > 0; off >>= 1) {
                const double ob = __shfl_xor(best, off);
                const int oi = __shfl_xor(bi, off);
                if (oi >= 0 && (bi < 0 || ob > best || (ob == best && oi < bi))) { best = ob; bi = oi; }
            }
            ctrj += (uint64_t)nc;
            if (bi < 0) bi = 0;
            const int child = cb + bi;
            // the lane that scanned child bi (i = lane + 64 k) holds its fields iff its own best is bi
            const int wl = bi & 63;
            nc = __shfl(b_nch, wl); cb = __shfl(b_cb, wl); nn = __shfl(b_n, wl); nq = __shfl(b_q, wl);
            const Move m = (Move)__shfl((int)b_mv, wl);
            const uint64_t k = tkey(pos);
            const bool irr = irreversible(pos, m);
            if (lane == 0) { pkey[depth] = k; pirr[depth] = irr ? 1 : 0; }
            make_move(pos, m);
            // the scanned in-flight count + 1 (no second load), and no barrier per level: nothing a level stores
            // (in-flight count, path, position keys) is read before the walk has ended
            const int vlw = __shfl(b_vl, wl);
            if (lane == 0 && c.virtual_loss_active) { A.vl[child] = vlw + 1; if (child < ncached) L_vl[child] = vlw + 1; }
            prev_from = mv_from(m); prev_to = mv_to(m);
            node = child; ++depth;
            if (c.tt_merge) {
                // mcts.py:919: node = self._tt_get(board._transposition_key()) or best_child -- the walk continues from
                // the node registered LAST for this position; the edge child keeps the statistics its parent scores
                const int tn = tt_lookup(TK, TN, d.tt_cap, tt_key_of(pos), lane);
                if (tn >= 0 && tn != child) {
                    node = tn;
                    nc = A.nch[node]; cb = A.cbase[node]; nn = A.n[node]; nq = A.q[node];
                }
                if (lane == 0) epath[depth] = child;
            }
            if (lane == 0) path[depth] = node;
        }
        __syncthreads();
        // leaf: is_game_over() (checkmate, insufficient, stalemate, 75-move, fivefold) -> _terminal_value
        const int nlegal = gen_legal_wave(pos, smoves, spseudo, lane);
        const bool chk = in_check(pos);
        bool term = false;
        double tv = 0.0;
        if (nlegal == 0) { term = true; tv = chk ? -1.0 : c.draw_penalty; }
        else if (is_insufficient(pos)) { term = true; tv = c.draw_penalty; }
        else if (pos.halfmove >= 150) { term = true; tv = c.draw_penalty; }
        else {
            const uint64_t lk = tkey(pos);
            int cnt = 1;
            bool broke = false;
            for (int dd = depth - 1; dd >= 0; --dd) {
                if (pirr[dd]) { broke = true; break; }
                if (pkey[dd] == lk) ++cnt;
            }
            if (!broke)                                   // the game's reversible-move window: one entry per lane
                for (int i0 = 0; i0 < hist_len; i0 += 64) {
                    const int i = i0 + lane;
                    cnt += __popcll(__ballot(i < hist_len && H[i] == lk));
                }
            if (cnt >= 5) { term = true; tv = c.draw_penalty; }
        }
        int row = -1;
        uint64_t ckey = 0;
        bool cached = false;
        if (!term && c.eval_cache && d.ec.sets > 0 && nlegal <= M0_EC_MAXLEGAL) {
            // the key covers what the planes and the expansion depend on: tkey (pieces, turn, cleaned castling rights, legal ep)
            // and the two counters as plane_consts clips them
            const uint64_t hm = pos.halfmove < 99 ? pos.halfmove : 99, fm = pos.fullmove < 199 ? pos.fullmove : 199;
            ckey = mix64(tkey(pos) ^ ((hm << 8 | fm) * 0x9E3779B97F4A7C15ull)) | 1ull;
            const size_t eb = ((size_t)g * d.ec.sets + (size_t)((ckey >> 1) & (uint64_t)(d.ec.sets - 1))) * 4;
            const uint64_t k = lane < 4 ? d.ec.keys[eb + lane] : 0ull;
            const unsigned long long hit = __ballot(lane < 4 && k == ckey);
            if (hit) {
                const int way = __builtin_ctzll(hit);
                const float* src = d.ec.payload + (eb + way) * M0_EC_WORDS;
                float* dst = d.ec.hit_stage + ((size_t)g * (d.L + 1) + s) * M0_EC_WORDS;
                for (int i = lane; i < M0_EC_WORDS; i += 64) dst[i] = src[i];
                // the legal moves must agree (count and checksum: legal_sig): a mismatch = a key collision, served as a miss
                cached = __float_as_int(src[1]) == legal_sig(smoves, nlegal, lane);
                if (cached && lane == 0) { d.ec.stamps[eb + way] = ++gd->cache_clock; }
            }
        }
        // The same unexpanded node reached again in this pass (a batch of 96 descents over a young tree lands on the same leaf
        // many times; the virtual loss only spreads them): it shares the batch row of its first occurrence instead of being
        // evaluated twice in one forward.  The pending row sits in the node's child-base field, which means nothing until the
        // node is expanded: cbase <= -2  <=>  row -(cbase + 2) of this pass (expand_kernel resets it).
        bool shared = false;
        if (!term && !cached && c.eval_cache) {
            const int pend = A.cbase[node];
            if (A.nch[node] < 0 && pend <= -2) { shared = true; row = -(pend + 2); }
        }
        if (!term && !cached && !shared) {
            if (lane == 0) row = atomicAdd(d.row_counter + gd->net_id, 1) + gd->net_id * d.net_row_base;
            row = __shfl(row, 0);
            encode_nhwc(pos, d.x0 + (size_t)row * 64 * 32, lane);
            if (c.eval_cache && lane == 0 && A.nch[node] < 0) A.cbase[node] = -(row + 2);
        }
        if (!term && !shared) for (int i = lane; i < nlegal; i += 64) LM[(size_t)s * M0_MAX_CHILDREN + i] = smoves[i];
        if (lane == 0) {
            Sample smp; smp.pos = pos; smp.kind = term ? 3 : (cached ? 5 : (shared ? 6 : 1)); smp.leaf = node; smp.depth = depth; smp.row = row;
            smp.nlegal = nlegal; smp.ckey = cached ? 0ull : ckey;
            S[s] = smp;
        }
        __syncthreads();                                 // path[] (lane 0) before the wave reads it
        if (term) {                                      // mcts.py:747-751: terminal leaves back up immediately
            backprop(A, path, depth, tv, lane, c.tt_merge != 0, L_n, L_q, ncached);   // (the cached copies of the path's statistics follow)
            __syncthreads();
            // Later descents of this launch read these nodes' statistics again, the uncached ones from the arena: drop this CU's
            // vector-L1 lines so that they come from L2, where the stores above are.  (Round 4: a load of the same addresses
            // issued right after the barrier returned the OLD values from time to time -- the lines were resident from the
            // children scan and a store does not refresh them at once; seen as run-to-run differences in whole games.)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
    if (reinfer) emit_reinfer(nleaf);
    if (lane == 0) { gd->ctr_jitter = ctrj; gd->nsamples = nleaf + (reinfer ? 1 : 0); }
}

__device__ __forceinline__ float wave_max_f(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum_f(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Node._expand (mcts.py:135-225) for one leaf; returns false if the arena is exhausted
struct ExpandScratch {          // workgroup-shared staging of one expansion (raw priors / pruning only)
    float pr[M0_MAX_CHILDREN];
    uint16_t mv[M0_MAX_CHILDREN];
    uint16_t idx[M0_MAX_CHILDREN];
    uint8_t keep[M0_MAX_CHILDREN];
    float total;
    int bad;                    // the logits of the last expansion held a non-finite value (not cached)
};

__device__ bool expand_node(const Arena& A, int cap, GameDev* gd, const TreeCfg& c, int leaf, const Pos& pos,
                            const float* lg, const uint16_t* smoves, int n, int lane, bool is_root, ExpandScratch* X,
                            uint64_t* TK, int* TN, int tt_cap, bool reg_children, const float* cl = nullptr, float* cw = nullptr) {
    // legal moves of the leaf come from select (same position, same order): no second move generation
    // cl: the legal moves' logits from the evaluation cache (lg is then null); cw: cache payload to fill with them
    if (n <= 0) return true;
    // non-finite logits anywhere -> uniform priors (mcts.py:147-149)
    // (16-byte loads, all of a lane's 19 in flight together: one memory latency instead of 73 dependent-looking ones)
    bool bad = false;
    if (!cl) {
        const uint4* lg4 = reinterpret_cast<const uint4*>(lg);         // 4672 floats = 1168 x 16 B, rows are 16-byte aligned
        uint4 v[19];
#pragma unroll
        for (int k = 0; k < 19; ++k) { const int j = lane + 64 * k; v[k] = lg4[j < 1168 ? j : 1167]; }   // unconditional loads
        uint32_t acc = 0;                                               // all-ones exponent = inf or nan; no short-circuit
#pragma unroll
        for (int k = 0; k < 19; ++k) {
            acc |= (uint32_t)((v[k].x & 0x7f800000u) == 0x7f800000u) | (uint32_t)((v[k].y & 0x7f800000u) == 0x7f800000u) |
                   (uint32_t)((v[k].z & 0x7f800000u) == 0x7f800000u) | (uint32_t)((v[k].w & 0x7f800000u) == 0x7f800000u);
        }
        bad = acc != 0;
    }
    bad = __any(bad);
    if (lane == 0) X->bad = bad ? 1 : 0;
    float pr[4];
    int idx[4];
    Move mvv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = lane + 64 * k;
        pr[k] = 0.f; idx[k] = 0; mvv[k] = 0;
        if (i < n) { mvv[k] = smoves[i]; idx[k] = move_to_index(pos, mvv[k]); }
    }
    const bool raw = c.raw_legal_priors && c.legal_softmax && !is_root;
    if (raw) {
        // Node._expand_with_legal_priors (mcts.py:227-256), the reference's in-process-model branch (mcts.py:697-703):
        // priors = legal logits / their float32 sum (numpy pairwise order), uniform when the sum is <= 0 or not finite;
        // no softmax, no entropy noise, and only the LEGAL logits are looked at
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int i = lane + 64 * k; if (i < n) X->pr[i] = lg[idx[k]]; }
        __syncthreads();
        if (lane == 0) X->total = np_sum_f32_dev(X->pr, n);
        __syncthreads();
        const float total = X->total;
        const bool ok = isfinite(total) && total > 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int i = lane + 64 * k; pr[k] = i < n ? (ok ? X->pr[i] / total : 1.0f / (float)n) : 0.f; }
        __syncthreads();
    } else if (bad) {
#pragma unroll
        for (int k = 0; k < 4; ++k) pr[k] = 1.0f / (float)n;
    } else {
        // Softmax numerics: (logit - max) in float32 as torch does, exp/sum/divide in float64, result rounded
        // to float32.  Within one float32 ulp of the reference's torch.softmax (mcts.py:158-168) and
        // reproducible bit-for-bit on the host (oracle mode "engine"); entropy in float64.
        float mx = -3.0e38f;
        int cnt;                                      // size of the active distribution
        double ent = 0.0;
        double fsum = 1.0;                            // full-softmax normaliser (legal_softmax == 0)
        if (c.legal_softmax) {
            cnt = n;
            float l[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = lane + 64 * k;
                l[k] = i < n ? (cl ? cl[i] : lg[idx[k]]) : -3.0e38f;
                mx = fmaxf(mx, l[k]);
                if (cw && i < n && i < M0_EC_MAXLEGAL) cw[2 + i] = l[k];
            }
            mx = wave_max_f(mx);
            double e[4], sum = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int i = lane + 64 * k; e[k] = i < n ? exp((double)(l[k] - mx)) : 0.0; sum += e[k]; }
            sum = wave_sum_d(sum);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = lane + 64 * k;
                pr[k] = (float)(e[k] / sum);
                if (i < n) ent -= (double)pr[k] * log((double)pr[k] + 1e-8);
            }
        } else {
            cnt = 4672;
            for (int j = lane; j < 4672; j += 64) mx = fmaxf(mx, lg[j]);
            mx = wave_max_f(mx);
            double sum = 0.0;
            for (int j = lane; j < 4672; j += 64) sum += exp((double)(lg[j] - mx));
            sum = wave_sum_d(sum);
            fsum = sum;
            for (int j = lane; j < 4672; j += 64) {
                const double pj = (double)(float)(exp((double)(lg[j] - mx)) / sum);
                ent -= pj * log(pj + 1e-8);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int i = lane + 64 * k; pr[k] = i < n ? (float)(exp((double)(lg[idx[k]] - mx)) / sum) : 0.f; }
        }
        ent = wave_sum_d(ent);
        const double ratio = ent / fmax(1e-9, log((double)(n > 1 ? n : 1)));
        if (c.enable_entropy_noise && ratio > 0.9) {
            const uint64_t ctr = gd->ctr_noise;
            const uint64_t seed = gd->seed_noise;
            if (c.legal_softmax) {
                double dd[4], ds = 0.0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = lane + 64 * k;
                    dd[k] = 0.0;
                    if (i < n) { dd[k] = fmax((double)pr[k] + 0.1 * normal_at(seed, ctr + 2ull * (uint64_t)i), 1e-8); ds += dd[k]; }
                }
                ds = wave_sum_d(ds);
#pragma unroll
                for (int k = 0; k < 4; ++k) pr[k] = (float)(dd[k] / ds);
            } else {
                double ds = 0.0;
                for (int j = lane; j < 4672; j += 64)
                    ds += fmax((double)(float)(exp((double)(lg[j] - mx)) / fsum) + 0.1 * normal_at(seed, ctr + 2ull * (uint64_t)j), 1e-8);
                ds = wave_sum_d(ds);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = lane + 64 * k;
                    if (i < n) {
                        const int j = idx[k];
                        pr[k] = (float)(fmax((double)pr[k] + 0.1 * normal_at(seed, ctr + 2ull * (uint64_t)j), 1e-8) / ds);
                    }
                }
            }
            if (lane == 0) gd->ctr_noise = ctr + 2ull * (uint64_t)cnt;
        }
        // renormalise over the legal moves (mcts.py:205-212): float32 values, float64 sum rounded to float32
        double tot = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int i = lane + 64 * k; if (i < n) { if (!(pr[k] >= 0.f) || !isfinite(pr[k])) pr[k] = 0.f; tot += (double)pr[k]; } }
        tot = wave_sum_d(tot);
        const float totf = (float)tot;
        if (totf > 0.f && isfinite(totf)) {
#pragma unroll
            for (int k = 0; k < 4; ++k) pr[k] = pr[k] / totf;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) pr[k] = 1.0f / (float)n;
        }
    }
    // MCTS._prune_children (mcts.py:806-826): drop children below min_child_prior, then keep the max_children largest
    // priors (Python's stable sort: ties stay in move order; the kept children are then IN sorted order); priors are
    // not renormalised.  pos[k] = slot of this lane's k-th child in the node's child block, -1 = dropped.
    int slot[4] = {lane, lane + 64, lane + 128, lane + 192};
    int nkeep = n;
    if (c.max_children > 0 || c.min_child_prior > 0.0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = lane + 64 * k;
            if (i < n) { X->pr[i] = pr[k]; X->keep[i] = (c.min_child_prior > 0.0 && !((double)pr[k] >= c.min_child_prior)) ? 0 : 1; }
        }
        __syncthreads();
        int kept = 0;
        for (int j = 0; j < n; ++j) kept += X->keep[j];
        const bool topk = c.max_children > 0 && kept > c.max_children;
        nkeep = topk ? c.max_children : kept;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = lane + 64 * k;
            slot[k] = -1;
            if (i < n && X->keep[i]) {
                int r = 0;
                if (topk) { for (int j = 0; j < n; ++j) r += (X->keep[j] && (X->pr[j] > pr[k] || (X->pr[j] == pr[k] && j < i))) ? 1 : 0; }
                else { for (int j = 0; j < i; ++j) r += X->keep[j]; }
                if (r < nkeep) slot[k] = r;
            }
        }
        __syncthreads();
    }
    const int cb = gd->next;
    if (cb + nkeep > cap) { if (lane == 0) gd->overflow = 1; return false; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = lane + 64 * k;
        if (i < n && slot[k] >= 0) {
            const int ci = cb + slot[k];
            A.prior[ci] = (double)pr[k]; A.w[ci] = 0.0; A.q[ci] = 0.0; A.n[ci] = 0; A.vl[ci] = 0;
            A.cbase[ci] = -1; A.nch[ci] = -1; A.mv[ci] = mvv[k]; A.midx[ci] = (uint16_t)idx[k];
            // _register_children_in_tt (mcts.py:1330-1346): every child of an expanded NON-root node goes into the table
            // under the key of the position it leads to (run() registers only the fresh root itself, mcts.py:344-358)
            // A root that run() FOUND in the table and had to expand registers its children like any other node
            // (mcts.py:398-413): reg_children is false only for the brand-new root, which is registered itself instead.
            if (TK && reg_children) {
                Pos q = pos;
                make_move(q, mvv[k]);
                tt_insert(TK, TN, tt_cap, tt_key_of(q), ci);
            }
        }
    }
    if (TK && !reg_children && lane == 0) tt_insert(TK, TN, tt_cap, tt_key_of(pos), leaf);
    if (lane == 0) { A.cbase[leaf] = cb; A.nch[leaf] = (int16_t)nkeep; gd->next = cb + nkeep; }
    __syncthreads();
    return true;
}

__global__ __launch_bounds__(64) void expand_kernel(TreeDev d, TreeCfg c) {
    __shared__ ExpandScratch X;
    const int g = blockIdx.x, lane = threadIdx.x;
    GameDev* gd = &d.games[g];
    if (!gd->active) return;
    uint64_t* TK = c.tt_merge ? d.tt_keys + tt_table_of(d, g, gd) : nullptr;
    int* TN = c.tt_merge ? d.tt_nodes + tt_table_of(d, g, gd) : nullptr;
    const uint16_t* LM = d.leaf_moves + (size_t)g * (d.L + 1) * M0_MAX_CHILDREN;
    const int ns = gd->nsamples;
    if (ns <= 0) return;
    const Arena A = arena_of(d.t, g, gd->arena);
    const Sample* S = d.samples + (size_t)g * (d.L + 1);
    const int* P = d.paths + (size_t)g * (d.L + 1) * M0_MAX_DEPTH;
    int sims = 0;
    uint64_t evals = 0, hits = 0;
    for (int s = 0; s < ns; ++s) {
        const int kind = S[s].kind;
        const int* path = P + (size_t)s * M0_MAX_DEPTH;
        if (kind == 6) {                         // same leaf as an earlier sample of this pass: its row's value, no second expansion
            const float v = d.values[S[s].row];
            backprop(A, path, S[s].depth, (double)v, lane, c.tt_merge != 0);
            ++sims; ++hits;
            __syncthreads();
        } else if (kind == 5) {                  // evaluation served by the cache: expand from the staged payload
            const int leaf = S[s].leaf, depth = S[s].depth;
            const float* pay = d.ec.hit_stage + ((size_t)g * (d.L + 1) + s) * M0_EC_WORDS;
            const float v = pay[0];
            if (A.nch[leaf] < 0) {
                const Pos pos = S[s].pos;
                expand_node(A, d.t.cap, gd, c, leaf, pos, nullptr, LM + (size_t)s * M0_MAX_CHILDREN, S[s].nlegal, lane,
                            false, &X, TK, TN, d.tt_cap, true, pay + 2, nullptr);
            }
            backprop(A, path, depth, (double)v, lane, c.tt_merge != 0);
            ++sims; ++hits;
            __syncthreads();
        } else if (kind == 1 || kind == 2) {
            const int leaf = S[s].leaf, depth = S[s].depth, row = S[s].row;
            const float* lg = d.logits + (size_t)row * 4672;
            const float v = d.values[row];
            if (A.nch[leaf] < 0) {
                const Pos pos = S[s].pos;
                // a cacheable leaf (kind 1, key set by select): its evaluation goes into the least recently used way of its set
                float* cw = nullptr;
                size_t ce = 0;
                const uint64_t ckey = kind == 1 ? S[s].ckey : 0ull;
                if (ckey != 0 && isfinite(v)) {
                    const size_t eb = ((size_t)g * d.ec.sets + (size_t)((ckey >> 1) & (uint64_t)(d.ec.sets - 1))) * 4;
                    const uint64_t k = lane < 4 ? d.ec.keys[eb + lane] : 0ull;
                    const uint32_t st = lane < 4 ? d.ec.stamps[eb + lane] : 0xffffffffu;
                    const unsigned long long same = __ballot(lane < 4 && k == ckey), empty = __ballot(lane < 4 && k == 0ull);
                    int way;
                    if (same) way = __builtin_ctzll(same);
                    else if (empty) way = __builtin_ctzll(empty);
                    else {
                        uint32_t m = st; int w = lane;
                        for (int o = 2; o > 0; o >>= 1) { const uint32_t om = __shfl_xor(m, o); const int ow = __shfl_xor(w, o); if (om < m || (om == m && ow < w)) { m = om; w = ow; } }
                        way = __shfl(w, 0);
                    }
                    ce = eb + way;
                    cw = d.ec.payload + ce * M0_EC_WORDS;
                    if (lane == 0) d.ec.keys[ce] = 0ull;            // invalid while it is being rewritten
                }
                const bool ok = expand_node(A, d.t.cap, gd, c, leaf, pos, lg, LM + (size_t)s * M0_MAX_CHILDREN, S[s].nlegal, lane,
                                            kind == 2, &X, TK, TN, d.tt_cap, !(kind == 2 && gd->root_fresh), nullptr, cw);
                // an expansion that did not happen (node arena exhausted) must not leave the pending-row mark behind
                if (c.eval_cache && kind == 1 && lane == 0 && A.nch[leaf] < 0) A.cbase[leaf] = -1;
                const int sig = cw ? legal_sig(LM + (size_t)s * M0_MAX_CHILDREN, S[s].nlegal, lane) : 0;
                if (cw && lane == 0) {
                    if (ok && !X.bad) {
                        cw[0] = v; cw[1] = __int_as_float(sig);
                        __threadfence_block();
                        d.ec.keys[ce] = ckey; d.ec.stamps[ce] = ++gd->cache_clock;
                    }
                }
            }
            ++evals;
            if (kind == 1) {
                backprop(A, path, depth, (double)v, lane, c.tt_merge != 0);
                ++sims;
            } else {
                double rv = fmax(-1.0, fmin(1.0, (double)v));
                if (gd->flip_root_v) rv = -rv;
                if (lane == 0) {
                    gd->root_v = rv;
                    if (gd->root_q_from_v) { A.q[leaf] = rv; gd->root_q_from_v = 0; }
                }
            }
            __syncthreads();
        } else if (kind == 3) {
            ++sims;
        } else if (kind == 4) {                   // value of a reused root (mcts.py:359-371); v is read only if root.n == 0
            double rv = fmax(-1.0, fmin(1.0, (double)d.values[S[s].row]));
            if (gd->flip_root_v) rv = -rv;
            if (lane == 0) gd->root_v = rv;
            ++evals;
        }
    }
    // release virtual losses of the whole batch (the reference's inflight dict dies with the batch)
    // (integer atomics: one lane per sample, any order gives the same counts)
    if (c.virtual_loss_active) {
        for (int s = lane; s < ns; s += 64) {
            const int kind = S[s].kind;
            if (kind == 1 || kind == 3 || kind == 5 || kind == 6) {
                // the in-flight counts sit on the EDGE children (tt_merge: the walk itself may have continued elsewhere)
                const int* path = (c.tt_merge ? d.epaths + (size_t)g * (d.L + 1) * M0_MAX_DEPTH : P) + (size_t)s * M0_MAX_DEPTH;
                for (int dd = 1; dd <= S[s].depth; ++dd) atomicSub(&A.vl[path[dd]], 1);
            }
        }
    }
    __syncthreads();
    const int done = gd->sims_done + sims;
    const int root = gd->root;
    const bool fin = (A.nch[root] >= 0 || gd->overflow) && done >= gd->sims_target;
    if (lane == 0) {
        gd->sims_done = done;
        gd->evals += evals;
        gd->cache_hits += hits;
        gd->finished = fin ? 1 : 0;
        gd->root_n = A.n[root];
        gd->root_q = A.q[root];
    }
    if (fin) {
        RootResult* R = d.results + g;
        const int k = A.nch[root] > 0 ? A.nch[root] : 0;
        const int cb = A.cbase[root];
        for (int i = lane; i < k; i += 64) {
            R->child_node[i] = cb + i; R->child_n[i] = A.n[cb + i];
            R->child_mv[i] = A.mv[cb + i]; R->child_idx[i] = A.midx[cb + i];
            R->child_prior[i] = A.prior[cb + i]; R->child_q[i] = A.q[cb + i];
        }
        if (lane == 0) { R->nchild = k; R->root_n = A.n[root]; R->root_q = A.q[root]; }
    }
}

// Re-root: child_slot >= 0 keeps that child's subtree (compacted into the other arena half);
// child_slot < 0 starts a fresh tree.
__global__ __launch_bounds__(64) void advance_kernel(TreeDev d, const int* game_ids, const int* child_slots, int count) {
    const int j = blockIdx.x, lane = threadIdx.x;
    if (j >= count) return;
    const int g = game_ids[j], slot = child_slots[j];
    GameDev* gd = &d.games[g];
    if (slot <= -2) {
        // match engine with per-side tables (tt_sides == 2): side = the network that searches now.  Nothing is compacted or
        // cleared between the searches of a game; the root is whatever node the side's table holds for the position
        // (MCTS.run: root = self._tt_get(key), mcts.py:343), else a new node.
        const int s = gd->net_id & 1;
        const int prev_side = gd->arena & 1, prev_next = gd->next;
        if (slot == -3) {
            uint4* tk = reinterpret_cast<uint4*>(d.tt_keys + (size_t)g * 2 * d.tt_cap);
            for (int i = lane; i < d.tt_cap; i += 64) tk[i] = make_uint4(0, 0, 0, 0);       // both tables: 2 * tt_cap keys
        }
        __syncthreads();
        int nxt = slot == -3 ? 0 : (s == prev_side ? prev_next : gd->side_next[s]);
        const Arena A = arena_of(d.t, g, s);
        const uint64_t* TK = d.tt_keys + ((size_t)g * 2 + s) * d.tt_cap;
        const int* TN = d.tt_nodes + ((size_t)g * 2 + s) * d.tt_cap;
        int node = slot == -3 ? -1 : tt_lookup(TK, TN, d.tt_cap, tt_key_of(gd->root_pos), lane);
        // a half that cannot take another search's worth of nodes starts over: its table is dropped and the search begins from a
        // fresh root (what the reference's _cleanup_memory does to an over-full table); engine.arena_nodes sizes it for a game
        if (nxt + (d.search_nodes > 4 * M0_MAX_CHILDREN ? d.search_nodes : 4 * M0_MAX_CHILDREN) >= d.t.cap) {
            uint4* tk = reinterpret_cast<uint4*>(d.tt_keys + ((size_t)g * 2 + s) * d.tt_cap);
            for (int i = lane; i < d.tt_cap / 2; i += 64) tk[i] = make_uint4(0, 0, 0, 0);
            nxt = 0; node = -1;
            __syncthreads();
        }
        const bool found = node >= 0;
        if (lane == 0) {
            if (slot != -3) gd->side_next[prev_side] = prev_next;
            else { gd->side_next[0] = 0; gd->side_next[1] = 0; }
            if (!found) {
                node = nxt;
                A.prior[node] = 0.0; A.w[node] = 0.0; A.q[node] = 0.0; A.n[node] = 0; A.vl[node] = 0; A.cbase[node] = -1; A.nch[node] = -1;
                A.mv[node] = 0; A.midx[node] = 0;
                nxt = node + 1;
            }
            gd->arena = s; gd->root = node; gd->next = nxt; gd->overflow = 0;
            gd->root_fresh = found ? 0 : 1; gd->root_found = found ? 1 : 0;
        }
        return;
    }
    if (slot < 0) {
        const Arena D = arena_of(d.t, g, 0);
        if (d.tt_keys) {                             // a fresh root starts with an empty position table
            uint4* tk = reinterpret_cast<uint4*>(d.tt_keys + (size_t)g * d.tt_sides * d.tt_cap);
            for (int i = lane; i < d.tt_sides * d.tt_cap / 2; i += 64) tk[i] = make_uint4(0, 0, 0, 0);
        }
        if (lane == 0) {
            D.prior[0] = 0.0; D.w[0] = 0.0; D.q[0] = 0.0; D.n[0] = 0; D.vl[0] = 0; D.cbase[0] = -1; D.nch[0] = -1;
            D.mv[0] = 0; D.midx[0] = 0;
            gd->arena = 0; gd->root = 0; gd->next = 1; gd->overflow = 0;
        }
        return;
    }
    const int a = gd->arena;
    const Arena Sx = arena_of(d.t, g, a), D = arena_of(d.t, g, a ^ 1);
    const int r = Sx.cbase[gd->root] + slot;
    if (lane == 0) {
        D.prior[0] = Sx.prior[r]; D.w[0] = Sx.w[r]; D.q[0] = Sx.q[r]; D.n[0] = Sx.n[r]; D.vl[0] = 0;
        D.cbase[0] = Sx.cbase[r]; D.nch[0] = Sx.nch[r]; D.mv[0] = Sx.mv[r]; D.midx[0] = Sx.midx[r];
    }
    __syncthreads();
    int head = 0, tail = 1;
    while (head < tail) {
        const int lim = tail < head + 64 ? tail : head + 64;
        const int i = head + lane;
        const int nc_i = i < lim ? (int)D.nch[i] : -1;
        const int cb_i = i < lim ? D.cbase[i] : -1;     // still the OLD child base
        unsigned long long mask = __ballot(nc_i > 0);
        while (mask) {
            const int b = __builtin_ctzll(mask);
            mask &= mask - 1;
            const int nc = __shfl(nc_i, b), ocb = __shfl(cb_i, b);
            const int ncb = tail;
            for (int k = lane; k < nc; k += 64) {
                const int so = ocb + k, dn = ncb + k;
                D.prior[dn] = Sx.prior[so]; D.w[dn] = Sx.w[so]; D.q[dn] = Sx.q[so]; D.n[dn] = Sx.n[so]; D.vl[dn] = 0;
                D.cbase[dn] = Sx.cbase[so]; D.nch[dn] = Sx.nch[so]; D.mv[dn] = Sx.mv[so]; D.midx[dn] = Sx.midx[so];
            }
            if (lane == 0) D.cbase[head + b] = ncb;
            tail += nc;
        }
        __syncthreads();
        head = lim;
    }
    if (lane == 0) { gd->arena = a ^ 1; gd->root = 0; gd->next = tail; gd->overflow = 0; }
}

hipError_t launch_select(const TreeDev& d, const TreeCfg& c, hipStream_t st) {
    static DeviceOnce once;
    hipError_t e = once.run([] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&select_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, M0_SEL_CACHE_BYTES);
    });
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(select_kernel, dim3(d.G), dim3(64), M0_SEL_CACHE_BYTES, st, d, c);
    return hipGetLastError();
}
hipError_t launch_expand(const TreeDev& d, const TreeCfg& c, hipStream_t st) {
    hipLaunchKernelGGL(expand_kernel, dim3(d.G), dim3(64), 0, st, d, c);
    return hipGetLastError();
}
hipError_t launch_advance(const TreeDev& d, const int* game_ids_dev, const int* child_slots_dev, int count, hipStream_t st) {
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(advance_kernel, dim3(count), dim3(64), 0, st, d, game_ids_dev, child_slots_dev, count);
    return hipGetLastError();
}

// ---- position-wise encoding.py on device: one wave per position ----
__global__ __launch_bounds__(64) void encode_positions_kernel(const Pos* pos, int n, float* planes, _Float16* nhwc,
                                                              uint8_t* mask, int32_t* nlegal, uint16_t* moves, int32_t* idxs) {
    __shared__ Move smoves[M0_MAX_MOVES];
    __shared__ Move spseudo[M0_MAX_MOVES];
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= n) return;
    const Pos p = pos[i];
    if (planes) {
        float c7[7];
        plane_consts(p, c7);
        const int s = (7 - (lane >> 3)) * 8 + (lane & 7);
        const int pl = piece_plane(p, s);
        float* o = planes + (size_t)i * 19 * 64;
        for (int k = 0; k < 12; ++k) o[k * 64 + lane] = pl == k ? 1.f : 0.f;
        for (int k = 0; k < 7; ++k) o[(12 + k) * 64 + lane] = c7[k];
    }
    if (nhwc) encode_nhwc(p, nhwc + (size_t)i * 64 * 32, lane);
    const int k = gen_legal_wave(p, smoves, spseudo, lane);   // the search's generator (same list as gen_legal)
    __syncthreads();
    if (mask) for (int j = lane; j < 4672; j += 64) mask[(size_t)i * 4672 + j] = 0;
    __syncthreads();
    for (int j = lane; j < M0_MAX_MOVES; j += 64) {
        int idx = -1;
        Move m = 0;
        if (j < k) { m = smoves[j]; idx = move_to_index(p, m); if (mask && idx >= 0) mask[(size_t)i * 4672 + idx] = 1; }
        if (moves) moves[(size_t)i * M0_MAX_MOVES + j] = m;
        if (idxs) idxs[(size_t)i * M0_MAX_MOVES + j] = idx;
    }
    if (lane == 0 && nlegal) nlegal[i] = k;
}

hipError_t launch_encode_positions(const Pos* pos_dev, int n, float* planes, _Float16* nhwc, uint8_t* mask,
                                   int32_t* nlegal, uint16_t* moves, int32_t* idxs, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(encode_positions_kernel, dim3(n), dim3(64), 0, st, pos_dev, n, planes, nhwc, mask, nlegal, moves, idxs);
    return hipGetLastError();
}

// ---- SSL training targets (azchess/ssl_algorithms.py:51-143, 256-557), one wave per position, lane = tensor square.
// All geometry in tensor space (row 0 = rank 8) exactly as the reference computes it, quirks included (SURVEY B-5):
// white pawns attack toward higher row index; the reference's pin map is identically zero (it ANDs a map that is
// non-zero only on the candidate square with one non-zero only on the next square).
// out f32 [n][17][64]: piece one-hot (13), threat, pin, fork, control.
__global__ __launch_bounds__(64) void ssl_targets_kernel(const Pos* pos, int n, float* out) {
    __shared__ int8_t pc[64];          // plane index 0..11 (white P..K, black P..K) or -1, by tensor square
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= n) return;
    const Pos p = pos[i];
    pc[lane] = (int8_t)piece_plane(p, (7 - (lane >> 3)) * 8 + (lane & 7));
    __syncthreads();
    const int r = lane >> 3, c = lane & 7;
    const bool stm_white = p.turn == WHITE;
    auto at = [&](int rr, int cc) -> int { return ((unsigned)rr < 8u && (unsigned)cc < 8u) ? (int)pc[rr * 8 + cc] : -2; };
    const int KN[8][2] = {{-2, -1}, {-2, 1}, {-1, -2}, {-1, 2}, {1, -2}, {1, 2}, {2, -1}, {2, 1}};
    const int KG[8][2] = {{-1, -1}, {-1, 0}, {-1, 1}, {0, -1}, {0, 1}, {1, -1}, {1, 0}, {1, 1}};
    int wa = 0, ba = 0;
    // pawns: white pawn at (r0,c0) attacks (r0+1,c0+-1); black pawn attacks (r0-1,c0+-1)
    for (int dc = -1; dc <= 1; dc += 2) {
        if (at(r - 1, c + dc) == 0) ++wa;
        if (at(r + 1, c + dc) == 6) ++ba;
    }
    for (int k = 0; k < 8; ++k) {
        int q = at(r + KN[k][0], c + KN[k][1]);
        if (q == 1) ++wa; else if (q == 7) ++ba;
        q = at(r + KG[k][0], c + KG[k][1]);
        if (q == 5) ++wa; else if (q == 11) ++ba;
    }
    const int me = pc[lane];
    const bool own_tactical = me >= 0 && ((stm_white && me >= 1 && me <= 5) || (!stm_white && me >= 7 && me <= 11));
    const int mytype = me >= 0 ? me % 6 : -1;          // 0 P,1 N,2 B,3 R,4 Q,5 K
    int forks = 0;
    if (own_tactical && (mytype == 1 || mytype == 5)) {
        for (int k = 0; k < 8; ++k) {
            const int q = mytype == 1 ? at(r + KN[k][0], c + KN[k][1]) : at(r + KG[k][0], c + KG[k][1]);
            if (q >= 0 && ((q < 6) != stm_white)) ++forks;
        }
    }
    for (int k = 0; k < 8; ++k) {                       // the 8 ray directions (KG is also the ray set)
        const int dr = KG[k][0], dc = KG[k][1];
        const bool diag = dr != 0 && dc != 0;
        int rr = r + dr, cc = c + dc, q = -1;
        while ((unsigned)rr < 8u && (unsigned)cc < 8u) {
            q = pc[rr * 8 + cc];
            if (q >= 0) break;
            rr += dr; cc += dc;
        }
        if (q < 0) continue;
        const int t = q % 6;
        // the first piece along the ray attacks this square if it slides along the ray
        if (t == 4 || (diag && t == 2) || (!diag && t == 3)) { if (q < 6) ++wa; else ++ba; }
        // and this square's own slider attacks that first piece if it is an enemy
        if (own_tactical && (mytype == 4 || (diag && mytype == 2) || (!diag && mytype == 3)) && ((q < 6) != stm_white)) ++forks;
    }
    float* o = out + (size_t)i * 17 * 64;
    for (int k = 0; k < 12; ++k) o[k * 64 + lane] = me == k ? 1.f : 0.f;
    o[12 * 64 + lane] = me < 0 ? 1.f : 0.f;
    o[13 * 64 + lane] = (stm_white ? ba : wa) > 0 ? 1.f : 0.f;
    o[14 * 64 + lane] = 0.f;
    o[15 * 64 + lane] = (own_tactical && forks >= 2) ? 1.f : 0.f;
    o[16 * 64 + lane] = wa > ba ? 1.f : (wa < ba ? -1.f : 0.f);
}

hipError_t launch_ssl_targets(const Pos* pos_dev, int n, float* out_dev, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(ssl_targets_kernel, dim3(n), dim3(64), 0, st, pos_dev, n, out_dev);
    return hipGetLastError();
}
