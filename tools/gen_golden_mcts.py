#!/usr/bin/env python3
"""Golden vectors for the search / encoding / self-play half of the hot path, produced by running the REAL reference
code (azchess/encoding.py, mcts.py, draw.py, selfplay/internal.py, arena.py) in the build container.

How the reference runs here without python-chess: tools/refshim.py registers the oracle's rules engine as module
`chess` (SURVEY App. A.5 surface) and a synthetic `azchess` package pointing at /root/reference/azchess.  What the
goldens therefore pin is everything the reference's OWN files compute -- plane layout, move-index arithmetic, softmax /
entropy-noise / renormalisation of priors, PUCT scores and tie-breaks, backup signs, Dirichlet mixing, cpuct schedules,
batched leaf collection, policy targets, move sampling, temperature, resign rule, draw adjudication order, result
mapping, NPZ assembly -- on top of the oracle's move generator (itself pinned by perft and the reference's FEN
fixtures, tests/test_oracle_chess.py).

Randomness: the reference draws from Python `random` and numpy's global generator; here those entry points are
re-routed (refshim.injected) to the counter streams every implementation in this repository shares, so the same
draws reach the reference, the oracle and the HIP engine.  The evaluator is tests/hash_net.py (integer hashing, bit
reproducible anywhere) behind the reference's `inference_backend.infer_np` seam (mcts.py:618-621).

Transposition table: `tt` in a case says how MCTS._tt_get (mcts.py:1231-1239) was left:
  "on"   untouched reference code (search graph is a DAG through the table);
  "off"  MCTS._tt_get patched to return None: tree-only search, a fresh root per run() -- the mode the HIP engine's
         default (tree, no merging) is compared with.
Finding recorded by this script (section "tt_across_moves"): with the table on, the reference's second run() of a
game on the same MCTS object raises RuntimeError("... zero visits ...") because run() never registers a fresh root's
children (mcts.py:344-358) while every grandchild is registered (mcts.py:654-666), so the new root's children are all
bypassed at mcts.py:919 and keep n == 0.

Outputs (data only): tests/golden/ref_encoding.npz, ref_mcts.json.gz, ref_selfplay.json.gz, ref_worker_<name>.npz
         ref_mcts_vl.json.gz (the reference's own virtual-loss lines executed: class VLOn)
Usage: python tools/gen_golden_mcts.py [encoding] [mcts] [mcts_vl] [selfplay] [worker]     (default: all)
"""
from __future__ import annotations

import gzip
import json
import logging
import os
import queue
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..")
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import refshim  # noqa: E402

chess = refshim.install()
import azchess.encoding as renc  # noqa: E402
import azchess.mcts as rmcts  # noqa: E402
import azchess.draw as rdraw  # noqa: E402

from oracle import chess_py as ch  # noqa: E402
from tests.hash_net import HashNet  # noqa: E402

logging.disable(logging.CRITICAL)
OUT = os.path.join(ROOT, "tests", "golden")

BASE_MCTS = {"cpuct": 2.5, "cpuct_start": 3.0, "cpuct_end": 2.0, "cpuct_plies": 40, "dirichlet_alpha": 0.3,
             "dirichlet_frac": 0.25, "dirichlet_plies": 30, "selection_jitter": 0.05, "fpu_reduction": 0.1,
             "draw_penalty": -0.05, "virtual_loss": 1.0, "legal_softmax": True, "enable_entropy_noise": True,
             "no_instant_backtrack": True, "playout_random_frac": 0.0, "num_threads": 1, "enable_memory_cleanup": False,
             "encoder_cache": True}

FENS = [ch.START_FEN,
        "r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1",
        "8/2p5/3p4/KP5r/1R3p1k/8/4P1P1/8 w - - 0 1",
        "r1bq1rk1/pp2bppp/2n1pn2/2pp4/3P1B2/2PBPN2/PP1N1PPP/R2QK2R b KQ - 3 8",
        "6k1/5ppp/8/8/8/8/5PPP/3R2K1 w - - 0 1",
        "7k/5Q2/5K2/8/8/8/8/8 w - - 0 1",
        "rnbq1k1r/pp1Pbppp/2p5/8/2B5/8/PPP1NnPP/RNBQK2R w KQ - 1 8",
        "r4rk1/1pp1qppp/p1np1n2/2b1p1B1/2B1P1b1/P1NP1N2/1PP1QPPP/R4RK1 w - - 0 10",
        "8/P7/8/8/8/8/7p/K6k w - - 0 1",
        "4k3/8/8/3pP3/8/8/8/4K3 w - d6 0 2"]


def mv_code(m):
    return m.from_square | (m.to_square << 6) | ((m.promotion or 0) << 12)


def dump_json(name, obj):
    p = os.path.join(OUT, name)
    with gzip.open(p, "wt") as f:
        json.dump(obj, f, separators=(",", ":"))
    print(f"wrote {name}: {os.path.getsize(p)} bytes")


# ------------------------------------------------------------------------------------------------ A. encoding
def gen_encoding():
    rows = json.load(gzip.open(os.path.join(OUT, "tactical_legal_counts.json.gz"), "rt"))
    fens = [r[0] for r in rows] + FENS + [
        "r3k2r/8/8/8/8/8/8/R3K2R w KQkq - 0 1", "r3k2r/8/8/8/8/8/8/R3K2R b KQkq - 0 1",      # castling both ways
        "4k3/P6P/8/8/8/8/p6p/4K3 w - - 0 1", "4k3/P6P/8/8/8/8/p6p/4K3 b - - 0 1",          # promotions
        "rnbqkbnr/ppp1p1pp/8/3pPp2/8/8/PPPP1PPP/RNBQKBNR w KQkq f6 0 3",                   # e.p.
        "8/8/8/8/8/8/8/K1k5 w - - 120 250"]                                                # counters saturate
    enc = renc.MoveEncoder()
    N = len(fens)
    bits = np.zeros((N, 17, 8), np.uint8)
    counters = np.zeros((N, 2), np.float32)
    nlegal = np.zeros(N, np.int16)
    moves_cat, idx_cat, decode_probe = [], [], []
    rng = np.random.default_rng(7)
    for i, fen in enumerate(fens):
        b = chess.Board(fen)
        planes = renc.encode_board(b)
        assert planes.dtype == np.float32 and planes.shape == (19, 8, 8)
        head = planes[:17]
        assert np.all((head == 0.0) | (head == 1.0))
        bits[i] = np.packbits(head.astype(np.uint8).reshape(17, 64), axis=1)
        for k in (17, 18):
            assert np.all(planes[k] == planes[k, 0, 0])
            counters[i, k - 17] = planes[k, 0, 0]
        legal = list(b.legal_moves)
        idxs = [renc.move_to_index(b, m) for m in legal]
        assert idxs == [enc.encode_move(b, m) for m in legal]
        mask = enc.get_legal_actions(b)
        assert mask.dtype == bool and mask.shape == (4672,) and int(mask.sum()) == len(set(idxs)) == len(legal)
        assert all(mask[j] for j in idxs)
        for m, j in zip(legal, idxs):                       # decode round trip (encoding.py:174-229)
            d = enc.decode_move(b, j)
            assert (d.from_square, d.to_square) == (m.from_square, m.to_square), (fen, m, d)
        nlegal[i] = len(legal)
        moves_cat += [mv_code(m) for m in legal]
        idx_cat += idxs
        if i % 25 == 0:                                     # decode of arbitrary (mostly illegal) indices: the fallbacks
            for j in rng.integers(0, 4672, size=6):
                decode_probe.append((i, int(j), mv_code(enc.decode_move(b, int(j)))))
        if i % 2000 == 0:
            print("encoding", i, "/", N, flush=True)
    # error behaviour (encoding.py:120-121): an illegal move raises ValueError
    try:
        renc.move_to_index(chess.Board(), chess.Move.from_uci("e2e5"))
        raise SystemExit("reference accepted an illegal move")
    except ValueError:
        pass
    np.savez_compressed(os.path.join(OUT, "ref_encoding.npz"), fens=np.array(fens), plane_bits=bits, counters=counters,
                        nlegal=nlegal, moves=np.array(moves_cat, np.uint16), idx=np.array(idx_cat, np.int16),
                        decode_probe=np.array(decode_probe, np.int32),
                        hflip=renc.build_horizontal_flip_permutation().astype(np.int16),
                        rot180=renc.build_rotate180_permutation().astype(np.int16))
    print("wrote ref_encoding.npz:", os.path.getsize(os.path.join(OUT, "ref_encoding.npz")), "bytes;", N, "positions,",
          len(idx_cat), "moves")


# ------------------------------------------------------------------------------------------------ B. mcts
class TTOff:
    """MCTS._tt_get -> None (tree-only search)."""

    def __enter__(self):
        self.saved = rmcts.MCTS._tt_get
        rmcts.MCTS._tt_get = lambda self_, key: None
        return self

    def __exit__(self, *a):
        rmcts.MCTS._tt_get = self.saved


class VLOn:
    """Virtual loss exactly as the reference wrote it and never calls it.  MCTS._select takes `inflight_counts` and applies /
    maintains it (mcts.py:851, 889-890, 922-923), but its only callers (_collect_leaf_position mcts.py:745, _run_simulation
    :777) never pass one.  This patch passes one -- a dict that lives for ONE batch of _run_simulations_parallel_batched
    (mcts.py:535-558: batch k of a run() = calls k*L .. k*L+batch_n-1 of _collect_leaf_position) -- and changes nothing else:
    _select's body, the batch loop, expansion and backup are the reference's own code."""

    def __enter__(self):
        self.saved = (rmcts.MCTS._select, rmcts.MCTS._collect_leaf_position, rmcts.MCTS._run_simulations_parallel_batched)
        o_select, o_collect, o_batched = self.saved

        def select(self_, board, root, inflight_counts=None, base_ply=0):
            if inflight_counts is None:
                inflight_counts = getattr(self_, "_m0_inflight", None)
            return o_select(self_, board, root, inflight_counts=inflight_counts, base_ply=base_ply)

        def collect(self_, board, root, leaf_samples, append_lock):
            L = int(getattr(self_.cfg, "inference_batch_size", None) or getattr(self_.cfg, "simulation_batch_size", 96))
            if self_._m0_calls % L == 0:
                self_._m0_inflight = {}
            self_._m0_calls += 1
            return o_collect(self_, board, root, leaf_samples, append_lock)

        def batched(self_, board, root, num_simulations):
            self_._m0_calls = 0
            self_._m0_inflight = {}
            try:
                return o_batched(self_, board, root, num_simulations)
            finally:
                self_._m0_inflight = None

        rmcts.MCTS._select, rmcts.MCTS._collect_leaf_position, rmcts.MCTS._run_simulations_parallel_batched = select, collect, batched
        return self

    def __exit__(self, *a):
        rmcts.MCTS._select, rmcts.MCTS._collect_leaf_position, rmcts.MCTS._run_simulations_parallel_batched = self.saved


class Both:
    def __init__(self, *cms):
        self.cms = cms

    def __enter__(self):
        for c in self.cms:
            c.__enter__()
        return self

    def __exit__(self, *a):
        for c in reversed(self.cms):
            c.__exit__(*a)


class Nop:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        pass


class TorchModel:
    """The evaluator as an in-process `model` (mcts.py:672-687, 1053-1078): called with a torch tensor, returns tensors."""

    class _Cfg:
        policy_size = 4672

    cfg = _Cfg()

    def __init__(self, net):
        self.net = net

    def __call__(self, x):
        import torch
        p, v = self.net.infer_np(x.detach().cpu().numpy())
        return torch.from_numpy(p), torch.from_numpy(v)


def new_mcts(mcfg, net, model_path=False):
    if model_path:          # the branch that expands non-root leaves with Node._expand_with_legal_priors (mcts.py:697-703)
        return rmcts.MCTS(rmcts.MCTSConfig.from_dict(dict(mcfg)), TorchModel(net), device="cpu", inference_backend=None)
    return rmcts.MCTS(rmcts.MCTSConfig.from_dict(dict(mcfg)), None, device="cpu", inference_backend=net)


def root_dump(mc, visit_counts, pi, root_q):
    root = mc._last_root
    kids = list(root.children.values())
    assert [c.move for c in kids] == list(visit_counts.keys())
    nz = np.nonzero(pi)[0]
    return {"moves": [mv_code(c.move) for c in kids], "idx": [int(c.move_idx) for c in kids],
            "n": [int(c.n) for c in kids], "prior": [float(c.prior) for c in kids], "q": [float(c.q) for c in kids],
            "w": [float(c.w) for c in kids], "root_n": int(root.n), "root_q_node": float(root.q), "root_q": float(root_q),
            "pi_idx": [int(j) for j in nz], "pi_val": [float(pi[j]) for j in nz], "sims": int(mc._last_sims_run)}


def gen_mcts():
    out = {"base_mcts": BASE_MCTS, "fens": FENS}

    # -- _cpuct_at (mcts.py:927-944)
    tab = []
    for extra in ({}, {"cpuct_start": None, "cpuct_end": None, "cpuct_plies": 0}, {"cpuct_c_base": 19652.0, "cpuct_c_init": 1.25},
                  {"cpuct_start": 1.5, "cpuct_end": 4.0, "cpuct_plies": 7}):
        mc = new_mcts(dict(BASE_MCTS, **extra), None)
        tab.append({"cfg": extra, "values": [mc._cpuct_at(p) for p in range(-2, 64)]})
    out["cpuct_at"] = {"plies": list(range(-2, 64)), "tables": tab}

    # -- _backpropagate (mcts.py:946-953)
    bp = []
    rng = np.random.default_rng(11)
    mc = new_mcts(BASE_MCTS, None)
    for _ in range(24):
        depth = int(rng.integers(1, 12))
        nodes = [rmcts.Node() for _ in range(depth)]
        for nd in nodes:
            nd.n = int(rng.integers(0, 50)); nd.w = float(rng.normal()) * nd.n * 0.3; nd.q = nd.w / nd.n if nd.n else 0.0
        before = [[nd.n, nd.w] for nd in nodes]
        vals = [float(x) for x in rng.uniform(-1.4, 1.4, size=3)]
        for v in vals:
            mc._backpropagate(nodes, v)
        bp.append({"before": before, "values": vals, "after": [[nd.n, nd.w, nd.q] for nd in nodes]})
    out["backpropagate"] = bp

    # -- Node._expand (mcts.py:135-225): priors for every mode
    ex = []
    uid = 0
    for fi, fen in enumerate(FENS):
        for legal_only in (True, False):
            for noise in (True, False):
                for sharp, poison in ((0.5, False), (8.0, False), (30.0, False), (6.0, True)):
                    uid += 1
                    net = HashNet(seed=fi + 1, sharp=sharp, poison=poison)
                    b = chess.Board(fen)
                    lg, _ = net.infer_np(renc.encode_board(b))
                    st = refshim.Streams(4321, uid)
                    nd = rmcts.Node()
                    raised = None
                    with refshim.injected(st):
                        try:
                            nd._expand(b, lg[0], encoder=renc.MoveEncoder(), legal_only=legal_only, allow_noise=noise)
                        except Exception as e:          # non-finite logits: mcts.py:147-149 leaves `idxs` unbound -> :214 raises
                            raised = type(e).__name__
                            assert not np.all(np.isfinite(lg[0]))
                    kids = list(nd.children.values())
                    ex.append({"fen": fi, "uid": uid, "legal_only": legal_only, "noise": noise, "sharp": sharp, "poison": poison,
                               "net_seed": fi + 1, "moves": [mv_code(c.move) for c in kids], "idx": [int(c.move_idx) for c in kids],
                               "prior": [float(c.prior) for c in kids], "noise_draws": st.noise.ctr, "raised": raised,
                               "finite": bool(np.all(np.isfinite(lg[0])))})
    out["expand"] = {"seed": 4321, "cases": ex}

    # -- _add_dirichlet (mcts.py:955-992)
    dr = []
    for k, (alpha, frac) in enumerate(((0.3, 0.25), (0.03, 0.5), (1.0, 0.1), (2.5, 0.25))):
        for fi in (0, 1, 5):
            b = chess.Board(FENS[fi])
            nd = rmcts.Node()
            lg, _ = HashNet(seed=9, sharp=8.0).infer_np(renc.encode_board(b))
            nd._expand(b, lg[0], legal_only=True, allow_noise=False)
            before = [float(c.prior) for c in nd.children.values()]
            mc = new_mcts(dict(BASE_MCTS, dirichlet_alpha=alpha, dirichlet_frac=frac), None)
            st = refshim.Streams(99, 10 * k + fi)
            with refshim.injected(st):
                mc._add_dirichlet(nd)
                mid = [float(c.prior) for c in nd.children.values()]
                mc._add_dirichlet(nd)                      # applied again to already-noised priors, as on a reused root
            dr.append({"fen": fi, "uid": 10 * k + fi, "alpha": alpha, "frac": frac, "before": before, "after": mid,
                       "after2": [float(c.prior) for c in nd.children.values()], "draws": st.dirichlet.ctr})
    out["dirichlet"] = {"seed": 99, "cases": dr}

    # -- _select, one level (mcts.py:851-925): hand-set statistics on the children of an expanded node
    sel = []
    rng = np.random.default_rng(5)
    for k in range(40):
        fi = k % len(FENS)
        b = chess.Board(FENS[fi])
        legal = list(b.legal_moves)
        if not legal:
            continue
        parent = rmcts.Node()
        parent.n = int(rng.integers(0, 400)); parent.q = float(rng.uniform(-0.8, 0.8)); parent.w = parent.q * parent.n
        parent.expanded = True
        pri = rng.dirichlet([0.5] * len(legal))
        stats = []
        for m, p in zip(legal, pri):
            c = rmcts.Node(prior=float(p), move=m, parent=parent)
            if rng.random() < 0.5:
                c.n = int(rng.integers(1, 60)); c.q = float(rng.uniform(-1, 1)); c.w = c.q * c.n
            parent.children[m] = c
            stats.append([c.n, c.q, float(p)])
        jit = [0.05, 0.0, 0.01][k % 3]
        mc = new_mcts(dict(BASE_MCTS, selection_jitter=jit, fpu_reduction=[0.1, 0.3][k % 2]), None)
        st = refshim.Streams(555, k)
        with TTOff(), refshim.injected(st):
            node, path, _ = mc._select(b.copy(), parent)
        chosen = list(parent.children.values()).index(path[1])
        sel.append({"fen": fi, "uid": k, "parent_n": parent.n, "parent_q": parent.q, "children": stats, "jitter": jit,
                    "fpu_reduction": [0.1, 0.3][k % 2], "chosen": chosen, "draws": st.jitter.ctr})
    out["select_one_level"] = {"seed": 555, "cases": sel}

    # -- whole MCTS.run (mcts.py:318-512)
    runs = []

    def run_case(name, fen_i, sims, L, tt, seed, uid, net_kw, extra=None, dirichlet=True, repeats=1, model_path=False):
        mcfg = dict(BASE_MCTS, inference_batch_size=L, **(extra or {}))
        net = HashNet(**net_kw)
        mc = new_mcts(mcfg, net, model_path)
        b = chess.Board(FENS[fen_i])
        st = refshim.Streams(seed, uid)
        res = []
        with (TTOff() if tt == "off" else Nop()), refshim.injected(st):
            for r in range(repeats):
                vc, pi, rq = mc.run(b, num_simulations=sims, ply=(0 if dirichlet else 1000))
                res.append(root_dump(mc, vc, pi, rq))
        runs.append({"name": name, "fen": fen_i, "sims": sims, "L": L, "tt": tt, "seed": seed, "uid": uid, "net": net_kw,
                     "mcts_extra": extra or {}, "dirichlet": dirichlet, "repeats": repeats, "results": res, "model_path": model_path,
                     "evals": net.calls, "tt_entries": len(mc.tt),
                     "draws": {"jitter": st.jitter.ctr, "noise": st.noise.ctr, "dirichlet": st.dirichlet.ctr, "game": st.game.ctr}})
        print("run", name, "fen", fen_i, "sims", sims, "tt", tt, "evals", net.calls, "root_n", res[-1]["root_n"], flush=True)

    uid = 1000
    for tt in ("off", "on"):
        for fi in range(len(FENS)):
            for sharp, dirichlet in ((8.0, True), (0.5, False)):
                uid += 1
                run_case("basic", fi, 96, 8, tt, 1234, uid, {"seed": 3, "sharp": sharp}, dirichlet=dirichlet)
        uid += 1
        run_case("batch96", 0, 300, 96, tt, 1234, uid, {"seed": 4, "sharp": 10.0})                   # reference default batch
        uid += 1
        run_case("full_softmax", 1, 64, 4, tt, 1234, uid, {"seed": 5, "sharp": 12.0}, extra={"legal_softmax": False})
        uid += 1
        run_case("full_softmax_nonoise", 0, 40, 4, tt, 1234, uid, {"seed": 5, "sharp": 12.0},
                 extra={"legal_softmax": False, "enable_entropy_noise": False}, dirichlet=False)
        uid += 1
        run_case("c_base_1600", 1, 1600, 16, tt, 1234, uid, {"seed": 21, "sharp": 10.0},
                 extra={"cpuct_c_base": 19652.0, "cpuct_c_init": 1.25})
        uid += 1
        run_case("long_1600", 0, 1600, 32, tt, 1234, uid, {"seed": 22, "sharp": 14.0})
        uid += 1
        run_case("playout_cap", 3, 80, 8, tt, 1234, uid, {"seed": 6, "sharp": 8.0}, extra={"playout_random_frac": 0.25})
        uid += 1
        run_case("value_from_white", 3, 48, 8, tt, 1234, uid, {"seed": 7, "sharp": 8.0, "stm_oriented": False},
                 extra={"value_from_white": True})
        # MCTS._prune_children (mcts.py:806-826)
        uid += 1
        run_case("prune_topk", 1, 128, 8, tt, 1234, uid, {"seed": 31, "sharp": 8.0}, extra={"max_children": 6})
        uid += 1
        run_case("prune_minprior", 3, 128, 8, tt, 1234, uid, {"seed": 32, "sharp": 10.0}, extra={"min_child_prior": 0.03})
        uid += 1
        run_case("prune_both", 0, 200, 16, tt, 1234, uid, {"seed": 33, "sharp": 6.0}, extra={"max_children": 9, "min_child_prior": 0.01})
        # the in-process-model branch: Node._expand_with_legal_priors for every non-root expansion (mcts.py:227-256, 697-703)
        for fi, sharp in ((0, 8.0), (1, 3.0), (7, 14.0)):
            uid += 1
            run_case("raw_legal_priors", fi, 96, 8, tt, 1234, uid, {"seed": 41 + fi, "sharp": sharp}, model_path=True)
    # repeated runs on the same board with the table on: the root is fetched from the table, visits accumulate, the
    # Dirichlet noise is applied to already-noised priors (mcts.py:342-376; tests/test_integration.py:192-236)
    uid += 1
    run_case("same_board_x3", 1, 64, 8, "on", 1234, uid, {"seed": 3, "sharp": 8.0}, repeats=3)
    out["runs"] = runs

    # -- the table across the moves of one game (the finding in the module docstring)
    mc = new_mcts(dict(BASE_MCTS, inference_batch_size=8), HashNet(seed=3, sharp=8.0))
    b = chess.Board()
    trace = []
    with refshim.injected(refshim.Streams(1, 0)):
        for ply in range(3):
            try:
                vc, pi, rq = mc.run(b, num_simulations=64, ply=ply)
                trace.append({"ply": ply, "total_visits": int(sum(vc.values())), "root_n": int(mc._last_root.n)})
                b.push(max(vc, key=vc.get))
            except RuntimeError as e:
                trace.append({"ply": ply, "raised": "RuntimeError", "message": str(e)[:120]})
                break
    out["tt_across_moves"] = trace
    print("tt_across_moves:", trace)
    dump_json("ref_mcts.json.gz", out)


# ------------------------------------------------------------------------------------------------ B2. virtual loss
def gen_mcts_vl():
    """ref_mcts_vl.json.gz: the reference's own virtual-loss lines executed (class VLOn)."""
    out = {"base_mcts": BASE_MCTS, "fens": FENS}

    # -- _select, one level, with a hand-set inflight_counts dict (mcts.py:889-890 penalty, :922-923 bookkeeping)
    sel = []
    rng = np.random.default_rng(15)
    for k in range(60):
        fi = k % len(FENS)
        b = chess.Board(FENS[fi])
        legal = list(b.legal_moves)
        parent = rmcts.Node()
        parent.n = int(rng.integers(0, 400)); parent.q = float(rng.uniform(-0.8, 0.8)); parent.w = parent.q * parent.n
        parent.expanded = True
        pri = rng.dirichlet([0.5] * len(legal))
        stats, kids = [], []
        for m, p in zip(legal, pri):
            c = rmcts.Node(prior=float(p), move=m, parent=parent)
            if rng.random() < 0.5:
                c.n = int(rng.integers(1, 60)); c.q = float(rng.uniform(-1, 1)); c.w = c.q * c.n
            parent.children[m] = c
            kids.append(c)
            stats.append([c.n, c.q, float(p)])
        vloss = [1.0, 0.3, 3.0, 0.0][k % 4]
        jit = [0.05, 0.0, 0.01][k % 3]
        # in-flight counts concentrated on the children that would otherwise win
        infl = {}
        order = np.argsort([-(st[1] if st[0] else parent.q) - 2.5 * st[2] * np.sqrt(max(1, parent.n)) / (1 + st[0]) for st in stats])
        for j in order[: int(rng.integers(0, 5))]:
            infl[kids[int(j)]] = int(rng.integers(1, 6))
        before = [int(infl.get(c, 0)) for c in kids]
        mc = new_mcts(dict(BASE_MCTS, selection_jitter=jit, fpu_reduction=[0.1, 0.3][k % 2], virtual_loss=vloss), None)
        st = refshim.Streams(556, k)
        with TTOff(), refshim.injected(st):
            node, path, _ = mc._select(b.copy(), parent, inflight_counts=infl)
        chosen = kids.index(path[1])
        sel.append({"fen": fi, "uid": k, "parent_n": parent.n, "parent_q": parent.q, "children": stats, "jitter": jit,
                    "fpu_reduction": [0.1, 0.3][k % 2], "virtual_loss": vloss, "inflight": before,
                    "inflight_after": [int(infl.get(c, 0)) for c in kids], "chosen": chosen, "draws": st.jitter.ctr})
    out["select_one_level"] = {"seed": 556, "cases": sel}
    n_diff = 0
    for c in sel:                                            # the cases must actually exercise the penalty
        n_diff += int(c["inflight"][c["chosen"]] == 0 and any(c["inflight"]))
    print("select_one_level: chosen child had no in-flight count while others had in", n_diff, "of", len(sel), "cases")

    # -- whole MCTS.run with the batch dict
    runs = []

    def run_case(name, fen_i, sims, L, tt, seed, uid, net_kw, extra=None, dirichlet=True, repeats=1):
        mcfg = dict(BASE_MCTS, inference_batch_size=L, **(extra or {}))
        net = HashNet(**net_kw)
        mc = new_mcts(mcfg, net)
        b = chess.Board(FENS[fen_i])
        st = refshim.Streams(seed, uid)
        res = []
        with Both(TTOff() if tt == "off" else Nop(), VLOn()), refshim.injected(st):
            for r in range(repeats):
                vc, pi, rq = mc.run(b, num_simulations=sims, ply=(0 if dirichlet else 1000))
                res.append(root_dump(mc, vc, pi, rq))
        # the same search WITHOUT the dict, for the record: virtual loss must change the visit distribution
        mc0 = new_mcts(mcfg, HashNet(**net_kw))
        with (TTOff() if tt == "off" else Nop()), refshim.injected(refshim.Streams(seed, uid)):
            vc0, _, _ = mc0.run(chess.Board(FENS[fen_i]), num_simulations=sims, ply=(0 if dirichlet else 1000))
        differs = [int(v) for v in vc0.values()] != res[0]["n"]
        runs.append({"name": name, "fen": fen_i, "sims": sims, "L": L, "tt": tt, "seed": seed, "uid": uid, "net": net_kw,
                     "mcts_extra": extra or {}, "dirichlet": dirichlet, "repeats": repeats, "results": res, "model_path": False,
                     "evals": net.calls, "tt_entries": len(mc.tt), "differs_from_vl_off": differs,
                     "draws": {"jitter": st.jitter.ctr, "noise": st.noise.ctr, "dirichlet": st.dirichlet.ctr, "game": st.game.ctr}})
        print("vl run", name, "fen", fen_i, "sims", sims, "L", L, "tt", tt, "evals", net.calls, "root_n", res[-1]["root_n"],
              "differs from vl-off:", differs, flush=True)

    uid = 3000
    for tt in ("off", "on"):
        for fi in range(len(FENS)):
            uid += 1
            run_case("basic16", fi, 96, 16, tt, 1234, uid, {"seed": 3, "sharp": 8.0})
        for fi in (0, 1, 3, 7):
            uid += 1
            run_case("batch96", fi, 96 * 3, 96, tt, 1234, uid, {"seed": 4, "sharp": 10.0})
        uid += 1
        run_case("long_1600_L96", 0, 1600, 96, tt, 1234, uid, {"seed": 22, "sharp": 14.0})
        uid += 1
        run_case("long_1600_L16", 1, 1600, 16, tt, 1234, uid, {"seed": 21, "sharp": 10.0},
                 extra={"cpuct_c_base": 19652.0, "cpuct_c_init": 1.25})
        uid += 1
        run_case("bench_800_L96", 3, 800, 96, tt, 1234, uid, {"seed": 23, "sharp": 6.0})       # the bench's search shape
        uid += 1
        run_case("vloss_0.3", 1, 200, 32, tt, 1234, uid, {"seed": 5, "sharp": 12.0}, extra={"virtual_loss": 0.3})
        uid += 1
        run_case("vloss_3_nodir", 7, 200, 32, tt, 1234, uid, {"seed": 6, "sharp": 4.0}, extra={"virtual_loss": 3.0}, dirichlet=False)
        uid += 1
        run_case("vloss_zero", 0, 96, 16, tt, 1234, uid, {"seed": 7, "sharp": 8.0}, extra={"virtual_loss": 0.0})
        uid += 1
        run_case("mate_in_reach", 5, 128, 32, tt, 1234, uid, {"seed": 8, "sharp": 2.0})            # terminal leaves keep their counts
        uid += 1
        run_case("prune_topk", 1, 128, 16, tt, 1234, uid, {"seed": 31, "sharp": 8.0}, extra={"max_children": 6})
    out["runs"] = runs
    dump_json("ref_mcts_vl.json.gz", out)


if __name__ == "__main__":
    what = sys.argv[1:] or ["encoding", "mcts", "mcts_vl", "selfplay", "arena", "worker"]
    os.makedirs(OUT, exist_ok=True)
    if "encoding" in what:
        gen_encoding()
    if "mcts" in what:
        gen_mcts()
    if "mcts_vl" in what:
        gen_mcts_vl()
    if "selfplay" in what:
        from gen_golden_selfplay import gen_selfplay
        gen_selfplay()
    if "arena" in what:
        from gen_golden_selfplay import gen_arena
        gen_arena()
    if "worker" in what:
        from gen_golden_selfplay import gen_worker
        gen_worker()
