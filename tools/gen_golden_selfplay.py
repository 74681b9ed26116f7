#!/usr/bin/env python3
"""Golden vectors for the self-play decision logic and for WHOLE GAMES, produced by running the real reference
azchess/selfplay/internal.py (selfplay_worker, sample_move_from_counts, game_result), azchess/draw.py and the move choice of
azchess/arena.py in the build container (see tools/gen_golden_mcts.py for how the reference runs here and how its random draws
are re-routed to the shared counter streams).

gen_selfplay() -> tests/golden/ref_selfplay.json.gz   per-function cases
gen_worker()   -> tests/golden/ref_worker_<name>.npz  one finished game each, exactly as the reference's selfplay_worker
                  produced it: the NPZ its DataManager wrote (s, pi, z, legal_mask, meta_*, ssl_*), its queue message, and a
                  per-ply trace (visit counts, root value, temperature, chosen move, simulations).  The worker runs with the
                  evaluator of tests/hash_net.py behind the reference's InferenceClient seam, and with MCTS._tt_get patched to
                  return None: with its table on, the reference worker dies on the second move of every game (RuntimeError
                  'zero visits', recorded in ref_mcts.json.gz::tt_across_moves), so tree-only is the only mode in which the
                  reference finishes a game here.
"""
from __future__ import annotations

import glob
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
import azchess.logging_utils as rlog  # noqa: E402


def _quiet_logging(log_dir="logs", level=logging.INFO, name=None):
    lg = logging.getLogger(name)
    lg.handlers.clear()
    lg.addHandler(logging.NullHandler())
    lg.propagate = False
    return lg


rlog.setup_logging = _quiet_logging            # internal.py calls it at import time and would create ./logs
import azchess.draw as rdraw  # noqa: E402
import azchess.mcts as rmcts  # noqa: E402
import azchess.selfplay.internal as rsp  # noqa: E402

from oracle import chess_py as ch  # noqa: E402
from oracle import mcts_ref as oref  # noqa: E402
from tests.hash_net import HashNet  # noqa: E402
from gen_golden_mcts import BASE_MCTS, FENS, OUT, Both, Nop, TTOff, VLOn, dump_json, mv_code  # noqa: E402

logging.disable(logging.CRITICAL)


# ------------------------------------------------------------------------------------------------ per-function cases
def gen_selfplay():
    out = {"fens": FENS}
    rng = np.random.default_rng(17)

    # -- sample_move_from_counts (internal.py:690-735)
    cases = []
    for k in range(160):
        fi = k % len(FENS)
        b = chess.Board(FENS[fi])
        legal = list(b.legal_moves)
        if not legal:
            continue
        kind = k % 8
        if kind == 0:
            visits = [0] * len(legal)
        elif kind == 1:
            visits = [int(x) for x in rng.integers(0, 3, size=len(legal))]
        else:
            visits = [int(x) for x in (rng.dirichlet([0.3] * len(legal)) * rng.integers(20, 1600)).astype(int)]
        temp = [1.2, 1.0, 0.3, 0.8, 0.05, 1e-4, 0.0, 2.0][(k // 8) % 8]
        counts = dict(zip(legal, visits))
        st = refshim.Streams(777, k)
        with refshim.injected(st):
            mv = rsp.sample_move_from_counts(b, counts, temp)
        cases.append({"fen": fi, "uid": k, "visits": visits, "temperature": temp, "chosen": legal.index(mv), "draws": st.game.ctr})
    out["sample_move"] = {"seed": 777, "cases": cases}

    # -- game_result (internal.py:738-750) and Board.result(claim_draw=True) on final positions reached by move lists
    gr = []
    finals = [
        ("fools_mate", ch.START_FEN, ["f2f3", "e7e5", "g2g4", "d8h4"]),
        ("scholars_mate", ch.START_FEN, ["e2e4", "e7e5", "d1h5", "b8c6", "f1c4", "g8f6", "h5f7"]),
        ("stalemate", "7k/5Q2/5K2/8/8/8/8/8 b - - 0 1", []),
        ("insufficient", "8/8/8/8/8/8/8/K1k5 w - - 0 1", []),
        ("threefold", ch.START_FEN, ["g1f3", "g8f6", "f3g1", "f6g8", "g1f3", "g8f6", "f3g1", "f6g8"]),
        ("twofold_only", ch.START_FEN, ["g1f3", "g8f6", "f3g1", "f6g8"]),
        ("fifty_claim", "8/8/8/8/8/5k2/8/R3K3 w - - 99 120", ["a1a2"]),
        ("fifty_not_yet", "8/8/8/8/8/5k2/8/R3K3 w - - 97 120", ["a1a2"]),
        ("ongoing", ch.START_FEN, ["e2e4"]),
        ("mate_white_loses", "rnb1kbnr/pppp1ppp/8/4p3/6Pq/5P2/PPPPP2P/RNBQKBNR w KQkq - 1 3", []),
    ]
    for name, fen, ucis in finals:
        b = chess.Board(fen)
        for u in ucis:
            b.push(chess.Move.from_uci(u))
        gr.append({"name": name, "fen": fen, "moves": ucis, "game_result": rsp.game_result(b),
                   "game_over": bool(b.is_game_over()), "game_over_claim": bool(b.is_game_over(claim_draw=True)),
                   "result_claim": b.result(claim_draw=True)})
    out["game_result"] = gr

    # -- should_adjudicate_draw (draw.py:8-84) along random playouts, two configurations
    cfgs = [{}, {"enabled": True, "min_plies": 10, "window": 8, "min_unique": 3, "halfmove_cap": 20, "material_draw_threshold": 12},
            {"enabled": True, "stalemate_draw": False, "min_plies": 0, "window": 0, "halfmove_cap": 0, "material_draw_threshold": 0}]
    dr = []
    starts = [ch.START_FEN, FENS[2], FENS[4], FENS[5], "8/8/8/8/8/5k2/8/R3K3 w - - 90 120", "4k3/8/8/8/8/8/8/4K2R w K - 0 1",
              "8/5k2/8/8/8/8/3N4/4K3 w - - 0 1", "8/3b1k2/8/8/8/8/3B4/4K3 w - - 0 1"]
    for gi in range(24):
        b = chess.Board(starts[gi % len(starts)])
        st = oref.Stream(oref.derive_seed(31, gi, 4))
        moves, flags = [], [[] for _ in cfgs]
        for ply in range(120):
            for ci, c in enumerate(cfgs):
                flags[ci].append(bool(rdraw.should_adjudicate_draw(b, moves, c)))
            if b.is_game_over():
                break
            legal = list(b.legal_moves)
            # shuffle-prone policy: prefer reversible piece moves so repetitions and clock draws actually happen
            u = st.next()
            quiet = [m for m in legal if b.piece_at(m.from_square).piece_type != chess.PAWN and not b.is_capture(m)]
            pool = quiet if (quiet and st.next() < 0.85) else legal
            mv = pool[min(len(pool) - 1, int(u * len(pool)))]
            moves.append(mv)
            b.push(mv)
        dr.append({"start": starts[gi % len(starts)], "moves": [m.uci() for m in moves], "flags": flags})
    out["adjudicate_draw"] = {"cfgs": cfgs, "games": dr}
    print("adjudicate_draw: true flags per cfg:", [sum(sum(g["flags"][ci]) for g in dr) for ci in range(len(cfgs))])

    # -- arena move choice (arena.py:59-106): the reference's own game loop with two scripted searchers
    import azchess.arena as rarena
    from azchess.config import Config as RConfig

    class Scripted:
        """Stands in for MCTS in _arena_run_one_game: visit counts are a hash of the position."""

        def __init__(self, salt, log):
            self.salt, self.log = salt, log

        def run(self, board, ply=None):
            legal = list(board.legal_moves)
            key = int.from_bytes(board._transposition_key()[:8], "little") ^ (self.salt * 0x9E3779B97F4A7C15 & oref.MASK64)
            visits = {}
            for i, m in enumerate(legal):
                r = oref.mix64(key + i * 0xD6E8FEB86659FD93)
                visits[m] = int(r % 97) if (r >> 40) % 5 else 0
            if all(v == 0 for v in visits.values()):
                visits[legal[0]] = 1
            self.log.append({"ply": ply, "visits": list(visits.values())})
            return visits, np.zeros(4672, np.float32), 0.0

    ar = []
    for gi, (temp, temp_plies) in enumerate(((1.0, 12), (0.5, 6), (0.0, 0), (1e-4, 20), (1.3, 40))):
        log = []
        rarena._P_MCTS_A, rarena._P_MCTS_B = Scripted(1, log), Scripted(2, log)
        rarena._P_CFG = RConfig({"draw": {}})
        st = refshim.Streams(4242, gi)
        # record the chosen move by watching the board: wrap Board.push through a list the loop appends to
        with refshim.injected(st):
            score, nmoves, res = rarena._arena_run_one_game((gi, 30, temp, temp_plies, False))
        # replay to find which child was chosen at each ply
        b = chess.Board()
        chosen = []
        # the loop's own history is not returned; re-run the scripted searchers deterministically to rebuild it
        log2 = []
        A, B = Scripted(1, log2), Scripted(2, log2)
        st2 = refshim.Streams(4242, gi)
        a_is_white = gi % 2 == 0
        for ply in range(nmoves):
            eng = (A if (b.turn == chess.WHITE) == a_is_white else B)
            visits, _, _ = eng.run(b, ply=ply)
            legal = list(visits.keys())
            vis = [visits[m] for m in legal]
            if temp > 1e-3 and ply < temp_plies:
                u = st2.game.next()
                k = oref_arena_choice(vis, temp, u)
            else:
                k = int(np.argmax(np.array(vis, np.float32)))
            chosen.append(k)
            b.push(legal[k])
        assert st2.game.ctr == st.game.ctr
        # the rebuilt game must be the game the reference played: same searcher inputs in the same order
        assert [e["visits"] for e in log] [:nmoves] == [e["visits"] for e in log2], "arena replay diverged from the reference loop"
        ar.append({"uid": gi, "temp": temp, "temp_plies": temp_plies, "plies": nmoves, "result": res, "score": score,
                   "visits": [e["visits"] for e in log2], "chosen": chosen, "draws": st.game.ctr})
    out["arena_choice"] = {"seed": 4242, "games": ar}
    dump_json("ref_selfplay.json.gz", out)


def oref_arena_choice(vis, temp, u):
    """arena.py:73-86 restated only to REBUILD which child the reference picked (checked against the reference's own game
    by comparing the searcher inputs ply by ply, see the assertion in gen_selfplay)."""
    v = np.array(vis, dtype=np.float32)
    logits = np.log(v + 1e-8) / max(temp, 1e-3)
    probs = np.exp(logits - np.max(logits))
    s = probs.sum()
    if s <= 0 or not np.isfinite(s):
        return int(np.argmax(v))
    probs /= s
    cdf = np.cumsum(probs.astype(np.float64))
    cdf /= cdf[-1]
    return int(min(len(vis) - 1, np.searchsorted(cdf, u, side="right")))


# ------------------------------------------------------------------------------------------------ whole games
WORKER_CASES = {
    # name: (seed, game uid (= proc_id 0 / game 0), net kwargs, selfplay section, mcts extra, opening book FENs, model ssl)
    "lengthcap": (1234, {"seed": 11, "sharp": 8.0, "vscale": 0.6},
                  {"num_simulations": 48, "max_game_len": 24, "opening_random_plies": 6, "temperature_start": 1.2,
                   "temperature_end": 0.3, "temperature_moves": 40, "resign_threshold": -0.85, "min_resign_plies": 50},
                  {"inference_batch_size": 8, "playout_random_frac": 0.05}, None, True),
    "resign": (77, {"seed": 5, "sharp": 10.0, "vscale": 0.5, "vbias": -0.12, "stm_oriented": False},
               {"num_simulations": 32, "max_game_len": 60, "opening_random_plies": 2, "temperature_start": 1.0,
                "temperature_end": 0.1, "temperature_moves": 10, "resign_threshold": -0.02, "min_resign_plies": 6,
                "resign_window": 4, "resign_consecutive_bad": 3, "resign_min_entropy": 0.3, "resign_value_margin": 0.05},
               {"inference_batch_size": 4, "dirichlet_plies": 4}, None, False),
    "mate_kqk": (6, {"seed": 4, "sharp": 3.0, "vscale": 0.5},
                 {"num_simulations": 24, "max_game_len": 40, "opening_random_plies": 0, "temperature_start": 2.0,
                  "temperature_end": 2.0, "temperature_moves": 0, "resign_threshold": -1.0},
                 {"inference_batch_size": 8}, ["7k/5Q2/6K1/8/8/8/8/8 w - - 0 1", "8/8/8/8/8/1k6/2q5/K7 b - - 0 1"], False),
    "mate_or_draw_kqk": (5, {"seed": 8, "sharp": 6.0, "vscale": 0.5},
                         {"num_simulations": 40, "max_game_len": 80, "opening_random_plies": 0, "temperature_start": 0.6,
                          "temperature_end": 0.2, "temperature_moves": 20, "resign_threshold": -1.0},
                         {"inference_batch_size": 8}, ["8/8/8/4k3/8/8/3QK3/8 w - - 0 1"], False),
    "mate_black": (27, {"seed": 4, "sharp": 3.0, "vscale": 0.5},
                   {"num_simulations": 24, "max_game_len": 40, "opening_random_plies": 0, "temperature_start": 2.0,
                    "temperature_end": 2.0, "temperature_moves": 0, "resign_threshold": -1.0},
                   {"inference_batch_size": 8}, ["8/8/8/8/8/1k6/2q5/K7 b - - 0 1"], False),
    "repetition_krk": (9, {"seed": 2, "sharp": 2.0, "vscale": 0.3},
                       {"num_simulations": 24, "max_game_len": 120, "opening_random_plies": 0, "temperature_start": 0.0,
                        "temperature_end": 0.0, "temperature_moves": 0, "resign_threshold": -1.0},
                       {"inference_batch_size": 8, "dirichlet_frac": 0.0, "selection_jitter": 0.0},
                       ["8/8/8/4k3/8/8/3RK3/8 w - - 0 1"], False),
    "draw_heuristics": (21, {"seed": 13, "sharp": 4.0, "vscale": 0.4},
                        {"num_simulations": 24, "max_game_len": 100, "opening_random_plies": 0, "temperature_start": 1.0,
                         "temperature_end": 0.5, "temperature_moves": 30, "resign_threshold": -1.0,
                         "draw": {"enabled": True, "min_plies": 8, "window": 8, "min_unique": 4, "halfmove_cap": 16,
                                  "material_draw_threshold": 8}},
                        {"inference_batch_size": 8}, ["4k3/8/8/8/3n4/8/3N4/4K2R w - - 0 1"], False),
    # the length of a real self-play game at the reference's own leaf batch (mcts.inference_batch_size = 96, config.yaml:157)
    "lengthcap200_batch96": (4242, {"seed": 17, "sharp": 6.0, "vscale": 0.5},
                             {"num_simulations": 96, "max_game_len": 200, "opening_random_plies": 12, "temperature_start": 1.0,
                              "temperature_end": 0.1, "temperature_moves": 20, "resign_threshold": -0.98, "min_resign_plies": 24},
                             {"inference_batch_size": 96}, None, False),
    # virtual loss ON: the reference's own lines (mcts.py:889-890, 922-923) executed for a whole game (gen_golden_mcts.VLOn)
    "vl_on_batch96": (99, {"seed": 19, "sharp": 7.0, "vscale": 0.5},
                      {"num_simulations": 192, "max_game_len": 40, "opening_random_plies": 4, "temperature_start": 1.0,
                       "temperature_end": 0.3, "temperature_moves": 30, "resign_threshold": -0.98, "min_resign_plies": 24},
                      {"inference_batch_size": 96, "playout_random_frac": 0.1}, None, False),
}
WORKER_VL = {"vl_on_batch96"}


def run_worker_case(name):
    seed, net_kw, sp, mextra, book, ssl = WORKER_CASES[name]
    net = HashNet(**net_kw)
    mcts = dict(BASE_MCTS, **mextra)
    model = {"planes": 19, "channels": 8, "blocks": 1, "policy_size": 4672, "self_supervised": bool(ssl),
             "ssl_tasks": ["piece", "threat", "pin", "fork", "control"] if ssl else []}
    trace = {"visits": [], "moves": [], "v": [], "sims": [], "temperature": [], "chosen": [], "chosen_move": []}
    orig_run = rmcts.MCTS.run
    orig_sample = rsp.sample_move_from_counts

    def run_wrap(self, board, num_simulations=None, ply=None):
        vc, pi, v = orig_run(self, board, num_simulations, ply)
        trace["visits"].append([int(x) for x in vc.values()])
        trace["moves"].append([mv_code(m) for m in vc.keys()])
        trace["v"].append(float(v))
        trace["sims"].append(int(self._last_sims_run))
        return vc, pi, v

    def sample_wrap(board, counts, temperature):
        mv = orig_sample(board, counts, temperature)
        trace["temperature"].append(float(temperature))
        trace["chosen"].append(list(counts.keys()).index(mv))
        trace["chosen_move"].append(mv_code(mv))
        return mv

    tmp = tempfile.mkdtemp(prefix="m0gold_")
    cfg = {"device": "auto", "seed": seed, "data_dir": os.path.join(tmp, "data"), "model": model, "selfplay": sp, "mcts": mcts,
           "draw": {}, "openings": {}, "tablebases": {"enabled": False}, "presets": {}}
    q = queue.Queue()
    saved = (rsp.select_device, rsp.InferenceClient, rsp.OPENING_BOOK)
    rsp.select_device = lambda req="auto": "meta"           # any non-"cpu" name: the worker then takes its InferenceClient seam
    rsp.InferenceClient = lambda res: net
    rsp.OPENING_BOOK = [chess.Board(f) for f in book] if book else []
    rmcts.MCTS.run = run_wrap
    rsp.sample_move_from_counts = sample_wrap
    st = refshim.Streams(seed, 0)
    try:
        with Both(TTOff(), VLOn() if name in WORKER_VL else Nop()), refshim.injected(st):
            rsp.selfplay_worker(0, cfg, None, 1, q, {"fake": True})
    finally:
        rsp.select_device, rsp.InferenceClient, rsp.OPENING_BOOK = saved
        rmcts.MCTS.run = orig_run
        rsp.sample_move_from_counts = orig_sample
    msgs = []
    while not q.empty():
        msgs.append(q.get())
    game = [m for m in msgs if m["type"] == "game"][0]
    files = glob.glob(os.path.join(tmp, "data", "selfplay", "*.npz"))
    assert len(files) == 1 and os.path.abspath(game["file"]) == os.path.abspath(files[0])
    z = np.load(files[0])
    blob = {k: z[k] for k in z.files}
    T = int(blob["meta_moves"][0])
    assert len(trace["visits"]) == T
    meta = {"name": name, "seed": seed, "net": net_kw, "selfplay": sp, "mcts": mcts, "book": book or [], "ssl": bool(ssl),
            "npz_keys": sorted(z.files), "evals": net.calls,
            "virtual_loss_active": name in WORKER_VL,
            "message": {k: (v if not isinstance(v, float) else float(v)) for k, v in game.items() if k not in ("file", "secs", "avg_ms_per_move")},
            "draws": {"jitter": st.jitter.ctr, "noise": st.noise.ctr, "dirichlet": st.dirichlet.ctr, "game": st.game.ctr}}
    # planes are 0/1 except the two counter planes -> keep float32 as written; zlib shrinks them well
    blob["trace_visits"] = np.array([x for row in trace["visits"] for x in row], np.int32)
    blob["trace_moves"] = np.array([x for row in trace["moves"] for x in row], np.uint16)
    blob["trace_nchild"] = np.array([len(r) for r in trace["visits"]], np.int32)
    blob["trace_v"] = np.array(trace["v"], np.float64)
    blob["trace_sims"] = np.array(trace["sims"], np.int32)
    # on a resignation the last search is not followed by a move choice being PLAYED, but the choice is still made
    blob["trace_temperature"] = np.array(trace["temperature"], np.float64)
    blob["trace_chosen"] = np.array(trace["chosen"], np.int32)
    blob["trace_chosen_move"] = np.array(trace["chosen_move"], np.uint16)
    blob["meta_json"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, f"ref_worker_{name}.npz"), **blob)
    print(f"worker {name}: T={T} result={float(blob['meta_result'][0]):+.4f} resigned={int(blob['meta_resigned'][0])} "
          f"evals={net.calls} msg={meta['message']} -> {os.path.getsize(os.path.join(OUT, f'ref_worker_{name}.npz'))} bytes", flush=True)


# ------------------------------------------------------------------------------------------------ whole arena games
ARENA_CASES = [
    # (game index (even: A is White), net A, net B, sims, L, max_moves, temp, temp_plies, mcts extra, draw cfg)
    (0, {"seed": 51, "sharp": 8.0, "vscale": 0.5}, {"seed": 52, "sharp": 5.0, "vscale": 0.6}, 48, 8, 40, 1.0, 12, {}, {}),
    (1, {"seed": 51, "sharp": 8.0, "vscale": 0.5}, {"seed": 52, "sharp": 5.0, "vscale": 0.6}, 48, 8, 40, 0.0, 0, {}, {}),
    (2, {"seed": 53, "sharp": 6.0, "vscale": 0.4}, {"seed": 54, "sharp": 9.0, "vscale": 0.5}, 96, 32, 60, 0.5, 20,
     {"dirichlet_plies": 8}, {"enabled": True, "min_plies": 12, "window": 8, "min_unique": 5, "halfmove_cap": 7,
                               "material_draw_threshold": 10}),          # ends by heuristic adjudication (halfmove clock)
    (3, {"seed": 55, "sharp": 4.0, "vscale": 0.5}, {"seed": 56, "sharp": 12.0, "vscale": 0.5}, 64, 16, 90, 1.3, 90,
     {"selection_jitter": 0.0}, {}),
]


def _script_boosts(line, bonus0=45.0, step=6.0):
    """Rigged evaluators for a scripted game from the start position (the reference's arena always starts there, arena.py:63):
    the policy index of every scripted move with a bonus that falls along the line, per colour -- where an earlier scripted
    move is still legal it outranks the later ones, and a move that is no longer legal is masked by the legal softmax."""
    b = ch.Board()
    out = {True: [], False: []}
    for k, u in enumerate(line):
        m = ch.Move.from_uci(u)
        out[bool(b.turn)].append([int(ch.move_to_index(b, m)), float(bonus0 - step * (k // 2))])
        b.push(m)
    assert b.is_checkmate(), line
    return out[True], out[False]


def _decisive_cases():
    """Three games that END IN MATE inside the move cap (arena.py:112-120, the decisive branch): every search is prior-driven (few
    simulations, flat values, no Dirichlet noise -- as arena.py:365-381 sets evaluation searches up), each side's own evaluator
    is rigged along the script.  Note what running the reference shows here: `_select` scores a child by child.q, which
    `_backpropagate` keeps from the CHILD's side to move (mcts.py:866-881, 946-953), so a mating move's child has q = -1 and the
    search steers AWAY from it; with a prior of ~1 it still collects the most visits (c sqrt(N) / (1 + n) > 1 up to n ~ 13 of
    24) and is played by the most-visited rule."""
    scholar = ["e2e4", "e7e5", "d1h5", "b8c6", "f1c4", "g8f6", "h5f7"]          # 4. Qxf7#: White mates
    fool = ["f2f3", "e7e5", "g2g4", "d8h4"]                                     # 2... Qh4#: Black mates
    sw, sb = _script_boosts(scholar)
    fw, fb = _script_boosts(fool)
    return [
        # A is White and mates: 1-0, score 1
        (4, {"seed": 61, "sharp": 2.0, "vscale": 0.05, "boost_white": sw, "boost_black": sb},
            {"seed": 62, "sharp": 2.0, "vscale": 0.05, "boost_black": sb}, 24, 8, 40, 0.0, 0, {"selection_jitter": 0.0, "dirichlet_frac": 0.0}, {}),
        # B is White and mates: 1-0, score 0
        (5, {"seed": 63, "sharp": 2.0, "vscale": 0.05, "boost_black": sb},
            {"seed": 64, "sharp": 2.0, "vscale": 0.05, "boost_white": sw, "boost_black": sb}, 24, 8, 40, 0.0, 0, {"selection_jitter": 0.0, "dirichlet_frac": 0.0}, {}),
        # A is White and is mated by B: 0-1, score 0 (sampled moves: the rigged prior leaves the sampler nothing to choose)
        (6, {"seed": 65, "sharp": 2.0, "vscale": 0.05, "boost_white": fw},
            {"seed": 66, "sharp": 2.0, "vscale": 0.05, "boost_white": fw, "boost_black": fb}, 32, 16, 40, 0.3, 4, {"dirichlet_frac": 0.0}, {}),
    ]


def gen_arena():
    """ref_arena.json.gz: whole games of the reference's own _arena_run_one_game (arena.py:59-126) with REAL searches: two MCTS
    objects (one per side, kept across the moves as arena.py:157-158 does) behind two different evaluators.  tt = "off": the
    transposition table patched out as for the worker goldens (every run() starts from a fresh root: the match engine's
    default); tt = "on": the reference untouched -- each side's table lives for the whole game, roots are looked up in it and
    later searches merge into nodes earlier searches of that side left there."""
    import azchess.arena as rarena
    from azchess.config import Config as RConfig
    games = []
    for tt in ("off", "on"):
      for (gi, net_a, net_b, sims, L, max_moves, temp, temp_plies, mextra, draw) in ARENA_CASES + _decisive_cases():
          mcfg = dict(BASE_MCTS, inference_batch_size=L, num_simulations=sims, **mextra)
          na, nb = HashNet(**net_a), HashNet(**net_b)
          A = rmcts.MCTS(rmcts.MCTSConfig.from_dict(dict(mcfg)), None, device="cpu", inference_backend=na)
          B = rmcts.MCTS(rmcts.MCTSConfig.from_dict(dict(mcfg)), None, device="cpu", inference_backend=nb)
          trace = []
          orig_run = rmcts.MCTS.run

          def run_wrap(self, board, num_simulations=None, ply=None):
              vc, pi, v = orig_run(self, board, num_simulations, ply)
              kids = list(self._last_root.children.values())
              trace.append({"side": "A" if self is A else "B", "ply": ply, "fen": board.fen(),
                            "moves": [mv_code(m) for m in vc.keys()], "idx": [int(c.move_idx) for c in kids],
                            "visits": [int(x) for x in vc.values()], "root_q": float(v)})
              return vc, pi, v

          rarena._P_MCTS_A, rarena._P_MCTS_B = A, B
          rarena._P_CFG = RConfig({"draw": draw})
          st = refshim.Streams(8080, gi)
          rmcts.MCTS.run = run_wrap
          try:
              with (TTOff() if tt == "off" else Nop()), refshim.injected(st):
                  score, nmoves, res = rarena._arena_run_one_game((gi, max_moves, temp, temp_plies, False))
          finally:
              rmcts.MCTS.run = orig_run
          # which child was played at every ply: replay the choice (checked below against the positions the reference searched)
          b = chess.Board()
          st2 = refshim.Streams(8080, gi)
          chosen = []
          for ply in range(nmoves):
              t = trace[ply]
              assert t["fen"] == b.fen(), "arena replay diverged from the reference loop"
              if temp > 1e-3 and ply < temp_plies:
                  k = oref_arena_choice(t["visits"], temp, st2.game.next())
              else:
                  k = int(np.argmax(np.array(t["visits"], np.float32)))
              chosen.append(k)
              legal = list(b.legal_moves)
              assert [mv_code(m) for m in legal] == t["moves"]
              b.push(legal[k])
          assert st2.game.ctr == st.game.ctr
          assert len(trace) in (nmoves, nmoves + 0)
          games.append({"tt": tt, "uid": gi, "net_a": net_a, "net_b": net_b, "sims": sims, "L": L, "max_moves": max_moves, "temp": temp,
                        "temp_plies": temp_plies, "mcts": mcfg, "draw": draw, "plies": nmoves, "result": res, "score": score,
                        "trace": trace, "chosen": chosen, "final_fen": b.fen(), "evals_a": na.calls, "evals_b": nb.calls,
                        "draws": {"jitter": st.jitter.ctr, "noise": st.noise.ctr, "dirichlet": st.dirichlet.ctr, "game": st.game.ctr}})
          print(f"arena game {gi} tt={tt}: plies={nmoves} result={res} score_A={score} evals A/B = {na.calls}/{nb.calls}", flush=True)
          if gi >= 4:
              assert res == {4: "1-0", 5: "1-0", 6: "0-1"}[gi] and score == {4: 1.0, 5: 0.0, 6: 0.0}[gi], (gi, res, score)
    dump_json("ref_arena.json.gz", {"seed": 8080, "games": games})


def gen_worker(names=None):
    for name in (names or WORKER_CASES):
        run_worker_case(name)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    args = sys.argv[1:]
    if not args or "selfplay" in args:
        gen_selfplay()
    if not args or "arena" in args:
        gen_arena()
    if not args or "worker" in args:
        gen_worker([a for a in args if a in WORKER_CASES] or None)
