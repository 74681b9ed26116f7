"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the decision functions of the reference's evaluation arena
(azchess/arena.py, azchess/elo.py).  Imported by tests/ only; the product (matrix0_amd/arena.py, csrc/host_rules.h)
must never import it.

  arena_choose_move   arena.py:73-106   temperature sampling over log-visits / most-visited move
  game_score          arena.py:110-126  score of the game from A's point of view
  wilson_interval     arena.py:272-278
  expected_score / update_elo   elo.py:10-22
"""
from __future__ import annotations

import math
from typing import Sequence, Tuple

import numpy as np


def arena_choose_move(visits: Sequence[int], temp: float, ply: int, temp_plies: int, u: float) -> int:
    """Index into the visit list (dict insertion order = legal-move order).  `u` replaces the uniform that
    np.random.choice(len(moves), p=probs) draws (inverse CDF, as numpy implements it)."""
    vis = np.array(list(visits), dtype=np.float32)
    if temp > 1e-3 and ply < temp_plies:
        logits = np.log(vis + np.float32(1e-8)) / np.float32(max(temp, 1e-3))
        probs = np.exp(logits - np.max(logits))
        s = probs.sum()
        if s <= 0 or not np.isfinite(s):
            return int(np.argmax(vis))
        probs = probs / s
        cdf = np.cumsum(probs.astype(np.float64))
        cdf /= cdf[-1]
        return int(min(np.searchsorted(cdf, u, side="right"), len(vis) - 1))
    # max(visits.items(), key=lambda kv: kv[1])[0]: the FIRST maximum in insertion order
    best = 0
    for i in range(1, len(vis)):
        if vis[i] > vis[best]:
            best = i
    return best


def game_score(result: str, a_is_white: bool) -> float:
    if result == "1-0":
        return 1.0 if a_is_white else 0.0
    if result == "0-1":
        return 0.0 if a_is_white else 1.0
    return 0.5


def wilson_interval(p: float, n: int, z: float = 1.96) -> Tuple[float, float]:
    if n == 0:
        return 0.0, 0.0
    denom = 1 + z * z / n
    center = (p + (z * z) / (2 * n)) / denom
    half = (z * ((p * (1 - p) / n) + (z * z) / (4 * n * n)) ** 0.5) / denom
    return max(0.0, center - half), min(1.0, center + half)


def expected_score(ra: float, rb: float) -> float:
    return 1.0 / (1.0 + 10.0 ** ((rb - ra) / 400.0))


def update_elo(ra: float, rb: float, sa: float, k: float = 20.0) -> Tuple[float, float]:
    delta = k * (sa - expected_score(ra, rb))
    return ra + delta, rb - delta


def play_game(idx: int, mcts_cfg: dict, infer_a, infer_b, seed: int, *, sims: int, max_moves: int, temp: float, temp_plies: int,
              draw_cfg: dict, numerics: str = "reference", virtual_loss_active: bool = False, use_tt: bool = False):
    """One game of _arena_run_one_game (arena.py:59-126) on the oracle's MCTS: game `idx` has A as White when idx is even, each
    side searches with its OWN MCTS object (arena.py:157-158), and the two objects draw from ONE set of streams, as both of the
    reference's draw from the process-global generators.  use_tt = False: tree-only, a fresh root per run() (the reference with
    its table patched out); use_tt = True: each side's table lives for the whole game, as in the untouched reference -- both
    modes are in tests/golden/ref_arena.json.gz.  Returns the per-ply trace."""
    from . import chess_py as ch
    from . import mcts_ref as ref
    cfg = ref.MCTSConfig.from_dict(dict(mcts_cfg, num_simulations=sims, use_tt=use_tt, virtual_loss_active=virtual_loss_active,
                                        numerics=numerics))
    A = ref.MCTS(cfg, infer_a, seed=seed, game=idx)
    B = ref.MCTS(cfg, infer_b, seed=seed, game=idx)
    B.jitter, B.noise, B.dirichlet = A.jitter, A.noise, A.dirichlet
    game = ref.Stream(ref.derive_seed(seed, idx, ref.PURPOSE_GAME))
    board = ch.Board()
    a_is_white = idx % 2 == 0
    history, trace = [], []
    n = 0
    while not board.is_game_over(claim_draw=True) and n < max_moves:
        eng = A if (board.turn == ch.WHITE) == a_is_white else B
        if ref.should_adjudicate_draw(board, history, draw_cfg):
            break
        eng._last_root = None                           # tree-only oracle: no subtree carried between a side's searches
        visits, _, root_q = eng.run(board, ply=n)
        moves = list(visits.keys())
        vis = [visits[m] for m in moves]
        sampling = temp > 1e-3 and n < temp_plies
        k = arena_choose_move(vis, temp, n, temp_plies, game.next() if sampling else 0.0)
        trace.append({"side": "A" if eng is A else "B", "ply": n, "fen": board.fen(), "visits": vis, "chosen": k,
                      "root_q": root_q, "moves": [m.from_square | (m.to_square << 6) | ((m.promotion or 0) << 12) for m in moves]})
        board.push(moves[k])
        history.append(moves[k])
        n += 1
    if board.is_game_over(claim_draw=True):
        res = board.result(claim_draw=True)
    else:
        res = "1/2-1/2"
    return {"plies": n, "result": res, "score": game_score(res, a_is_white), "trace": trace, "final_fen": board.fen(),
            "evals_a": A.evals, "evals_b": B.evals,
            "draws": {"jitter": A.jitter.ctr, "noise": A.noise.ctr, "dirichlet": A.dirichlet.ctr, "game": game.ctr}}
