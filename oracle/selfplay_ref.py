"""ORACLE (test infrastructure, not product): CPU restatement of ONE game of the reference's self-play worker
(azchess/selfplay/internal.py:326-660) on top of oracle/mcts_ref.py, with the injected counter streams.

Pinned by tests/golden/ref_worker_*.npz, which the real selfplay_worker produced (tools/gen_golden_selfplay.py):
tests/test_golden_selfplay.py replays every golden game through `play_game` and requires identical moves, visit counts,
policy targets, value targets, masks and metadata.  Also the CPU leg of bench.py (`cpu_baseline`).

Draws from the game stream, in the reference's order: random.choice(OPENING_BOOK) if a book is loaded (:65-69),
random.choice(legal) per opening ply (:366-379), random.randint for the playout cap inside MCTS.run when
playout_random_frac > 0 (mcts.py:378-387), np.random.choice in sample_move_from_counts unless T < 1e-3 (:690-735)."""
from __future__ import annotations

import time
from typing import Callable, Dict, List, Optional

import numpy as np

from . import chess_py as ch
from . import mcts_ref as ref


def mcts_config_for_worker(cfg_dict: dict, value_from_white: bool = False, **switches) -> ref.MCTSConfig:
    """internal.py:269-304: `mcts` section with the self-play overrides."""
    sp = cfg_dict.get("selfplay", {}) or {}
    m = dict(cfg_dict.get("mcts", {}) or {})
    m.setdefault("inference_batch_size", 96)
    m.update({
        "num_simulations": int(sp.get("num_simulations", m.get("num_simulations", 800))),
        "cpuct": float(sp.get("cpuct", m.get("cpuct", 2.5))),
        "dirichlet_alpha": float(sp.get("dirichlet_alpha", m.get("dirichlet_alpha", 0.3))),
        "dirichlet_frac": float(sp.get("dirichlet_frac", m.get("dirichlet_frac", 0.25))),
        "selection_jitter": float(sp.get("selection_jitter", m.get("selection_jitter", 0.01))),
        "draw_penalty": float(m.get("draw_penalty", -0.1)),
        "value_from_white": bool(value_from_white),
    })
    m.update(switches)
    return ref.MCTSConfig.from_dict(m)


def draw_cfg_of(cfg_dict: dict) -> dict:
    """config.py Config.draw(): top-level `draw` overridden by `selfplay.draw`."""
    d = dict(cfg_dict.get("draw", {}) or {})
    d.update((cfg_dict.get("selfplay", {}) or {}).get("draw", {}) or {})
    return d


def detect_value_from_white(infer_np) -> bool:
    """internal.py:203-243: the start position with White and with Black to move; side-to-move heads flip sign."""
    b1 = ch.Board()
    x1 = ch.encode_board(b1)
    x2 = x1.copy()
    x2[12] = 0.0                                    # same position, Black to move (only the stm plane changes)
    _, v = infer_np(np.stack([x1, x2], axis=0))
    v1, v2 = float(v[0]), float(v[1])
    return not (abs(v2 + v1) < abs(v2 - v1))


def play_game(cfg_dict: dict, infer_np: Callable, seed: int, game_uid: int, *, book: Optional[List[str]] = None,
              use_tt: bool = False, tree_reuse: bool = False, virtual_loss_active: bool = False, numerics: str = "reference",
              value_from_white: Optional[bool] = None, max_seconds: Optional[float] = None) -> Dict:
    sp = cfg_dict.get("selfplay", {}) or {}
    draw_cfg = draw_cfg_of(cfg_dict)
    force_vfw = bool((cfg_dict.get("mcts", {}) or {}).get("value_from_white", False))
    if value_from_white is None:
        value_from_white = force_vfw or detect_value_from_white(infer_np)
    mcfg = mcts_config_for_worker(cfg_dict, value_from_white, use_tt=use_tt, virtual_loss_active=virtual_loss_active,
                                  numerics=numerics)
    mcts = ref.MCTS(mcfg, infer_np, seed=seed, game=game_uid)
    game = ref.Stream(ref.derive_seed(seed, game_uid, ref.PURPOSE_GAME))
    t0 = time.perf_counter()

    if book:
        u = game.next()
        board = ch.Board(book[min(len(book) - 1, int(u * len(book)))])
    else:
        board = ch.Board()
    move_history: List[ch.Move] = []
    rnd_plies = int(sp.get("opening_random_plies", (cfg_dict.get("openings", {}) or {}).get("random_plies", 0)))
    for _ in range(max(0, rnd_plies)):
        if board.is_game_over():
            break
        legal = board.legal_moves
        if not legal:
            break
        u = game.next()
        mv = legal[min(len(legal) - 1, int(u * len(legal)))]
        board.push(mv)
        move_history.append(mv)

    temp_start = float(sp.get("temperature_start", 1.0))
    temp_end = float(sp.get("temperature_end", 0.1))
    temp_moves = int(sp.get("temperature_moves", 20))
    low_visit_thr = int(sp.get("low_visit_threshold", 0) or 0)
    window_k = int(sp.get("resign_window", 4))
    max_len = int(sp.get("max_game_len", 200))
    states, pis, turns, masks, search_values, sims_used = [], [], [], [], [], []
    trace = {"visits": [], "moves": [], "v": [], "sims": [], "temperature": [], "chosen": []}
    st = ref.ResignState()
    entropy_sum, entropy_count = 0.0, 0
    resigned, resigner, z = False, None, None
    timed_out = False

    while not board.is_game_over() and len(states) < max_len:
        if ref.should_adjudicate_draw(board, move_history, draw_cfg):
            break
        if max_seconds is not None and time.perf_counter() - t0 > max_seconds:
            timed_out = True
            break
        temperature = ref.temperature_for(board.fullmove_number, temp_start, temp_end, temp_moves)
        turn_sign = 1 if board.turn else -1
        sims_override = None
        if mcfg.playout_random_frac > 0 and mcfg.num_simulations > 0:
            sims_override = ref.playout_cap(mcfg.num_simulations, mcfg.playout_random_frac, game.next())
        if not tree_reuse:
            mcts._last_move = None
        visit_counts, pi, v = mcts.run(board, ply=len(states), sims_override=sims_override)
        if not visit_counts or all(c == 0 for c in visit_counts.values()):
            raise RuntimeError(f"MCTS returned invalid visit counts: {visit_counts}")
        moves = list(visit_counts.keys())
        visits = list(visit_counts.values())
        temp_eff = temperature
        if low_visit_thr > 0 and max(visits) < low_visit_thr:
            temp_eff = max(temperature, 0.8)
        draws = ref.sample_move_draws(visits, temp_eff)
        k = ref.sample_move_index(visits, temp_eff, game.next() if draws else 0.0)
        move = moves[k]
        ent = ref.policy_entropy(pi)
        entropy_sum += ent
        entropy_count += 1
        st.recent_entropies.append(ent)
        if len(st.recent_entropies) > window_k:
            st.recent_entropies.pop(0)
        states.append(ch.encode_board(board))
        pis.append(pi)
        search_values.append(v)
        turns.append(turn_sign)
        masks.append(ch.get_legal_actions(board).astype(np.uint8))
        sims_used.append(int(mcts._last_sims_run))
        trace["visits"].append(visits); trace["moves"].append(moves); trace["v"].append(float(v))
        trace["sims"].append(int(mcts._last_sims_run)); trace["temperature"].append(float(temp_eff)); trace["chosen"].append(k)
        if ref.resign_update(st, v, len(states), sp):
            resigned = True
            z = -1.0 if board.turn else 1.0
            resigner = "W" if board.turn else "B"
            break
        move_history.append(move)
        board.push(move)
        mcts.note_move_played(move)

    if z is None:
        if board.is_game_over(claim_draw=True):
            z = ref.game_result(board)
        else:
            z = float(search_values[-1]) if search_values else 0.0
    T = len(states)
    out = {"moves": T, "result": float(z), "resigned": resigned, "resigner": resigner, "draw": bool(z == 0.0),
           "avg_policy_entropy": entropy_sum / max(1, entropy_count),
           "avg_sims": (float(sum(sims_used)) / max(1, len(sims_used))) if sims_used else 0.0,
           "secs": time.perf_counter() - t0, "evals": mcts.evals, "trace": trace, "history": move_history,
           "timed_out": timed_out, "final_fen": board.fen(),
           "streams": {"jitter": mcts.jitter.ctr, "noise": mcts.noise.ctr, "dirichlet": mcts.dirichlet.ctr, "game": game.ctr}}
    if T > 0:
        out["s"] = np.array(states, dtype=np.float32)
        out["pi"] = np.array(pis, dtype=np.float32)
        out["z"] = np.array([z * t for t in turns], dtype=np.float32)
        out["legal_mask"] = np.stack(masks, axis=0).astype(np.uint8)
        out["search_values"] = np.array(search_values, dtype=np.float64)
    return out
