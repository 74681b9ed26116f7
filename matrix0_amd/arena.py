"""Evaluation matches between two networks on one MI355X: the device-resident counterpart of
azchess/arena.py.  `play_match` keeps the reference's call shape and return value (the summed score of A);
all games of the match are played concurrently by one engine (m0_arena_create), each search evaluated by the network
of the side to move.

Semantics kept from the reference (file:line = azchess/arena.py): game i gives A the white pieces when i is even (:66);
a game runs while `not board.is_game_over(claim_draw=True)` and fewer than `max_moves` plies were played, with the draw
adjudication test of draw.py before every search (:67-72); the move is sampled from softmax(log(visits+1e-8)/temp) for
the first `temp_plies` plies when temp > 1e-3, otherwise the most visited move (:73-106); unfinished games score 1/2
(:110-124); evaluation searches use no Dirichlet noise unless `eval.dirichlet_frac` says so and no entropy noise
(:365-381); Wilson interval (:272-278), Elo update (elo.py:10-22), PGN files game_NNNN.pgn (:281-303).
Search structure: by default every move starts a fresh tree.  The reference keeps one MCTS object -- hence one transposition
table -- per side for the whole game (:157-158): `engine.compat.tt_merge: true` gives the match engine exactly that (one
device table per side and game, roots looked up in it, later searches merging into the nodes earlier ones left there; size
`engine.arena_nodes` for all nodes a side creates in a game).  Both modes replay the reference's own arena games move for move
(tests/golden/ref_arena.json.gz, tests/test_golden_selfplay_gpu.py)."""
from __future__ import annotations

import json
import os
import time
from datetime import datetime
from typing import Dict, List, Optional, Tuple

from . import engine as eng

K_DEFAULT = 20.0


def wilson_interval(p: float, n: int, z: float = 1.96) -> Tuple[float, float]:
    if n == 0:
        return 0.0, 0.0
    denom = 1 + z * z / n
    center = (p + (z * z) / (2 * n)) / denom
    half = (z * ((p * (1 - p) / n) + (z * z) / (4 * n * n)) ** 0.5) / denom
    return max(0.0, center - half), min(1.0, center + half)


def expected_score(ra: float, rb: float) -> float:
    return 1.0 / (1.0 + 10.0 ** ((rb - ra) / 400.0))


def update_elo(ra: float, rb: float, sa: float, k: float = K_DEFAULT) -> Tuple[float, float]:
    delta = k * (sa - expected_score(ra, rb))
    return ra + delta, rb - delta


def result_string(z_white: float, finished: bool) -> str:
    if not finished or z_white == 0.0:
        return "1/2-1/2"
    return "1-0" if z_white > 0 else "0-1"


def game_score(result: str, a_is_white: bool) -> float:
    if result == "1-0":
        return 1.0 if a_is_white else 0.0
    if result == "0-1":
        return 0.0 if a_is_white else 1.0
    return 0.5


def save_pgn(moves_raw, result: str, headers: Dict[str, str], out_dir: str, idx: int) -> str:
    """game_{idx:04d}.pgn with the seven-tag roster + caller headers and SAN movetext (lines wrapped at 80 columns)."""
    os.makedirs(out_dir, exist_ok=True)
    tags = {"Event": "?", "Site": "?", "Date": "????.??.??", "Round": "?", "White": "?", "Black": "?", "Result": result}
    tags.update({k: str(v) for k, v in headers.items()})
    tags["Result"] = result
    words = (eng.san_game(moves_raw).split() if len(moves_raw) else []) + [result]
    lines, cur = [], ""
    for w in words:
        if cur and len(cur) + 1 + len(w) > 80:
            lines.append(cur)
            cur = w
        else:
            cur = w if not cur else cur + " " + w
    if cur:
        lines.append(cur)
    path = os.path.join(out_dir, f"game_{idx:04d}.pgn")
    with open(path, "w") as f:
        for k, v in tags.items():
            f.write(f'[{k} "{v}"]\n')
        f.write("\n" + "\n".join(lines) + "\n")
    return path


def arena_cfg_from_dict(cfg: dict, *, games: int, num_sims: int, max_moves: int, temp: float, temp_plies: int,
                        concurrent_games: int, leaves_per_step: Optional[int], seed: Optional[int]) -> eng.SelfplayCfg:
    """MCTS settings as play_match builds them (arena.py:361-381): config.yaml's mcts section with num_simulations,
    eval.dirichlet_frac (default 0), selfplay.selection_jitter, entropy noise off."""
    m = dict(cfg.get("mcts", {}) or {})
    sp = dict(cfg.get("selfplay", {}) or {})
    ev = dict(cfg.get("eval", {}) or {})
    m.update({"dirichlet_frac": float(ev.get("dirichlet_frac", 0.0)), "selection_jitter": float(sp.get("selection_jitter", 0.0)),
              "enable_entropy_noise": False})
    draw = dict(cfg.get("draw", {}) or {})
    ecfg = dict(cfg.get("engine", {}) or {})
    cfg2 = {"seed": cfg.get("seed", 1234), "mcts": m, "draw": draw, "engine": {"compat": dict(ecfg.get("compat", {}) or {})},
            "selfplay": {"num_simulations": int(num_sims), "max_game_len": int(max_moves), "opening_random_plies": 0,
                         "resign_threshold": -2.0, "min_resign_plies": 10 ** 9}}
    c = eng.selfplay_cfg_from_dict(cfg2, concurrent_games=concurrent_games, total_games=games, seed=seed,
                                   leaves_per_step=leaves_per_step, virtual_loss_active=bool(ecfg.get("virtual_loss_active", True)),
                                   record_games=False, arena_nodes=int(ecfg.get("arena_nodes", 0) or 0))
    c.arena_temp = float(temp)
    c.arena_temp_plies = int(temp_plies)
    return c


last_match_stats: Dict[str, float] = {}


def play_match(backend_a, backend_b, games: int, cfg: dict, seed: Optional[int] = None, pgn_out: Optional[str] = None,
               pgn_sample: int = 0, num_sims: int = 500, max_moves_override: Optional[int] = None, temp: float = 0.0,
               temp_plies: int = 0, concurrent_games: Optional[int] = None, leaves_per_step: Optional[int] = 16,
               elo_book: Optional[str] = None, log_dir: Optional[str] = None, progress=None) -> float:
    """A vs B over `games` games; returns the summed score of A (1 / 0.5 / 0 per game), as arena.play_match does.
    `backend_a` / `backend_b` are M0Backend objects on the same device (the reference takes checkpoint paths:
    M0Backend.from_checkpoint).  Details of the last match (W/L/D, win rate, Wilson interval, Elo) are left in
    `last_match_stats`."""
    global last_match_stats
    if games <= 0:
        return 0.0
    ev = dict(cfg.get("eval", {}) or {})
    max_moves = int(max_moves_override) if max_moves_override is not None else int(ev.get("max_moves", 300))
    conc = int(concurrent_games or min(games, 256))
    ecfg = dict(cfg.get("engine", {}) or {})
    if bool((ecfg.get("compat", {}) or {}).get("tt_merge", False)) and not int(ecfg.get("arena_nodes", 0) or 0):
        # per-side tables for the whole game: nothing is compacted, so a side's arena half must hold every node it creates in the
        # game (~40 per simulation and search).  Size it, and keep the resident games within ~96 GB of node storage.
        nodes = int(min(8_000_000, max(65536, num_sims * 40 * (max_moves // 2 + 2))))
        tcap = 1024
        while tcap < 2 * nodes:                                        # selfplay_create_impl rounds a table up to a power of two
            tcap <<= 1
        per_game = 2 * nodes * 46 + 2 * tcap * 12                      # two halves of SoA nodes + two tables (8-byte key + 4-byte node)
        conc = max(1, min(conc, int(96e9 // per_game)))
        cfg = dict(cfg, engine=dict(ecfg, arena_nodes=nodes))
    c = arena_cfg_from_dict(cfg, games=games, num_sims=num_sims, max_moves=max_moves, temp=temp, temp_plies=temp_plies,
                            concurrent_games=min(conc, games), leaves_per_step=leaves_per_step, seed=seed)
    e = eng.ArenaEngine(backend_a, backend_b, c)
    score = 0.0
    a_wins = b_wins = draws = 0
    done: List[dict] = []
    t0 = time.perf_counter()
    try:
        while e.running():
            e.step(8)
            while (r := e.poll()) is not None:
                a_white = (r["game_index"] % 2 == 0)
                # result from White's point of view; 0 also for games cut at max_moves / adjudicated (scored 1/2)
                res = result_string(float(r["result"]), True)
                sc = game_score(res, a_white)
                score += sc
                if sc == 1.0:
                    a_wins += 1
                elif sc == 0.0:
                    b_wins += 1
                else:
                    draws += 1
                r["result_str"] = res
                r["score_a"] = sc
                done.append(r)
                if pgn_out and (pgn_sample <= 0 or len(done) <= pgn_sample):
                    save_pgn(r["played_raw"], res, {"Event": "Matrix0 arena", "Round": r["game_index"] + 1,
                                                    "White": "A" if a_white else "B", "Black": "B" if a_white else "A",
                                                    "Date": datetime.now().strftime("%Y.%m.%d")}, pgn_out, r["game_index"])
                if progress is not None:
                    progress(len(done), games, a_wins, b_wins, draws)
    finally:
        st = e.stats()
        e.close()
    win_rate = score / float(max(1, games))
    lo, hi = wilson_interval(win_rate, games)
    stats = {"games": games, "a_wins": a_wins, "b_wins": b_wins, "draws": draws, "score": score, "win_rate": win_rate,
             "wilson_low": lo, "wilson_high": hi, "seconds": time.perf_counter() - t0, "evals": float(st["evals"]),
             "plies": float(st["plies"]), "records": done}
    if elo_book:
        book = {"best": 1500.0, "enhanced_best": 1500.0, "candidate": 1500.0, "baseline": 1500.0, "history": []}
        if os.path.exists(elo_book):
            try:
                book = json.loads(open(elo_book).read())
            except Exception:
                pass
        ra, rb = update_elo(float(book.get("candidate", 1500.0)), float(book.get("best", 1500.0)), win_rate)
        book["candidate"], book["best"] = ra, rb
        book.setdefault("history", []).append({"win_rate": win_rate, "games": games, "timestamp": int(time.time())})
        os.makedirs(os.path.dirname(elo_book) or ".", exist_ok=True)
        open(elo_book, "w").write(json.dumps(book, indent=2))
        stats["elo_candidate"], stats["elo_best"] = ra, rb
    if log_dir:
        os.makedirs(log_dir, exist_ok=True)
        with open(os.path.join(log_dir, "eval_summary.jsonl"), "a") as f:
            f.write(json.dumps({"type": "eval_summary", "games": int(games), "a_wins": a_wins, "b_wins": b_wins,
                                "draws": draws, "win_rate": float(win_rate), "timestamp": int(time.time())}) + "\n")
    last_match_stats = stats
    return score
