#!/usr/bin/env python3
"""Race screen of the tree kernels + the self-play step: the same generation (64 games, R24-320 x 3 blocks, 400 simulations per move,
96 leaves per pass, terminal positions reachable: KQK / KRK openings mixed in) is played N times; every run must reproduce the first
one bit for bit (moves, visit distributions, values).  For new synchronisation in select / expand / advance (round 4: the
LDS-resident top of the tree, written through from two places): `python tools/race_screen_search.py 8 [--tail-split | --half-split]`."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from matrix0_amd.backend import M0Backend
from matrix0_amd.weights import random_state_dict
from matrix0_amd import engine as eng

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 6
tail = "halves" if "--half-split" in sys.argv else ("--tail-split" in sys.argv)
net = dict(planes=19, channels=320, blocks=3, attention_heads=20, policy_size=4672, norm="group", activation="silu",
           preact=True, policy_factor_rank=128, self_supervised=False)
be = M0Backend.from_state_dict(net, random_state_dict(net, seed=5, varied=True))
cfg = {"seed": 4242,
       "mcts": {"cpuct": 2.5, "cpuct_start": 3.0, "cpuct_end": 2.0, "cpuct_plies": 40, "dirichlet_plies": 30, "dirichlet_frac": 0.25,
                "selection_jitter": 0.05, "fpu_reduction": 0.1, "draw_penalty": -0.05, "legal_softmax": True,
                "inference_batch_size": 96, "playout_random_frac": 0.0},
       "selfplay": {"num_simulations": 400, "max_game_len": 10, "min_resign_plies": 50, "resign_threshold": -0.85,
                    "opening_random_plies": 2, "temperature_start": 1.2, "temperature_end": 0.3, "temperature_moves": 40},
       "engine": {"opening_fens": ["8/8/8/4k3/8/8/3QK3/8 w - - 0 1", "8/8/8/4k3/8/8/3RK3/8 b - - 0 1",
                                   "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1"]}}


def play():
    e = eng.SelfplayEngine(be, eng.selfplay_cfg_from_dict(cfg, concurrent_games=64, total_games=64, eval_cache=True, tail_split=tail))
    fens = cfg["engine"]["opening_fens"]
    if hasattr(e, "set_openings"):
        e.set_openings(fens)
    games = {}
    while e.running():
        e.step(8)
        while (r := e.poll()) is not None:
            games[r["game_index"]] = r
    st = e.stats()
    e.close()
    return games, st


ref, st0 = play()
bad = 0
t0 = time.time()
for i in range(n):
    out, st = play()
    diff = [g for g in ref if out[g]["played"] != ref[g]["played"] or any(not np.array_equal(out[g][k], ref[g][k]) for k in ("pi", "z", "search_values"))]
    if diff or st["evals"] != st0["evals"]:
        bad += 1
        print(f"run {i}: {len(diff)} games differ {diff[:8]}, evals {int(st['evals'])} vs {int(st0['evals'])}", flush=True)
    else:
        print(f"run {i}: identical ({time.time() - t0:.0f} s; {int(st['evals'])} evals, {int(st['evals_cached'])} cached, tail rows {int(st['rows_tail'])})", flush=True)
print(f"search race screen: {n} repeats of 64 games, {bad} differed")
sys.exit(1 if bad else 0)
