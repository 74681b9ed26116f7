#!/usr/bin/env python3
"""Throughput of the evaluation-match engine at the benchmark size: two R24-320 networks, 256 concurrent games,
800 simulations per move, games cut after a few plies (the rate does not depend on game length)."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import net_ref
from matrix0_amd.backend import M0Backend
from matrix0_amd import arena
import bench

plies = int(sys.argv[1]) if len(sys.argv) > 1 else 4
a = M0Backend.from_state_dict(bench.R24_320, net_ref.random_state_dict(bench.R24_320, seed=0))
b = M0Backend.from_state_dict(bench.R24_320, net_ref.random_state_dict(bench.R24_320, seed=1))
cfg = dict(bench.SELFPLAY_CFG, eval={"max_moves": plies})
t0 = time.perf_counter()
score = arena.play_match(a, b, 256, cfg, seed=1, num_sims=800, temp=1.0, temp_plies=30, concurrent_games=256, leaves_per_step=16)
dt = time.perf_counter() - t0
st = arena.last_match_stats
print(json.dumps({"games": 256, "plies_per_game": plies, "seconds": round(dt, 2), "evals": st["evals"], "evals_per_s": round(st["evals"] / dt),
                  "searched_plies_per_s": round(st["plies"] / dt, 1), "score_a": score,
                  "games_per_s_at_100_plies": round(st["plies"] / dt / 100.0, 3)}))
