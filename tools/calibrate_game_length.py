#!/usr/bin/env python3
"""Play a full generation of self-play games (bench configuration: R24-320 random-init, 800 sims/move,
config.yaml self-play settings) to completion and record the game-length distribution that bench.py uses to
convert simulations/s into games/s for runs too short to finish games.  Writes profiles/game_length.json."""
import json, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
from matrix0_amd.weights import random_state_dict
from matrix0_amd.backend import M0Backend
from matrix0_amd import engine as eng
import bench

games = int(sys.argv[1]) if len(sys.argv) > 1 else 256
leaves = int(sys.argv[2]) if len(sys.argv) > 2 else 16
budget = float(sys.argv[3]) if len(sys.argv) > 3 and not sys.argv[3].startswith("--") else 1000.0
split = "halves" if "--half-split" in sys.argv else False       # engine.tail_split (the games are the same either way)
be = M0Backend.from_state_dict(bench.R24_320, random_state_dict(bench.R24_320, seed=0, varied=True))
cfg = eng.selfplay_cfg_from_dict(bench.SELFPLAY_CFG, concurrent_games=games, total_games=games, leaves_per_step=leaves,
                                 virtual_loss_active=True, record_games=True, eval_cache=True, tail_split=split)
e = eng.SelfplayEngine(be, cfg)
t0 = time.time(); recs = []; last = t0
while e.running() and time.time() - t0 < budget:
    e.step(20)
    while (r := e.poll()) is not None:
        recs.append({"moves": r["moves"], "total_plies": len(r["played"]), "result": r["result"], "resigned": r["resigned"],
                     "draw": r["draw"]})
    if time.time() - last > 30:
        st = e.stats(); last = time.time()
        print(f"t={last - t0:.0f}s finished={len(recs)} active={st['active_games']} evals={int(st['evals'])} plies={int(st['plies'])}", flush=True)
st = e.stats()
dt = time.time() - t0
moves = np.array([r["moves"] for r in recs], dtype=np.float64)
out = {"games": len(recs), "games_requested": games, "complete": len(recs) == games,
       "mean_plies_per_game": float(moves.mean()) if len(recs) else None,
       "median": float(np.median(moves)) if len(recs) else None, "min": float(moves.min()) if len(recs) else None,
       "max": float(moves.max()) if len(recs) else None,
       "draws": int(sum(r["draw"] for r in recs)), "resigned": int(sum(r["resigned"] for r in recs)),
       "decisive": int(sum(1 for r in recs if abs(r["result"]) == 1.0)),
       "evals": int(st["evals"]), "evals_cached": int(st["evals_cached"]), "sims": int(st["sims"]), "plies": int(st["plies"]), "seconds": dt,
       "evals_per_ply": float(st["evals"] / max(1, st["plies"])),
       # the generation lasts as long as its longest game: passes of the hot path per searched ply once the searches have
       # drifted apart (in step it is ceil(sims / leaves) = 9 at 800 / 96)
       "passes": int(st["steps"]), "passes_per_ply": float(st["steps"] / max(1.0, float(moves.max()))) if len(recs) else None,
       "games_per_s_end_to_end": (len(recs) / dt) if len(recs) == games else None,
       "note": "searched plies per game (NPZ rows); opening_random_plies=12 not included; R24-320 random init, 800 sims/move, "
               f"{games} concurrent games, {leaves} leaves/tree/step"}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
out["tail_split"] = split or "off"
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "game_length_halves.json" if split else "game_length.json"), "w"), indent=1)
print(json.dumps(out))
