#!/usr/bin/env python3
"""Experiment: one self-play engine with 256 games vs two engines with 128 games each on two streams / host threads
(bench configuration otherwise): total evaluations per second."""
import sys, os, json, threading, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle import net_ref
from matrix0_amd.backend import M0Backend
from matrix0_amd import engine as eng
import bench

sd = net_ref.random_state_dict(bench.R24_320, seed=0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60

def make(games, first):
    be = M0Backend.from_state_dict(bench.R24_320, sd)
    cfg = eng.selfplay_cfg_from_dict(bench.SELFPLAY_CFG, concurrent_games=games, total_games=0, first_game_index=first,
                                     leaves_per_step=16, virtual_loss_active=True, record_games=False)
    return be, eng.SelfplayEngine(be, cfg)

def run(engines):
    for e in engines: e.step(10)
    s0 = [e.stats() for e in engines]
    t0 = time.perf_counter()
    th = [threading.Thread(target=e.step, args=(steps,)) for e in engines]
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.perf_counter() - t0
    s1 = [e.stats() for e in engines]
    ev = sum(b["evals"] - a["evals"] for a, b in zip(s0, s1))
    return ev / dt, dt / steps * 1e3

sets = {}
for k in (1, 2, 4):
    sets[k] = [make(256 // k, i * (256 // k)) for i in range(k)]
for r in range(3):
    out = {"round": r}
    for k in (1, 2, 4):
        ev, ms = run([e for _, e in sets[k]])
        out[f"{k}_engines"] = {"evals_per_s": round(ev), "ms_per_step_each": round(ms, 2)}
    print(json.dumps(out), flush=True)
