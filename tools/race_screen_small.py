#!/usr/bin/env python3
"""Run-to-run determinism of small self-play generations (tests/test_selfplay_gpu.py::test_selfplay_is_deterministic_for_a_seed,
repeated): G games of the 32-channel test network, N repeats in one process; reports the repeats that differ from the first and
where (game, first differing ply).  `python tools/race_screen_small.py 40 [G]`"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import test_selfplay_gpu as T
from oracle import net_ref
from matrix0_amd.backend import M0Backend
from matrix0_amd import engine as eng

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
G = int(sys.argv[2]) if len(sys.argv) > 2 else 3
be = M0Backend.from_state_dict(T.NET, net_ref.random_state_dict(T.NET, seed=1))


def play():
    e = eng.SelfplayEngine(be, eng.selfplay_cfg_from_dict(T.CFG, concurrent_games=G, total_games=G))
    games = {}
    while e.running():
        e.step(16)
        while (r := e.poll()) is not None:
            games[r["game_index"]] = r
    st = e.stats()
    e.close()
    return games, st


ref, st0 = play()
bad = 0
prev = ref
for i in range(n):
    out, st = play()
    same_as_prev = all(np.array_equal(prev[g]["search_values"], out[g]["search_values"]) for g in ref)
    dv = max(float(np.abs(ref[g]["search_values"] - out[g]["search_values"]).max()) for g in ref)
    print(f"  run {i}: values same as previous run: {same_as_prev}; max |dv| vs first {dv:.3g}", flush=True)
    prev = out
    diffs = []
    for g in sorted(ref):
        a, b = ref[g]["played"], out[g]["played"]
        if a != b:
            k = next((j for j in range(min(len(a), len(b))) if a[j] != b[j]), min(len(a), len(b)))
            diffs.append((g, "move", k, len(a), len(b)))
        elif not np.array_equal(ref[g]["pi"], out[g]["pi"]):
            t = int(np.argmax(np.any(ref[g]["pi"] != out[g]["pi"], axis=1)))
            diffs.append((g, "pi", t, float(np.abs(ref[g]["pi"][t] - out[g]["pi"][t]).max())))
        elif not np.array_equal(ref[g]["search_values"], out[g]["search_values"]):
            diffs.append((g, "values"))
    if diffs or st["evals"] != st0["evals"]:
        bad += 1
        print(f"run {i}: {diffs} evals {int(st['evals'])} vs {int(st0['evals'])}", flush=True)
print(f"small-net race screen: {n} repeats of {G} games ({int(st0['evals'])} evals, {int(st0['plies'])} plies), {bad} differed")
sys.exit(1 if bad else 0)
