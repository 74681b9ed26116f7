#!/usr/bin/env python3
"""How many leaf evaluations of a self-play run are exact repeats (identical network input) of an earlier one?
Bench configuration (R24-320 random init, 800 sims/move, 96 leaves per pass), G games, P plies, through the external-evaluator
step so that every batch row is visible on the host.  Counts repeats within the same game and across games."""
import hashlib, json, sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import bench
from matrix0_amd.weights import random_state_dict
from matrix0_amd.backend import M0Backend
from matrix0_amd import engine as eng

G = int(sys.argv[1]) if len(sys.argv) > 1 else 16
P = int(sys.argv[2]) if len(sys.argv) > 2 else 12
be = M0Backend.from_state_dict(bench.R24_320, random_state_dict(bench.R24_320, seed=0, varied=True))
cfgd = json.loads(json.dumps(bench.SELFPLAY_CFG))
cfg = eng.selfplay_cfg_from_dict(cfgd, concurrent_games=G, total_games=0, leaves_per_step=96, virtual_loss_active=True, record_games=False)
e = eng.SelfplayEngine(None, cfg)
seen = set()
tot = dup = 0
t0 = time.time()
per_ply = []
last_plies = 0
while True:
    planes = e.ext_select()
    n = planes.shape[0]
    if n:
        b = planes.reshape(n, -1)
        # compact key: piece planes as bits + the 7 constant planes' first element
        bits = np.packbits(b[:, :12 * 64] > 0.5, axis=1)
        consts = b[:, 12 * 64::64][:, :7].astype(np.float32)
        for i in range(n):
            k = hashlib.blake2b(bits[i].tobytes() + consts[i].tobytes(), digest_size=12).digest()
            if k in seen:
                dup += 1
            else:
                seen.add(k)
        tot += n
        lg, v = be.infer_np(planes)
    else:
        lg, v = np.zeros((0, 4672), np.float32), np.zeros((0,), np.float32)
    e.ext_expand(lg, v)
    st = e.stats()
    if st["plies"] >= last_plies + G:
        last_plies = st["plies"]
        per_ply.append((int(st["plies"]), tot, dup))
        print(f"plies={int(st['plies'])} evals={tot} exact repeats={dup} ({100.0 * dup / max(1, tot):.2f} %) t={time.time() - t0:.0f}s", flush=True)
    if st["plies"] >= G * P:
        break
print(json.dumps({"games": G, "plies_per_game": P, "evals": tot, "exact_repeats": dup, "frac": dup / max(1, tot)}))
