"""Evaluation-cache size sweep: share of leaf evaluations served by engine.eval_cache at 16 384 / 65 536 / 262 144 entries per game
(bench configuration, N searched plies of 256 games each): python tools/exp/cache_sweep.py [plies]  ->  profiles/r03_exp_eval_cache_size.log"""
import sys, os, time, json, copy
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench
from matrix0_amd.weights import random_state_dict
from matrix0_amd.backend import M0Backend
from matrix0_amd import engine as eng
be = M0Backend.from_state_dict(bench.R24_320, random_state_dict(bench.R24_320, seed=0, varied=True))
plies = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for entries in (16384, 65536, 262144):
    cfgd = copy.deepcopy(bench.SELFPLAY_CFG)
    cfgd.setdefault("engine", {})["eval_cache_entries"] = entries
    cfg = eng.selfplay_cfg_from_dict(cfgd, concurrent_games=256, total_games=256, leaves_per_step=96, virtual_loss_active=True,
                                     record_games=False, eval_cache=True)
    e = eng.SelfplayEngine(be, cfg)
    t0 = time.time()
    while e.running():
        e.step(9)
        st = e.stats()
        if st["plies"] >= plies * 256: break
    st = e.stats(); dt = time.time() - t0
    print(json.dumps({"entries": entries, "plies": int(st["plies"]), "evals": int(st["evals"]), "cached": int(st["evals_cached"]),
                      "share": st["evals_cached"] / max(1, st["evals"] + st["evals_cached"]), "seconds": dt,
                      "plies_per_s": st["plies"] / dt}), flush=True)
    e.close()
