#!/usr/bin/env python3
"""Experiment: the partial last round of every conv launch.  A self-play pass evaluates ~23 000 boards = 22.5 rounds of 256
four-board tiles; the last half round leaves half the CUs idle in every one of the ~60 big launches of a forward (2 %).  The tower is
board-local, so the batch can be cut into a part that is a whole number of rounds and a small tail evaluated by a SECOND network
instance on its own stream at the same time: the tail's workgroups fill the CUs the main launches leave idle.
Usage: exp_tail_split.py [boards=23048] [iters=8]"""
import os, sys, json, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from matrix0_amd.weights import random_state_dict
from matrix0_amd.backend import M0Backend

B = int(sys.argv[1]) if len(sys.argv) > 1 else 23048
IT = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg = dict(planes=19, channels=320, blocks=24, attention_heads=20, policy_size=4672, norm="group",
           activation="silu", preact=True, policy_factor_rank=128, self_supervised=False)
sd = random_state_dict(cfg, seed=0, varied=True)


def run(parts, iters):
    nets = [M0Backend.from_state_dict(cfg, sd) for _ in parts]
    for n, b in zip(nets, parts):
        n.bench_forward(b, 1)
    bar = threading.Barrier(len(nets) + 1)
    ms = [0.0] * len(nets)

    def go(i):
        bar.wait()
        ms[i] = nets[i].bench_forward(parts[i], iters, 2)

    th = [threading.Thread(target=go, args=(i,)) for i in range(len(nets))]
    for t in th:
        t.start()
    bar.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    wall = (time.perf_counter() - t0) * 1e3 / (iters + 1)
    for n in nets:
        n.close()
    return {"parts": parts, "ms_event_timed": [round(x, 2) for x in ms], "wall_ms_per_batch": round(wall, 2)}


main = (B // 1024) * 1024
for rep in range(2):
    print(json.dumps(run([B], IT)), flush=True)
    print(json.dumps(run([main, B - main], IT)), flush=True)
    print(json.dumps(run([main], IT)), flush=True)
