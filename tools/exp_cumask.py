#!/usr/bin/env python3
"""Go / no-go experiment: de-synchronise halves of the chip at launch level.
Two R24-320 networks on their own streams -- without CU masks, and with complementary CU masks (hipExtStreamCreateWithCUMask via
M0_NET_CU_MASK: bits 0-127 = 16 CUs of every XCD, bits 128-255 = the other 16; tools/ubench/cumask_probe.hip) -- each running the
tower for half of the batch, the second started `offset_ms` late, against one network on the whole chip running the whole batch.
Timing: every thread first makes its input and workspaces (untimed), then all meet at a barrier; wall = barrier -> both done, over
N + 1 forwards per network (bench_forward's own warm-up forward included on both sides of the comparison).
Usage: exp_cumask.py [boards=24576] [iters=8]"""
import os, sys, json, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from matrix0_amd.weights import random_state_dict
from matrix0_amd.backend import M0Backend

B = int(sys.argv[1]) if len(sys.argv) > 1 else 24576
IT = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg = dict(planes=19, channels=320, blocks=24, attention_heads=20, policy_size=4672, norm="group",
           activation="silu", preact=True, policy_factor_rank=128, self_supervised=False)
sd = random_state_dict(cfg, seed=0, varied=True)
LO = "ffffffff,ffffffff,ffffffff,ffffffff,0,0,0,0"
HI = "0,0,0,0,ffffffff,ffffffff,ffffffff,ffffffff"


def make(mask):
    if mask:
        os.environ["M0_NET_CU_MASK"] = mask
    be = M0Backend.from_state_dict(cfg, sd)
    os.environ.pop("M0_NET_CU_MASK", None)
    return be


def side_by_side(masks, boards_each, offset_ms, iters):
    nets = [make(m) for m in masks]
    for n in nets:
        n.bench_forward(boards_each, 1)                      # input + workspaces
    bar = threading.Barrier(len(nets) + 1)
    ms = [0.0] * len(nets)

    def run(i):
        bar.wait()
        if i and offset_ms > 0:
            time.sleep(offset_ms / 1e3)
        ms[i] = nets[i].bench_forward(boards_each, iters, 2)    # flag 2: keep the resident input

    th = [threading.Thread(target=run, args=(i,)) for i in range(len(nets))]
    for t in th:
        t.start()
    bar.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    wall = (time.perf_counter() - t0) * 1e3
    for n in nets:
        n.close()
    total = boards_each * len(nets)
    return {"streams": len(nets), "masks": ["-" if m is None else ("lo" if m == LO else "hi") for m in masks], "offset_ms": offset_ms,
            "boards_each": boards_each, "ms_per_forward_event_timed": [round(x, 2) for x in ms],
            "wall_ms_per_%d_boards" % total: round(wall / (iters + 1), 2)}


for rep in range(2):
    print(json.dumps(side_by_side([None], B, 0.0, IT)), flush=True)
    print(json.dumps(side_by_side([None, None], B // 2, 0.0, IT)), flush=True)
    print(json.dumps(side_by_side([LO, HI], B // 2, 0.0, IT)), flush=True)
    print(json.dumps(side_by_side([LO, HI], B // 2, 1.0, IT)), flush=True)
    print(json.dumps(side_by_side([None, None], B // 2, 1.0, IT)), flush=True)
    print(json.dumps(side_by_side([None, None], B // 2, 33.0, IT)), flush=True)
