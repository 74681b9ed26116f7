#!/usr/bin/env python3
"""Time the HIP network forward (resident synthetic input) at several batch sizes."""
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from matrix0_amd.weights import random_state_dict
from matrix0_amd.backend import M0Backend

cfg = dict(planes=19, channels=320, blocks=24, attention_heads=20, policy_size=4672, norm="group",
           activation="silu", preact=True, policy_factor_rank=128, self_supervised=True,
           ssl_tasks=["piece", "threat", "pin", "fork", "control"])
sd = random_state_dict(cfg, seed=0, varied=True)
be = M0Backend.from_state_dict(cfg, sd)
fl = be.flops_per_position(False)
for B in [int(a) for a in sys.argv[1:]] or [256, 1024, 4096]:
    ms = be.bench_forward(B, int(os.environ.get("ITERS", "3")))
    print(json.dumps({"B": B, "ms": round(ms, 3), "pos_per_s": round(B / ms * 1e3), "TFLOPs": round(B * fl / ms / 1e9, 1)}), flush=True)
