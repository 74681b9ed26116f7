#!/usr/bin/env python3
"""A/B the two 3x3 conv kernels in one process: M0_CONV_PP=0 (conv_big_kernel) vs 1 (conv_pp_kernel).

Interleaved rounds (same device, same clocks), forward time at B boards + per-launch conv time from the
HIP-event profile, and the output difference between the two builds of the same network."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import net_ref
from matrix0_amd.backend import M0Backend

cfg = dict(planes=19, channels=320, blocks=24, attention_heads=20, policy_size=4672, norm="group",
           activation="silu", preact=True, policy_factor_rank=128, self_supervised=True,
           ssl_tasks=["piece", "threat", "pin", "fork", "control"])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
sd = net_ref.random_state_dict(cfg, seed=0)
nets = {}
for name, env in (("big", "0"), ("pp", "1")):
    os.environ["M0_CONV_PP"] = env
    nets[name] = M0Backend.from_state_dict(cfg, sd)
rng = np.random.default_rng(1)
x = (rng.random((64, 19, 8, 8)) < 0.1).astype(np.float32)
outs = {k: n.infer_np(x) for k, n in nets.items()}
dp = float(np.abs(outs["big"][0] - outs["pp"][0]).max())
dv = float(np.abs(outs["big"][1] - outs["pp"][1]).max())
print(json.dumps({"max_dlogit_big_vs_pp": dp, "max_dvalue": dv}), flush=True)
for r in range(rounds):
    for k, n in nets.items():
        n.profile_enable(True)
        ms = n.bench_forward(B, 3)
        cms, cfl, cl = n.profile_get(reset=True)
        n.profile_enable(False)
        print(json.dumps({"round": r, "kernel": k, "B": B, "fwd_ms": round(ms, 3),
                          "conv_us": round(cms / max(cl, 1) * 1e3, 1), "conv_TF": round(cfl / max(cms, 1e-9) / 1e9, 1),
                          "launches": cl}), flush=True)
