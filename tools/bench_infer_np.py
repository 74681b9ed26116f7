#!/usr/bin/env python3
"""The `infer_np` seam with HOST buffers (the reference's mcts.py:618-621 call: numpy planes in, numpy logits + values out), against
the same forward with device-resident input: the PCIe-inclusive rate of the boundary that hands over host memory (DESIGN.md
section 6; never the bench's `value`, whose batches are encoded on the device).  `python tools/bench_infer_np.py [B ...]`"""
import sys, os, json, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from matrix0_amd.weights import random_state_dict
from matrix0_amd.backend import M0Backend

cfg = dict(planes=19, channels=320, blocks=24, attention_heads=20, policy_size=4672, norm="group",
           activation="silu", preact=True, policy_factor_rank=128, self_supervised=False)
be = M0Backend.from_state_dict(cfg, random_state_dict(cfg, seed=0, varied=True))
rng = np.random.default_rng(1)
for B in [int(a) for a in sys.argv[1:]] or [96, 4096, 24576]:
    x = np.zeros((B, 19, 8, 8), np.float32)
    x[:, :12] = (rng.random((B, 12, 8, 8)) < 0.08).astype(np.float32)
    x[:, 12:17] = (rng.random((B, 5, 1, 1)) < 0.5).astype(np.float32)
    x[:, 17:] = rng.random((B, 2, 1, 1)).astype(np.float32)
    be.infer_np(x)                                           # workspace + first-touch of the output arrays
    n = 5 if B >= 4096 else 50
    t0 = time.perf_counter()
    for _ in range(n):
        p, v = be.infer_np(x)
    dt = (time.perf_counter() - t0) / n
    ms_dev = be.bench_forward(B, 3)
    mb = (x.nbytes + p.nbytes + v.nbytes) / 1e6
    print(json.dumps({"B": B, "infer_np_ms": round(dt * 1e3, 2), "evals_per_s_host_buffers": round(B / dt),
                      "forward_ms_device_resident": round(ms_dev, 2), "evals_per_s_device_resident": round(B / ms_dev * 1e3),
                      "host_MB_per_call": round(mb, 1), "share_of_call_outside_the_forward": round(1 - ms_dev / (dt * 1e3), 3)}), flush=True)
