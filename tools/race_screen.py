#!/usr/bin/env python3
"""Race screen of the network kernels: the same bench-size batch (256 games x 96 leaves + 256 roots = 24 832 boards, R24-320) goes
through the forward N times; every run must reproduce the first one bit for bit (policy logits, values and, with --ssl, the five
SSL maps).  A wave that reads an LDS or register operand before its wait covers it, or a DMA that lands in a slot still being
read, shows up as a handful of boards that differ from run to run (tests/test_net_gpu.py runs 4 repeats; this tool is for a new
synchronisation structure: `python tools/race_screen.py 50`)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench
from matrix0_amd.backend import M0Backend
from matrix0_amd.weights import random_state_dict

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 20
ssl = "--ssl" in sys.argv
B = 24832
be = M0Backend.from_state_dict(bench.R24_320, random_state_dict(bench.R24_320, seed=0, varied=True))
rng = np.random.default_rng(23)
x = np.zeros((B, 19, 8, 8), np.float32)
x[:, :12] = (rng.random((B, 12, 8, 8)) < 0.08).astype(np.float32)
x[:, 12:17] = (rng.random((B, 5, 1, 1)) < 0.5).astype(np.float32)
x[:, 17:] = rng.random((B, 2, 1, 1)).astype(np.float32)
run = (lambda: be.infer_np_ssl(x)) if ssl else (lambda: be.infer_np(x))
ref = run()
bad = 0
t0 = time.time()
for i in range(n):
    out = run()
    dp = np.flatnonzero((out[0] != ref[0]).any(axis=1))
    dv = np.flatnonzero(out[1] != ref[1])
    ds = []
    if ssl:
        ds = [k for k in ref[2] if not np.array_equal(out[2][k], ref[2][k])]
    if len(dp) or len(dv) or ds:
        bad += 1
        print(f"run {i}: {len(dp)} boards differ in the logits {dp[:16].tolist()}, {len(dv)} in the value, ssl maps {ds}", flush=True)
    elif i % 10 == 9:
        print(f"run {i}: identical ({time.time() - t0:.0f} s)", flush=True)
print(f"race screen: {n} repeats of {B} boards, {bad} differed")
sys.exit(1 if bad else 0)
