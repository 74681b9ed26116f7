#!/usr/bin/env python3
"""A/B a fused kernel against its split path in one process: the residual-block tail (M0_FUSE_TAIL: conv_tail.h vs
conv2 + se_gate + ew_board) or the attention block (M0_FUSE_ATTN: attn_block.hip vs qkv + attn_core + proj + ew_board).

The switches are read once, when a network is created, so each path gets its own network instance (same weights): output
difference on random positions, then interleaved timing rounds at B boards."""
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import net_ref
from matrix0_amd.backend import M0Backend

cfg = dict(planes=19, channels=320, blocks=24, attention_heads=20, policy_size=4672, norm="group",
           activation="silu", preact=True, policy_factor_rank=128, self_supervised=True,
           ssl_tasks=["piece", "threat", "pin", "fork", "control"])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
VAR = sys.argv[3] if len(sys.argv) > 3 else "M0_FUSE_TAIL"      # or M0_FUSE_ATTN (attn_block_kernel vs four kernels)
sd = net_ref.random_state_dict(cfg, seed=0)
nets = {}
for name, env in (("split", "0"), ("fused", "1")):
    os.environ[VAR] = env
    nets[name] = M0Backend.from_state_dict(cfg, sd)
os.environ.pop(VAR, None)
rng = np.random.default_rng(1)
x = (rng.random((70, 19, 8, 8)) < 0.1).astype(np.float32)
outs = {}
for name in ("split", "fused"):
    outs[name] = nets[name].infer_np(x)
dp = float(np.abs(outs["split"][0] - outs["fused"][0]).max())
dv = float(np.abs(outs["split"][1] - outs["fused"][1]).max())
print(json.dumps({"max_dlogit_split_vs_fused": dp, "max_dvalue": dv,
                  "logit_scale": float(np.abs(outs["split"][0]).max())}), flush=True)
for r in range(rounds):
    for name in ("split", "fused"):
        ms = nets[name].bench_forward(B, 3)
        print(json.dumps({"round": r, "path": name, "B": B, "fwd_ms": round(ms, 3)}), flush=True)
