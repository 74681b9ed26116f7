#!/usr/bin/env python3
"""Go / no-go experiment: de-synchronise halves of the chip at launch level.
Two R24-320 networks on streams with complementary CU masks (hipExtStreamCreateWithCUMask via M0_NET_CU_MASK), each
running the tower for half of the batch, the second started `offset_ms` late, against one network on the whole chip
running the whole batch.  Usage: exp_cumask.py [boards=24576] [iters=6]"""
import os, sys, json, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from matrix0_amd.weights import random_state_dict
from matrix0_amd.backend import M0Backend

B = int(sys.argv[1]) if len(sys.argv) > 1 else 24576
IT = int(sys.argv[2]) if len(sys.argv) > 2 else 6
cfg = dict(planes=19, channels=320, blocks=24, attention_heads=20, policy_size=4672, norm="group",
           activation="silu", preact=True, policy_factor_rank=128, self_supervised=False)
sd = random_state_dict(cfg, seed=0, varied=True)


def make(mask):
    if mask:
        os.environ["M0_NET_CU_MASK"] = mask
    else:
        os.environ.pop("M0_NET_CU_MASK", None)
    be = M0Backend.from_state_dict(cfg, sd)
    os.environ.pop("M0_NET_CU_MASK", None)
    return be


def words(pred):
    w = [0] * 8
    for b in range(256):
        if pred(b):
            w[b // 32] |= 1 << (b % 32)
    return ",".join(f"{x:08x}" for x in w)


def pair(ma, mb, offset_ms, boards_each, iters):
    a, b = make(ma), make(mb)
    a.bench_forward(boards_each, 1); b.bench_forward(boards_each, 1)      # workspaces + clocks
    out = {}

    def run(be, key, delay):
        if delay > 0:
            time.sleep(delay / 1e3)
        out[key] = be.bench_forward(boards_each, iters)

    ta = threading.Thread(target=run, args=(a, "a", 0.0)); tb = threading.Thread(target=run, args=(b, "b", offset_ms))
    t0 = time.perf_counter()
    ta.start(); tb.start(); ta.join(); tb.join()
    wall = (time.perf_counter() - t0) * 1e3
    a.close(); b.close()
    # bench_forward runs one untimed warm-up forward before its `iters` timed ones
    return {"ms_a": round(out["a"], 2), "ms_b": round(out["b"], 2), "wall_ms_per_batch": round(wall / (iters + 1), 2)}


res = {}
full = make(None)
full.bench_forward(B, 1)
res["one_network_whole_chip"] = {"boards": B, "ms": round(full.bench_forward(B, IT), 2)}
res["one_network_whole_chip_half_batch"] = {"boards": B // 2, "ms": round(full.bench_forward(B // 2, IT), 2)}
full.close()
print(json.dumps(res), flush=True)
H = B // 2
presets = {
    "no_masks_two_streams": (None, None),
    "xcd_halves(b%8<4 | b%8>=4)": (words(lambda b: b % 8 < 4), words(lambda b: b % 8 >= 4)),
    "low128 | high128": (words(lambda b: b < 128), words(lambda b: b >= 128)),
    "even | odd bits": (words(lambda b: b % 2 == 0), words(lambda b: b % 2 == 1)),
}
for name, (ma, mb) in presets.items():
    for off in (0.0, 1.0, 30.0):
        try:
            r = pair(ma, mb, off, H, IT)
        except Exception as e:      # noqa
            r = {"error": str(e)}
        r.update({"preset": name, "offset_ms": off, "boards_each": H})
        print(json.dumps(r), flush=True)
