#!/usr/bin/env python3
"""Observed error of the HIP forward against the reference goldens / the fp32 oracle (GPU box): max|dlogit|, |dv|,
relative L2, SSL maxima, top-1-over-legal agreement -- the numbers the tolerances of tests/test_net_gpu.py are set from."""
import gzip, json, os, sys
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from matrix0_amd.backend import M0Backend
from matrix0_amd.weights import random_state_dict
from oracle import net_ref, chess_py as ch
from tests.golden_util import load_net_golden

out = {}
for name in ["gn_silu_preact", "gn_dense_leaky", "stride2"]:
    cfg, sd, x, p_ref, v_ref, ssl_ref = load_net_golden(name)
    be = M0Backend.from_state_dict(cfg, sd)
    p, v, ssl = be.infer_np_ssl(x) if ssl_ref else (*be.infer_np(x), {})
    out[name] = {"dlogit": float(np.abs(p - p_ref).max()), "logit_range": float(np.abs(p_ref).max()), "dv": float(np.abs(v - v_ref).max()),
                 "rel_l2": float(np.linalg.norm(p - p_ref) / np.linalg.norm(p_ref)),
                 "ssl": {t: float(np.abs(ssl[t] - r).max()) for t, r in ssl_ref.items()},
                 "ssl_range": {t: float(np.abs(r).max()) for t, r in ssl_ref.items()}}
    be.close()
R24 = dict(planes=19, channels=320, blocks=24, attention_heads=20, policy_size=4672, norm="group", activation="silu", preact=True,
           policy_factor_rank=128, self_supervised=True, ssl_tasks=["piece", "threat", "pin", "fork", "control"])
rows = json.load(gzip.open(os.path.join(ROOT, "tests/golden/tactical_legal_counts.json.gz"), "rt"))
fens = [r[0] for r in rows[::40]][:256]
boards = [ch.Board(f) for f in fens]
x = np.stack([ch.encode_board(b) for b in boards])
masks = np.stack([ch.get_legal_actions(b) for b in boards])
for varied in (True, False):
    sd = random_state_dict(R24, seed=0, varied=varied)
    be = M0Backend.from_state_dict(R24, sd)
    p, v, ssl = be.infer_np_ssl(x)
    with torch.no_grad():
        p_ref, v_ref, ssl_ref = net_ref.forward(sd, R24, torch.from_numpy(x), return_ssl=True)
    p_ref, v_ref = p_ref.numpy(), v_ref.numpy()
    a = np.where(masks, p, -1e9).argmax(1); b = np.where(masks, p_ref, -1e9).argmax(1)
    # margin between best and second-best legal logit in the reference where the arg-max differs
    flips = np.nonzero(a != b)[0]
    gaps = []
    for i in flips:
        srt = np.sort(p_ref[i][masks[i]])[::-1]
        gaps.append(float(srt[0] - srt[1]))
    out[f"r24_320_varied{int(varied)}"] = {"dlogit": float(np.abs(p - p_ref).max()), "logit_range": float(np.abs(p_ref).max()),
        "dv": float(np.abs(v - v_ref).max()), "rel_l2": float(np.linalg.norm(p - p_ref) / np.linalg.norm(p_ref)),
        "top1_legal_agree": float((a == b).mean()), "flip_gaps": gaps,
        "ssl": {t: float(np.abs(ssl[t] - ssl_ref[t].numpy()).max()) for t in ssl_ref},
        "ssl_range": {t: float(np.abs(ssl_ref[t].numpy()).max()) for t in ssl_ref}}
    be.close()
print(json.dumps(out, indent=1))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "net_margins.json"), "w"), indent=1)
