"""Helpers to read the committed golden fixtures (data only: inputs + expected outputs)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_net_golden(name):
    z = np.load(os.path.join(GOLDEN, f"net_{name}.npz"))
    cfg = json.loads(str(z["cfg_json"]))
    sd = {}
    for k in z.files:
        if k.startswith("sd::"):
            sd[k[4:]] = torch.from_numpy(z[k])
        elif k.startswith("sdseed::"):
            seed, scale = z[k]
            g = torch.Generator().manual_seed(int(seed))
            sd[k[8:]] = (float(scale) * torch.randn(4672, 4096, generator=g)).half().float()
    ssl = {k[4:]: z[k] for k in z.files if k.startswith("ssl_")}
    return cfg, sd, z["x"], z["p"], z["v"], ssl


NET_CASES = ["gn_silu_preact", "gn_dense_leaky", "bn_relu_postact", "stride2"]

SEEDED_NET_CASES = ["r320x7_seeded"]


def load_seeded_net_golden(name):
    """Full-width fixtures (tools/gen_golden_net.py::SEEDED_CASES): cfg, seed, x and the outputs of the REFERENCE module with the
    weights `matrix0_amd.weights.random_state_dict(cfg, seed, varied=True)` loaded into it; the weights are rebuilt here."""
    from matrix0_amd.weights import random_state_dict
    z = np.load(os.path.join(GOLDEN, f"net_{name}.npz"))
    cfg = json.loads(str(z["cfg_json"]))
    sd = random_state_dict(cfg, seed=int(z["seed"]), varied=True)
    ssl = {k[4:]: z[k] for k in z.files if k.startswith("ssl_")}
    return cfg, sd, z["x"], z["p"], z["v"], ssl
