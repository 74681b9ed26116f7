"""State-dict layout of PolicyValueNet (azchess/model/resnet.py:285-582, key names of SURVEY App. A.3) and a
seeded random initialisation for runs without a checkpoint ("No checkpoint provided, using untrained model",
selfplay/internal.py:189-190).  Torch is used for tensor creation only (weight I/O)."""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch

SSL_OUT = {"piece": 13, "threat": 1, "pin": 1, "fork": 1, "control": 3}

_DEFAULTS = dict(planes=19, channels=160, blocks=14, se=True, se_ratio=0.25, attention=True, attention_heads=8,
                 attention_relbias=True, attention_every_k=3, chess_features=True, self_supervised=True,
                 piece_square_tables=True, policy_factor_rank=0, norm="batch", ssl_tasks=["piece"])


def param_shapes(cfg: dict) -> Dict[str, Tuple[int, ...]]:
    c = dict(_DEFAULTS); c.update(cfg)
    C, P, H = int(c["channels"]), int(c["planes"]), int(c["attention_heads"])
    out: Dict[str, Tuple[int, ...]] = {}

    def norm(p, ch):
        out[p + ".weight"] = (ch,); out[p + ".bias"] = (ch,)
        if c["norm"] != "group":
            out[p + ".running_mean"] = (ch,); out[p + ".running_var"] = (ch,)

    out["stem.0.weight"] = (C, P, 3, 3); norm("stem.1", C)
    if c["chess_features"]:
        out["chess_features.position_encoding"] = (1, C, 8, 8)
        if c["piece_square_tables"]:
            out["chess_features.pst_conv.weight"] = (C, C, 1, 1); norm("chess_features.pst_norm", C)
        out["chess_features.interaction_conv.weight"] = (C, C, 3, 3); norm("chess_features.interaction_norm", C)
    hidden = max(8, int(C * float(c["se_ratio"])))
    idx, k = 0, int(c["attention_every_k"])
    for i in range(int(c["blocks"])):
        p = f"tower.{idx}"; idx += 1
        out[p + ".conv1.weight"] = (C, C, 3, 3); norm(p + ".bn1", C)
        out[p + ".conv2.weight"] = (C, C, 3, 3); norm(p + ".bn2", C)
        if c["se"]:
            out[p + ".se_fc1.weight"] = (hidden, C); out[p + ".se_fc1.bias"] = (hidden,)
            out[p + ".se_fc2.weight"] = (C, hidden); out[p + ".se_fc2.bias"] = (C,)
        if c["attention"] and k > 0 and (i % k) == (k - 1):
            p = f"tower.{idx}"; idx += 1
            out[p + ".qkv.weight"] = (3 * C, C, 1, 1); out[p + ".proj.weight"] = (C, C, 1, 1)
            out[p + ".norm.weight"] = (C,); out[p + ".norm.bias"] = (C,)
            if c["attention_relbias"]:
                out[p + ".rel_bias"] = (1, H, 64, 64)
    out["policy_head.0.weight"] = (64, C, 1, 1); norm("policy_head.1", 64)
    r = int(c["policy_factor_rank"])
    if r > 0:
        out["policy_fc1.weight"] = (r, 4096); out["policy_fc1.bias"] = (r,)
        out["policy_fc2.weight"] = (4672, r); out["policy_fc2.bias"] = (4672,)
    else:
        out["policy_fc.weight"] = (4672, 4096); out["policy_fc.bias"] = (4672,)
    out["_policy_logit_scale_raw"] = ()
    out["value_head.0.weight"] = (128, C, 1, 1); norm("value_head.1", 128)
    out["value_head.3.weight"] = (128, 128, 1, 1); norm("value_head.4", 128)
    out["value_fc1.weight"] = (2 * C, 8192); out["value_fc1.bias"] = (2 * C,)
    out["value_fc2.weight"] = (C, 2 * C); out["value_fc2.bias"] = (C,)
    out["value_gate.0.weight"] = (C, C); out["value_gate.0.bias"] = (C,)
    out["value_fc3.weight"] = (1, C); out["value_fc3.bias"] = (1,)
    if c["self_supervised"]:
        for t in c["ssl_tasks"]:
            if t in SSL_OUT:
                h = f"ssl_heads.{t}"
                out[h + ".0.weight"] = (C // 2, C, 1, 1); norm(h + ".1", C // 2)
                out[h + ".3.weight"] = (SSL_OUT[t], C // 2, 1, 1)
    return out


def random_state_dict(cfg: dict, seed: int = 0, varied: bool = False) -> Dict[str, torch.Tensor]:
    """Fan-in scaled normal weights, logit scale 0.2 (resnet.py:476-480).  varied=False: unit norm gains, zero biases and
    relative-position bias (a freshly constructed module); varied=True: gains 1 +- 0.1, biases +- 0.05, relative-position
    bias +- 0.5, running statistics perturbed -- every term of the forward carries signal, as in a trained network (the
    synthetic weights of bench.py and of the full-size parity tests)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in param_shapes(cfg).items():
        if name == "_policy_logit_scale_raw":
            sd[name] = torch.tensor(math.log(math.expm1(0.2 - 1e-3)))
        elif varied and name.endswith("running_var"):
            sd[name] = torch.rand(shape, generator=g) * 0.5 + 0.75
        elif varied and name.endswith("running_mean"):
            sd[name] = torch.randn(shape, generator=g) * 0.1
        elif varied and len(shape) == 1 and name.endswith(".weight"):
            sd[name] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif varied and name.endswith(".bias"):
            sd[name] = 0.05 * torch.randn(shape, generator=g)
        elif varied and name.endswith("position_encoding"):
            sd[name] = 0.1 * torch.randn(shape, generator=g)
        elif varied and name.endswith("rel_bias"):
            sd[name] = 0.5 * torch.randn(shape, generator=g)
        elif name.endswith("running_var") or (len(shape) == 1 and name.endswith(".weight")):
            sd[name] = torch.ones(shape)
        elif name.endswith("running_mean") or name.endswith(".bias") or name.endswith("rel_bias"):
            sd[name] = torch.zeros(shape)
        elif name.endswith("position_encoding"):
            sd[name] = 0.1 * torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            sd[name] = torch.randn(shape, generator=g) * (1.0 / math.sqrt(max(1, fan_in)))
    return sd
