"""ORACLE (test infrastructure, not product): fp32 CPU restatement of the
reference network forward, PolicyValueNet.forward (azchess/model/resnet.py:656-760).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this.  It is a *functional* restatement: it takes a plain ``state_dict`` (name ->
tensor, reference key names) and a config dict and replays the
op sequence of the reference in torch fp32 on CPU.

Pinning: tests/golden/net_*.npz hold (state_dict, input, p, v, ssl) produced by
the *real* reference module imported in the build container by
tools/gen_golden_net.py; tests/test_oracle_net.py checks this file against them
(max|dlogit| <= 1e-4, |dv| <= 1e-5).

Reference map (file:line, all in azchess/model/resnet.py):
  _norm                    18-24     GroupNorm(groups=C//16) | BatchNorm(eval)
  ResidualBlock.forward    44-84
  ChessAttention           104-135 (mask), 137-190 (forward)
  ChessSpecificFeatures    229-244
  _forward_features        656-695
  _compute_policy_value    697-753
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

DEFAULTS = dict(
    planes=19, channels=160, blocks=14, policy_size=4672, se=True, se_ratio=0.25,
    attention=True, attention_heads=8, attention_unmasked_mix=0.2, attention_relbias=True,
    attention_every_k=3, chess_features=True, self_supervised=True, piece_square_tables=True,
    wdl=False, policy_factor_rank=0, norm="batch", activation="relu", value_activation="silu",
    preact=False, droppath=0.0, ssl_tasks=["piece"], infer_attention_stride=1,
    infer_amp_tower=False,
)

SSL_OUT = {"piece": 13, "threat": 1, "pin": 1, "fork": 1, "control": 3,
           "pawn_structure": 8, "king_safety": 3}


def full_cfg(cfg: dict) -> dict:
    """NetConfig defaults (resnet.py:247-282) merged with the given keys."""
    out = dict(DEFAULTS)
    for k, v in cfg.items():
        out[k] = v
    return out


def attention_mask() -> torch.Tensor:
    """bool [64,64]; resnet.py:104-130.  Token n = row*8+col of the tensor."""
    n = 8
    rows = torch.arange(n).repeat_interleave(n)
    cols = torch.arange(n).repeat(n)
    dr = rows[:, None] - rows[None, :]
    dc = cols[:, None] - cols[None, :]
    same_row = dr == 0
    same_col = dc == 0
    diag = dr.abs() == dc.abs()
    knight = ((dr.abs() == 2) & (dc.abs() == 1)) | ((dr.abs() == 1) & (dc.abs() == 2))
    adjacent = (dr.abs() <= 1) & (dc.abs() <= 1)
    return same_row | same_col | diag | knight | adjacent


def tower_layout(cfg: dict):
    """List of ('res'|'att', tower_index) in nn.Sequential order (resnet.py:346-356)."""
    cfg = full_cfg(cfg)
    out = []
    idx = 0
    k = int(cfg["attention_every_k"])
    for i in range(int(cfg["blocks"])):
        out.append(("res", idx)); idx += 1
        if cfg["attention"] and k > 0 and (i % k) == (k - 1):
            out.append(("att", idx)); idx += 1
    return out


def _norm(x, sd, prefix, cfg):
    if cfg["norm"] == "group":
        C = x.shape[1]
        return F.group_norm(x, max(1, C // 16), sd[prefix + ".weight"], sd[prefix + ".bias"], 1e-5)
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                        sd[prefix + ".weight"], sd[prefix + ".bias"], False, 0.0, 1e-5)


def _act(x, cfg):
    return F.silu(x) if cfg["activation"] == "silu" else F.relu(x)


def _res_block(x, sd, p, cfg):
    if cfg["preact"]:
        out = _act(_norm(x, sd, p + ".bn1", cfg), cfg)
        out = F.conv2d(out, sd[p + ".conv1.weight"], padding=1)
        out = _act(_norm(out, sd, p + ".bn2", cfg), cfg)
        out = F.conv2d(out, sd[p + ".conv2.weight"], padding=1)
    else:
        out = F.conv2d(x, sd[p + ".conv1.weight"], padding=1)
        out = _act(_norm(out, sd, p + ".bn1", cfg), cfg)
        out = F.conv2d(out, sd[p + ".conv2.weight"], padding=1)
        out = _norm(out, sd, p + ".bn2", cfg)
    if cfg["se"]:
        w = out.mean(dim=(2, 3))
        w = _act(F.linear(w, sd[p + ".se_fc1.weight"], sd[p + ".se_fc1.bias"]), cfg)
        w = torch.sigmoid(F.linear(w, sd[p + ".se_fc2.weight"], sd[p + ".se_fc2.bias"]))
        out = out * w[:, :, None, None]
    out = x + out
    if not cfg["preact"]:
        out = _act(out, cfg)
    return out


def _attention(x, sd, p, cfg, mask):
    B, C, H, W = x.shape
    heads = int(cfg["attention_heads"])
    D = C // heads
    n = H * W
    qkv = F.conv2d(x, sd[p + ".qkv.weight"]).reshape(B, 3, heads, D, n)
    qkv = qkv.permute(1, 0, 2, 4, 3)
    q, k, v = qkv[0], qkv[1], qkv[2]
    s = torch.matmul(q, k.transpose(-2, -1)) * (1.0 / math.sqrt(D))
    if cfg["attention_relbias"]:
        s = s + sd[p + ".rel_bias"]
    s = torch.clamp(s, -50.0, 50.0)
    mix = float(cfg["attention_unmasked_mix"])
    sm = s.masked_fill(~mask[None, None], -1e4)
    out_m = torch.matmul(F.softmax(sm, dim=-1), v)
    if 0.0 < mix < 1.0:
        out_u = torch.matmul(F.softmax(s, dim=-1), v)
        blend = 1.0 - mix
        out = blend * out_m + (1.0 - blend) * out_u
    elif mix >= 1.0:
        out = out_m
    else:
        out = torch.matmul(F.softmax(s, dim=-1), v)
    out = out.transpose(1, 2).reshape(B, n, C).transpose(1, 2).reshape(B, C, H, W)
    out = F.conv2d(out, sd[p + ".proj.weight"]) + x
    out = out.permute(0, 2, 3, 1)
    out = F.layer_norm(out, (C,), sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-5)
    return out.permute(0, 3, 1, 2).contiguous()


def _value_act(x, cfg):
    name = cfg["value_activation"]
    if name == "silu":
        return F.silu(x)
    if name == "leaky_relu":
        return F.leaky_relu(x, 0.05)
    return F.relu(x)


@torch.no_grad()
def forward(sd: Dict[str, torch.Tensor], cfg: dict, x: torch.Tensor, return_ssl: bool = False,
            return_feats: bool = False):
    """x f32 [B,planes,8,8] -> (p f32 [B,4672], v f32 [B], ssl dict|None)."""
    cfg = full_cfg(cfg)
    sd = {k: v.float() for k, v in sd.items()}
    x = x.float()
    mask = attention_mask()
    # stem (resnet.py:314-318, 662-664)
    x = F.conv2d(x, sd["stem.0.weight"], padding=1)
    x = _act(_norm(x, sd, "stem.1", cfg), cfg)
    # chess features (resnet.py:229-244)
    if cfg["chess_features"]:
        x = x + sd["chess_features.position_encoding"]
        if cfg["piece_square_tables"]:
            t = F.conv2d(x, sd["chess_features.pst_conv.weight"])
            x = x + _act(_norm(t, sd, "chess_features.pst_norm", cfg), cfg)
        t = F.conv2d(x, sd["chess_features.interaction_conv.weight"], padding=1)
        x = x + _act(_norm(t, sd, "chess_features.interaction_norm", cfg), cfg)
    # tower (resnet.py:676-689)
    stride = max(1, int(cfg["infer_attention_stride"]))
    att_seen = 0
    for kind, idx in tower_layout(cfg):
        p = f"tower.{idx}"
        if kind == "res":
            x = _res_block(x, sd, p, cfg)
        else:
            att_seen += 1
            if stride > 1 and (att_seen % stride) != 0:
                continue
            x = _attention(x, sd, p, cfg, mask)
    feats = torch.nan_to_num(x, nan=0.0, posinf=0.0, neginf=0.0)
    # policy (resnet.py:699-711)
    pf = F.conv2d(feats, sd["policy_head.0.weight"])
    pf = _act(_norm(pf, sd, "policy_head.1", cfg), cfg)
    pflat = pf.reshape(pf.shape[0], -1)
    if int(cfg["policy_factor_rank"]) > 0:
        p_ = F.relu(F.linear(pflat, sd["policy_fc1.weight"], sd["policy_fc1.bias"]))
        p_ = F.linear(p_, sd["policy_fc2.weight"], sd["policy_fc2.bias"])
    else:
        p_ = F.linear(pflat, sd["policy_fc.weight"], sd["policy_fc.bias"])
    scale = torch.clamp(F.softplus(sd["_policy_logit_scale_raw"]) + 1e-3, max=5.0)
    p_ = torch.nan_to_num(p_ * scale, nan=0.0, posinf=0.0, neginf=0.0)
    # value (resnet.py:721-734)
    v = F.conv2d(feats, sd["value_head.0.weight"])
    v = _act(_norm(v, sd, "value_head.1", cfg), cfg)
    v = F.conv2d(v, sd["value_head.3.weight"])
    v = _act(_norm(v, sd, "value_head.4", cfg), cfg)
    v = v.reshape(v.shape[0], -1)
    v = _value_act(F.linear(v, sd["value_fc1.weight"], sd["value_fc1.bias"]), cfg)
    v = _value_act(F.linear(v, sd["value_fc2.weight"], sd["value_fc2.bias"]), cfg)
    gate = torch.sigmoid(F.linear(v, sd["value_gate.0.weight"], sd["value_gate.0.bias"]))
    v = v * gate
    v = torch.tanh(F.linear(v, sd["value_fc3.weight"], sd["value_fc3.bias"])).squeeze(-1)
    ssl = None
    if return_ssl and cfg["self_supervised"]:
        ssl = {}
        for task in cfg["ssl_tasks"]:
            if task not in SSL_OUT:
                continue
            h = f"ssl_heads.{task}"
            t = F.conv2d(feats, sd[h + ".0.weight"])
            t = _act(_norm(t, sd, h + ".1", cfg), cfg)
            ssl[task] = F.conv2d(t, sd[h + ".3.weight"])
    if return_feats:
        return p_, v, ssl, feats
    return p_, v, ssl


def param_shapes(cfg: dict) -> Dict[str, Tuple[int, ...]]:
    """State-dict key -> shape for a config; used to build
    random-init weights without the reference module."""
    cfg = full_cfg(cfg)
    C = int(cfg["channels"]); P = int(cfg["planes"]); H = int(cfg["attention_heads"])
    out: Dict[str, Tuple[int, ...]] = {}

    def norm(prefix, ch):
        out[prefix + ".weight"] = (ch,); out[prefix + ".bias"] = (ch,)
        if cfg["norm"] != "group":
            out[prefix + ".running_mean"] = (ch,); out[prefix + ".running_var"] = (ch,)

    out["stem.0.weight"] = (C, P, 3, 3); norm("stem.1", C)
    if cfg["chess_features"]:
        out["chess_features.position_encoding"] = (1, C, 8, 8)
        if cfg["piece_square_tables"]:
            out["chess_features.pst_conv.weight"] = (C, C, 1, 1); norm("chess_features.pst_norm", C)
        out["chess_features.interaction_conv.weight"] = (C, C, 3, 3)
        norm("chess_features.interaction_norm", C)
    hidden = max(8, int(C * float(cfg["se_ratio"])))
    for kind, idx in tower_layout(cfg):
        p = f"tower.{idx}"
        if kind == "res":
            out[p + ".conv1.weight"] = (C, C, 3, 3); norm(p + ".bn1", C)
            out[p + ".conv2.weight"] = (C, C, 3, 3); norm(p + ".bn2", C)
            if cfg["se"]:
                out[p + ".se_fc1.weight"] = (hidden, C); out[p + ".se_fc1.bias"] = (hidden,)
                out[p + ".se_fc2.weight"] = (C, hidden); out[p + ".se_fc2.bias"] = (C,)
        else:
            out[p + ".qkv.weight"] = (3 * C, C, 1, 1); out[p + ".proj.weight"] = (C, C, 1, 1)
            out[p + ".norm.weight"] = (C,); out[p + ".norm.bias"] = (C,)
            if cfg["attention_relbias"]:
                out[p + ".rel_bias"] = (1, H, 64, 64)
    out["policy_head.0.weight"] = (64, C, 1, 1); norm("policy_head.1", 64)
    r = int(cfg["policy_factor_rank"])
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
    if cfg["self_supervised"]:
        for task in cfg["ssl_tasks"]:
            if task in SSL_OUT:
                h = f"ssl_heads.{task}"
                out[h + ".0.weight"] = (C // 2, C, 1, 1); norm(h + ".1", C // 2)
                out[h + ".3.weight"] = (SSL_OUT[task], C // 2, 1, 1)
    return out


def random_state_dict(cfg: dict, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Synthetic random-init weights of the right architecture (bench/tests on the
    GPU box, where the reference module does not exist).  Scales are chosen so that
    activations stay O(1) through the tower (fan-in scaled normal), norm gains ~1."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in param_shapes(cfg).items():
        if name == "_policy_logit_scale_raw":
            sd[name] = torch.tensor(math.log(math.expm1(0.2 - 1e-3)))
        elif name.endswith("running_var"):
            sd[name] = torch.rand(shape, generator=g) * 0.5 + 0.75
        elif name.endswith("running_mean"):
            sd[name] = torch.randn(shape, generator=g) * 0.1
        elif len(shape) == 1 and name.endswith(".weight"):
            sd[name] = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(".bias"):
            sd[name] = 0.05 * torch.randn(shape, generator=g)
        elif name.endswith("position_encoding"):
            sd[name] = 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("rel_bias"):
            sd[name] = 0.5 * torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            sd[name] = torch.randn(shape, generator=g) * (1.0 / math.sqrt(max(1, fan_in)))
    return sd
