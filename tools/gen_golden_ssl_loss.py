#!/usr/bin/env python3
"""Golden vectors for the SSL LOSS heads: runs the REAL reference module's
PolicyValueNet.get_enhanced_ssl_loss / _compute_task_loss (azchess/model/resnet.py:892-1130, called from
training/train.py:285-305) on a small GroupNorm/SiLU network with all five SSL heads, with targets produced by the REAL
azchess/ssl_algorithms.py on encoded positions.  Build container only.

Output tests/golden/ssl_loss.npz: the planes, the targets, the five heads' outputs of the reference forward (eval mode), and
for several (task subset, per-task weight) configurations the total loss the reference returned, plus the single-task losses.
The network itself is tests/golden/net_gn_silu_preact.npz's (same seed and re-randomisation: tools/gen_golden_net.py)."""
import importlib.util
import json
import logging
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..")
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import gen_golden_net as gnet          # noqa: E402
from oracle import chess_py as ch      # noqa: E402

REF = os.environ.get("M0_REFERENCE", "/root/reference")
logging.disable(logging.CRITICAL)


def main():
    mod = gnet.load_reference_resnet()
    spec = importlib.util.spec_from_file_location("ref_ssl", os.path.join(REF, "azchess/ssl_algorithms.py"))
    sslmod = importlib.util.module_from_spec(spec)
    sys.modules["ref_ssl"] = sslmod
    spec.loader.exec_module(sslmod)
    alg = sslmod.ChessSSLAlgorithms()

    cfg, _ = gnet.CASES["gn_silu_preact"]
    gold = np.load(os.path.join(gnet.OUT, "net_gn_silu_preact.npz"))
    torch.manual_seed(1234)
    net = mod.PolicyValueNet(mod.NetConfig(**cfg)).eval()
    sd = {}
    for k in gold.files:
        if k.startswith("sd::"):
            sd[k[4:]] = torch.from_numpy(gold[k])
        elif k.startswith("sdseed::"):
            seed, scale = gold[k]
            shape = net.state_dict()[k[8:]].shape
            sd[k[8:]] = (float(scale) * torch.randn(shape, generator=torch.Generator().manual_seed(int(seed)))).half().float()
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected

    fens = [ch.START_FEN, "r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1",
            "r1bq1rk1/pp2bppp/2n1pn2/2pp4/3P1B2/2PBPN2/PP1N1PPP/R2QK2R b KQ - 3 8", "8/2p5/3p4/KP5r/1R3p1k/8/4P1P1/8 w - - 0 1",
            "6k1/5ppp/8/8/8/8/5PPP/3R2K1 w - - 0 1", "rnbq1k1r/pp1Pbppp/2p5/8/2B5/8/PPP1NnPP/RNBQK2R w KQ - 1 8"]
    x = torch.from_numpy(np.stack([ch.encode_board(ch.Board(f)) for f in fens])).float()
    targets = alg.create_enhanced_ssl_targets(x)
    with torch.no_grad():
        _, _, heads = net(x, return_ssl=True)
    blob = {"fens": np.array(fens), "x": x.numpy(), "cfg_json": np.array(json.dumps(cfg))}
    for t in ("piece", "threat", "pin", "fork", "control"):
        blob[f"target_{t}"] = targets[t].numpy().astype(np.float32)
        blob[f"head_{t}"] = heads[t].numpy().astype(np.float32)
    cases = []
    configs = [
        (["piece", "threat", "pin", "fork", "control"], {}),
        (["piece"], {}),
        (["threat", "fork"], {"threat": 2.0}),
        (["control", "pin"], {"control": 0.5, "pin": 3.0}),
        (["piece", "control"], {"control": 0.25}),
    ]
    for tasks, weights in configs:
        net.cfg.ssl_tasks = list(tasks)
        for t in ("threat", "pin", "fork", "control"):
            if hasattr(net.cfg, f"ssl_{t}_weight"):
                delattr(net.cfg, f"ssl_{t}_weight")
        for t, w in weights.items():
            setattr(net.cfg, f"ssl_{t}_weight", w)
        with torch.no_grad():
            total = float(net.get_enhanced_ssl_loss(x, {k: v for k, v in targets.items()}))
        single = {}
        for t in tasks:                                   # targets restricted to one task: that task's (weighted) term
            with torch.no_grad():
                single[t] = float(net.get_enhanced_ssl_loss(x, {t: targets[t]}))
        cases.append({"tasks": tasks, "weights": weights, "total": total, "single": single})
        print(tasks, weights, "-> total", total, single)
    # degenerate targets: all-zero threat map (BCE still > 0), piece targets as class indices instead of one-hot
    net.cfg.ssl_tasks = ["piece", "threat"]
    with torch.no_grad():
        idx_total = float(net.get_enhanced_ssl_loss(x, {"piece": torch.argmax(targets["piece"], dim=1), "threat": torch.zeros_like(targets["threat"])}))
    cases.append({"tasks": ["piece", "threat"], "weights": {}, "total": idx_total, "single": {}, "piece_as_index": True, "threat_zero": True})
    blob["cases_json"] = np.array(json.dumps(cases))
    np.savez_compressed(os.path.join(gnet.OUT, "ssl_loss.npz"), **blob)
    print("wrote ssl_loss.npz", os.path.getsize(os.path.join(gnet.OUT, "ssl_loss.npz")), "bytes")


if __name__ == "__main__":
    main()
