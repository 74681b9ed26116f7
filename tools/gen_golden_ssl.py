#!/usr/bin/env python3
"""Golden vectors for the SSL target generators: runs the REAL reference azchess/ssl_algorithms.py
(ChessSSLAlgorithms.create_enhanced_ssl_targets, file loaded directly with an empty `chess` stub) on planes of
FENs taken from the reference's fixture data.  Build container only.  Output tests/golden/ssl_targets.npz:
   fens (str), planes u8-coded, piece i8 [N,13,8,8], threat/pin/fork/control i8 [N,8,8]."""
import gzip, importlib.util, json, os, sys, types
import numpy as np
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from oracle import chess_py as ch
REF = os.environ.get("M0_REFERENCE", "/root/reference")
sys.modules.setdefault("chess", types.ModuleType("chess"))
spec = importlib.util.spec_from_file_location("ref_ssl", os.path.join(REF, "azchess/ssl_algorithms.py"))
mod = importlib.util.module_from_spec(spec); sys.modules["ref_ssl"] = mod; spec.loader.exec_module(mod)
alg = mod.ChessSSLAlgorithms()
rows = json.load(gzip.open(os.path.join(ROOT, "tests/golden/tactical_legal_counts.json.gz"), "rt"))
sf = json.load(gzip.open(os.path.join(ROOT, "tests/golden/stockfish_best_moves.json.gz"), "rt"))
fens = [ch.START_FEN] + [r[0] for r in rows[::125]] + [r[0] for r in sf[::100]]
out = {"piece": [], "threat": [], "pin": [], "fork": [], "control": []}
for fen in fens:
    x = torch.from_numpy(ch.encode_board(ch.Board(fen))[None]).float()       # batch of 1, as the worker calls it
    t = alg.create_enhanced_ssl_targets(x)
    for k in out:
        out[k].append(t[k].squeeze(0).numpy())
blob = {k: np.stack(v).astype(np.int8) for k, v in out.items()}
for k, v in out.items():
    assert np.array_equal(np.stack(v), blob[k].astype(np.stack(v).dtype)), k     # integer valued
blob["fens"] = np.array(fens)
np.savez_compressed(os.path.join(ROOT, "tests/golden/ssl_targets.npz"), **blob)
print(len(fens), {k: (blob[k].shape, int(np.abs(blob[k]).sum())) for k in out})
