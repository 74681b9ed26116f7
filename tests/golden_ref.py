"""Loaders for the golden files produced by the real reference code (tools/gen_golden_mcts.py)."""
import gzip
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_json(name):
    with gzip.open(os.path.join(GOLD, name), "rt") as f:
        return json.load(f)


def load_npz(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def uci(code):
    if code == 0:
        return "0000"                                  # chess.Move.null()
    f, t, p = code & 63, (code >> 6) & 63, (code >> 12) & 7
    s = "abcdefgh"[f & 7] + str((f >> 3) + 1) + "abcdefgh"[t & 7] + str((t >> 3) + 1)
    return s + ("  nbrq"[p] if p else "")          # p = python-chess piece type (KNIGHT=2 .. QUEEN=5)


def planes_from_bits(bits17x8, counters2):
    """ref_encoding.npz stores planes 0..16 bit-packed (they are exactly 0/1) and the two constant planes as their
    float32 value; this rebuilds the float32 [19,8,8] tensor encode_board returned."""
    p = np.zeros((19, 8, 8), np.float32)
    p[:17] = np.unpackbits(bits17x8, axis=1).reshape(17, 8, 8).astype(np.float32)
    p[17] = counters2[0]
    p[18] = counters2[1]
    return p
