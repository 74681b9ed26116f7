"""Pin oracle/ssl_ref.py against outputs of the real reference ssl_algorithms.py (tests/golden/ssl_targets.npz)."""
import os

import numpy as np

from oracle import chess_py as ch
from oracle import ssl_ref

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ssl_targets.npz")


def test_ssl_targets_match_reference_goldens():
    z = np.load(GOLDEN)
    fens = [str(f) for f in z["fens"]]
    assert len(fens) == 121
    for i, fen in enumerate(fens):
        t = ssl_ref.targets(ch.encode_board(ch.Board(fen)))
        for k in ("piece", "threat", "pin", "fork", "control"):
            assert np.array_equal(t[k], z[k][i]), (fen, k)
    assert z["fork"].sum() > 0 and z["threat"].sum() > 0 and (z["control"] < 0).any() and z["pin"].sum() == 0
