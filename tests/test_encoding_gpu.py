"""GPU parity (bit-exact): device encode_board / legal mask / move order / move_to_index vs the oracle,
on the reference's own FEN fixtures and test cases, called through the C-ABI (m0_encode_fens)."""
import gzip
import json
import os

import numpy as np
import pytest

from oracle import chess_py as ch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KIWIPETE = "r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1"


def test_device_encoding_matches_oracle_on_reference_fens():
    from matrix0_amd import encoding as enc
    rows = json.load(gzip.open(os.path.join(GOLDEN, "tactical_legal_counts.json.gz"), "rt"))
    fens = [r[0] for r in rows]                       # all 10 000
    planes, mask, moves = enc.encode_fens(fens)
    for i, (fen, n, _) in enumerate(rows):
        assert len(moves[i][0]) == n and mask[i].sum() == n
    for i in range(0, len(fens), 5):
        b = ch.Board(fens[i])
        om, oi = ch.legal_moves_with_indices(b)
        assert moves[i][0] == [m.uci() for m in om], fens[i]       # same ORDER
        assert moves[i][1] == oi
        assert np.array_equal(planes[i], ch.encode_board(b))
        assert np.array_equal(mask[i], ch.get_legal_actions(b))


def test_reference_unit_cases_on_device():
    from matrix0_amd import encoding as enc
    # tests/test_encoding.py
    f = "r3k2r/8/8/8/8/8/8/R3K2R w KQkq - 0 1"
    assert enc.move_to_index(f, "e1g1") != enc.move_to_index(f, "e1c1")
    assert 0 <= enc.move_to_index("8/8/8/3pP3/8/8/8/8 w - d6 0 2", "e5d6") < 4672
    f = "8/P7/8/8/8/8/8/4k2K w - - 0 1"
    assert enc.move_to_index(f, "a7a8n") != enc.move_to_index(f, "a7a8q")
    planes, mask, moves = enc.encode_fens([ch.START_FEN, KIWIPETE])
    assert mask[0].sum() == 20 and mask.dtype == bool and mask.shape == (2, 4672)
    assert len(set(moves[1][1])) == 48
    e = planes[0]
    assert e.dtype == np.float32 and np.all(e[0][6, :] == 1.0) and np.all(e[6][1, :] == 1.0)
    for i in range(12, 17):
        assert np.all(e[i] == 1.0)
    with pytest.raises(ValueError):
        enc.move_to_index(ch.START_FEN, "a1a8")
    # tests/test_board_tensor.py
    t = enc.encode_board("rnbqkbnr/pppppppp/8/8/4P3/8/PPPP1PPP/RNBQKBNR b KQkq e3 0 1")
    assert t[0, 4, 4] == 1.0 and t[6, 1, 0] == 1.0 and np.all(t[12] == 0)
    assert abs(t[17].mean()) < 1e-6 and abs(t[18].mean() - 0.005025) < 1e-6
    # tests/test_encoding_random.py
    perm = enc.build_horizontal_flip_permutation()
    assert np.array_equal(np.arange(73)[perm][perm], np.arange(73))
    rot = enc.build_rotate180_permutation()
    assert np.array_equal(np.arange(73)[rot][rot], np.arange(73))


def test_ssl_targets_on_device_match_reference_goldens_and_oracle():
    """ssl_algorithms.py targets computed by the HIP kernel: bit-exact vs the real reference's outputs
    (tests/golden/ssl_targets.npz) and vs the oracle on further reference FENs."""
    from matrix0_amd import engine as eng
    from oracle import ssl_ref
    z = np.load(os.path.join(GOLDEN, "ssl_targets.npz"))
    fens = [str(f) for f in z["fens"]]
    t = eng.ssl_targets_fens(fens)
    for k in ("piece", "threat", "pin", "fork", "control"):
        assert np.array_equal(t[k], z[k].astype(np.float32)), k
    rows = json.load(gzip.open(os.path.join(GOLDEN, "tactical_legal_counts.json.gz"), "rt"))
    fens2 = [r[0] for r in rows[7::40]]
    t2 = eng.ssl_targets_fens(fens2)
    for i, fen in enumerate(fens2):
        o = ssl_ref.targets(ch.encode_board(ch.Board(fen)))
        for k in ("piece", "threat", "pin", "fork", "control"):
            assert np.array_equal(t2[k][i], o[k].astype(np.float32)), (fen, k)


def test_decode_move_roundtrip_and_fallbacks():
    """MoveEncoder.decode_move (encoding.py:174-229): Kiwipete round trip (tests/test_encoding.py), then arbitrary
    indices (illegal / off-board / under-promotion slots) against the oracle's restatement incl. the fallbacks."""
    from matrix0_amd import encoding as enc
    me = enc.MoveEncoder()
    b = ch.Board(KIWIPETE)
    for m, i in zip(*ch.legal_moves_with_indices(b)):
        assert me.decode_move(KIWIPETE, i) == m.uci()
    rng = np.random.default_rng(4)
    rows = json.load(gzip.open(os.path.join(GOLDEN, "tactical_legal_counts.json.gz"), "rt"))
    for fen, _, _ in rows[::500] + [["8/P7/8/8/8/8/8/4k2K w - - 0 1", 0, ""], ["8/8/8/8/8/8/p7/4K2k b - - 0 1", 0, ""]]:
        bb = ch.Board(fen)
        for idx in rng.integers(0, 4672, size=60):
            want = ch.decode_move(bb, int(idx)).uci()
            assert me.decode_move(fen, int(idx)) == want, (fen, int(idx))
    with pytest.raises(ValueError):
        me.decode_move(KIWIPETE, 4672)
