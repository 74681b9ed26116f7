"""CPU cross-check of the bitboard core the device kernels are built from
(matrix0_amd/csrc/chess_core.h, compiled for the host by tests/host_shim) against the oracle:
perft, legal-move ORDER, policy indices, planes, state after moves, key equality, irreversibility."""
import ctypes as C
import gzip
import json
import os
import subprocess

import numpy as np
import pytest

from oracle import chess_py as ch

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def shim():
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "host_shim")])
    l = C.CDLL(os.path.join(HERE, "_build", "libchess_shim.so"))
    l.hc_perft.restype = C.c_uint64
    l.hc_perft.argtypes = [C.c_char_p, C.c_int]
    return l


def _legal(shim, fen):
    mv = (C.c_int32 * 256)()
    idx = (C.c_int32 * 256)()
    n = shim.hc_legal(fen.encode(), mv, idx)
    assert n >= 0
    return [(mv[i] & 255, (mv[i] >> 8) & 255, (mv[i] >> 16) or None) for i in range(n)], [idx[i] for i in range(n)]


PERFT = [
    (ch.START_FEN, 4, 197281),
    ("r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1", 3, 97862),
    ("8/2p5/3p4/KP5r/1R3p1k/8/4P1P1/8 w - - 0 1", 5, 674624),
    ("r3k2r/Pppp1ppp/1b3nbN/nP6/BBP1P3/q4N2/Pp1P2PP/R2Q1RK1 w kq - 0 1", 4, 422333),
    ("rnbq1k1r/pp1Pbppp/2p5/8/2B5/8/PPP1NnPP/RNBQK2R w KQ - 1 8", 3, 62379),
    ("r4rk1/1pp1qppp/p1np1n2/2b1p1B1/2B1P1b1/P1NP1N2/1PP1QPPP/R4RK1 w - - 0 10", 3, 89890),
]


@pytest.mark.parametrize("fen,d,n", PERFT)
def test_perft(shim, fen, d, n):
    assert shim.hc_perft(fen.encode(), d) == n


def test_move_order_indices_planes_on_reference_fens(shim):
    rows = json.load(gzip.open(os.path.join(HERE, "golden", "tactical_legal_counts.json.gz"), "rt"))
    buf = np.zeros((19, 8, 8), np.float32)
    for fen, n, _ in rows[::4]:
        b = ch.Board(fen)
        moves, idxs = ch.legal_moves_with_indices(b)
        hm, hi = _legal(shim, fen)
        assert len(hm) == n
        assert hm == [(m.from_square, m.to_square, m.promotion) for m in moves], fen   # same ORDER
        assert hi == idxs, fen
        assert shim.hc_encode(fen.encode(), buf.ctypes.data_as(C.c_void_p)) == 0
        assert np.array_equal(buf, ch.encode_board(b)), fen


def test_two_phase_generator_equals_sequential(shim):
    """pseudo-legal list + per-move legality filter (what the tree kernels do one move per lane) == gen_legal."""
    rows = json.load(gzip.open(os.path.join(HERE, "golden", "tactical_legal_counts.json.gz"), "rt"))
    mv = (C.c_int32 * 256)()
    for fen, n, _ in rows[::3]:
        k = shim.hc_legal_two_phase(fen.encode(), mv)
        assert k == n
        assert [(mv[i] & 255, (mv[i] >> 8) & 255, (mv[i] >> 16) or None) for i in range(k)] == _legal(shim, fen)[0], fen


def test_random_playouts_state_keys_irreversible(shim):
    rng = np.random.default_rng(11)
    for g in range(40):
        b = ch.Board()
        ucis, okeys = [], [b._transposition_key()]
        for ply in range(int(rng.integers(5, 160))):
            if b.is_game_over():
                break
            lm = b.legal_moves
            m = lm[int(rng.integers(len(lm)))]
            ucis.append(m.uci())
            b.push(m)
            okeys.append(b._transposition_key())
        n = len(ucis)
        arr = (C.c_char_p * max(1, n))(*[u.encode() for u in ucis])
        out = (C.c_int32 * 8)()
        keys = (C.c_uint64 * (n + 1))()
        irr = (C.c_int32 * max(1, n))()
        planes = np.zeros((19, 8, 8), np.float32)
        assert shim.hc_play(ch.START_FEN.encode(), arr, n, out, keys, irr, planes.ctypes.data_as(C.c_void_p)) == 0
        assert np.array_equal(planes, ch.encode_board(b))
        assert bool(out[0]) == b.is_check() and bool(out[1]) == b.is_insufficient_material()
        assert out[4] == b.halfmove_clock and out[5] == b.fullmove_number and out[7] == int(b.turn)
        assert out[6] == (-1 if b.ep_square is None else b.ep_square)
        # key equality structure identical to the oracle's tuple keys
        hk = [keys[i] for i in range(n + 1)]
        for i in range(0, n + 1, 3):
            for j in range(i + 1, n + 1, 2):
                assert (hk[i] == hk[j]) == (okeys[i] == okeys[j])
        # legal moves + order at the final position
        hm, hi = _legal(shim, b.fen().replace(" - ", " - ") if False else _fen_raw(b))
        moves, idxs = ch.legal_moves_with_indices(b)
        assert hm == [(m.from_square, m.to_square, m.promotion) for m in moves] and hi == idxs


def _fen_raw(b):
    """FEN with the raw ep square (python-chess keeps ep_square after every double push)."""
    f = b.fen().split(" ")
    if b.ep_square is not None:
        f[3] = ch.square_name(b.ep_square)
    return " ".join(f)
