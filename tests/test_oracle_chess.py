"""Pin the oracle's chess rules + azchess/encoding.py restatement (oracle/chess_oracle.c) against
(1) published perft known answers, (2) the reference's own fixtures and test assertions
(tests/test_encoding.py, tests/test_board_tensor.py, tests/test_encoding_random.py,
azchess/validate_moves.py:29-66, data/tactical/tactical_metadata.json, data/stockfish_games)."""
import gzip
import json
import os

import numpy as np
import pytest

from oracle import chess_py as ch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KIWIPETE = "r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1"

PERFT = [
    (ch.START_FEN, [20, 400, 8902, 197281]),
    (KIWIPETE, [48, 2039, 97862]),
    ("8/2p5/3p4/KP5r/1R3p1k/8/4P1P1/8 w - - 0 1", [14, 191, 2812, 43238, 674624]),
    ("r3k2r/Pppp1ppp/1b3nbN/nP6/BBP1P3/q4N2/Pp1P2PP/R2Q1RK1 w kq - 0 1", [6, 264, 9467, 422333]),
    ("rnbq1k1r/pp1Pbppp/2p5/8/2B5/8/PPP1NnPP/RNBQK2R w KQ - 1 8", [44, 1486, 62379]),
    ("r4rk1/1pp1qppp/p1np1n2/2b1p1B1/2B1P1b1/P1NP1N2/1PP1QPPP/R4RK1 w - - 0 10", [46, 2079, 89890]),
]


@pytest.mark.parametrize("fen,counts", PERFT)
def test_perft_known_answers(fen, counts):
    b = ch.Board(fen)
    for d, c in enumerate(counts, 1):
        assert b.perft(d) == c, (fen, d)


def test_tactical_metadata_legal_counts():
    """10 000 FENs with the reference data's `legal_moves` count; `move` is a legal move."""
    rows = json.load(gzip.open(os.path.join(GOLDEN, "tactical_legal_counts.json.gz"), "rt"))
    assert len(rows) == 10000
    for fen, n, mv in rows:
        b = ch.Board(fen)
        lm = b.legal_moves
        assert len(lm) == n, fen
        if mv:
            assert ch.Move.from_uci(mv) in lm, (fen, mv)


def test_stockfish_best_moves_are_legal_and_encodable():
    rows = json.load(gzip.open(os.path.join(GOLDEN, "stockfish_best_moves.json.gz"), "rt"))
    for fen, mv in rows:
        b = ch.Board(fen)
        m = ch.Move.from_uci(mv)
        assert m in b.legal_moves, (fen, mv)
        assert 0 <= ch.move_to_index(b, m) < 4672


# ---- tests/test_encoding.py restated ----
def test_castling_indices_different():
    b = ch.Board("r3k2r/8/8/8/8/8/8/R3K2R w KQkq - 0 1")
    assert ch.move_to_index(b, ch.Move.from_uci("e1g1")) != ch.move_to_index(b, ch.Move.from_uci("e1c1"))
    # king 2-step E / W ray from e1: dir E=2, W=3, steps=2
    assert ch.move_to_index(b, ch.Move.from_uci("e1g1")) == 4 * 73 + 2 * 7 + 1
    assert ch.move_to_index(b, ch.Move.from_uci("e1c1")) == 4 * 73 + 3 * 7 + 1


def test_en_passant_and_promotions():
    b = ch.Board("8/8/8/3pP3/8/8/8/8 w - d6 0 2")
    assert 0 <= ch.move_to_index(b, ch.Move.from_uci("e5d6")) < 4672
    b = ch.Board("8/P7/8/8/8/8/8/4k2K w - - 0 1")
    n = ch.move_to_index(b, ch.Move.from_uci("a7a8n"))
    q = ch.move_to_index(b, ch.Move.from_uci("a7a8q"))
    assert n != q and n == 48 * 73 + 64 and q == 48 * 73 + 0


def test_kiwipete_indices_unique_and_decodable():
    b = ch.Board(KIWIPETE)
    moves, idxs = ch.legal_moves_with_indices(b)
    assert len(moves) == 48 and len(set(idxs)) == 48
    for m, i in zip(moves, idxs):
        frm, off = divmod(i, 73)
        assert frm == m.from_square and 0 <= off < 73


def test_legal_mask_startpos():
    b = ch.Board()
    m = ch.get_legal_actions(b)
    assert m.shape == (4672,) and m.dtype == bool and m.sum() == 20
    b.push(ch.Move.from_uci("e2e4"))
    assert ch.get_legal_actions(b).sum() == 20


def test_board_encoding_startpos():
    e = ch.encode_board(ch.Board())
    assert e.shape == (19, 8, 8) and e.dtype == np.float32
    assert np.all(e[0][6, :] == 1.0) and np.all(e[6][1, :] == 1.0)
    for i in range(12, 17):
        assert np.all(e[i] == 1.0)


def test_board_tensor_fen():  # tests/test_board_tensor.py
    t = ch.encode_board(ch.Board("rnbqkbnr/pppppppp/8/8/4P3/8/PPPP1PPP/RNBQKBNR b KQkq e3 0 1"))
    assert t[0, 4, 4] == 1.0 and t[6, 1, 0] == 1.0
    assert np.all(t[12] == 0.0)
    for i in range(13, 17):
        assert np.all(t[i] == 1.0)
    assert abs(t[17].mean()) < 1e-6 and abs(t[18].mean() - 0.005025) < 1e-6
    assert t[18, 0, 0] == np.float32(1 / 199.0)


def test_illegal_move_raises():
    with pytest.raises(ValueError):
        ch.move_to_index(ch.Board(), ch.Move.from_uci("a1a8"))


def test_validate_moves_edge_cases():  # azchess/validate_moves.py:29-66
    for fen, uci in [("r3k2r/pppppppp/8/8/8/8/PPPPPPPP/R3K2R w KQkq - 0 1", "e1g1"),
                     ("r3k2r/pppppppp/8/8/8/8/PPPPPPPP/R3K2R w KQkq - 0 1", "e1c1"),
                     ("r3k2r/pppppppp/8/8/8/8/PPPPPPPP/R3K2R b KQkq - 0 1", "e8g8"),
                     ("r3k2r/pppppppp/8/8/8/8/PPPPPPPP/R3K2R b KQkq - 0 1", "e8c8")]:
        b = ch.Board(fen)
        assert ch.Move.from_uci(uci) in b.legal_moves
        assert 0 <= ch.move_to_index(b, ch.Move.from_uci(uci)) < 4672
    b = ch.Board("rnbqkbnr/pppppppp/8/8/4P3/8/PPPP1PPP/RNBQKBNR b KQkq e3 0 1")
    b.push(ch.Move.from_uci("d7d5"))
    assert ch.Move.from_uci("e4d5") in b.legal_moves
    for promo in "qnbr":
        t = ch.Board("8/3P4/8/8/8/8/8/8 w - - 0 1")
        assert ch.Move.from_uci("d7d8" + promo) in t.legal_moves


def test_random_playouts_unique_indices_and_rules():
    """tests/test_encoding_random.py + rule predicates on random playouts (seeded)."""
    rng = np.random.default_rng(7)
    ends = set()
    for g in range(60):
        b = ch.Board()
        for ply in range(300):
            if b.is_game_over(claim_draw=True):
                ends.add(b.outcome_code(True))
                break
            moves, idxs = ch.legal_moves_with_indices(b)
            assert len(set(idxs)) == len(idxs) and min(idxs) >= 0 and max(idxs) < 4672
            assert ch.get_legal_actions(b).sum() == len(moves)
            b.push(moves[int(rng.integers(len(moves)))])
    assert ends  # some games ended by rule


def test_repetition_and_claims():
    b = ch.Board()
    for u in ["g1f3", "g8f6", "f3g1", "f6g8", "g1f3", "g8f6", "f3g1"]:
        b.push(ch.Move.from_uci(u))
    assert not b.is_repetition(3)
    assert b.can_claim_threefold_repetition()       # f6g8 would repeat the start position a 3rd time
    b.push(ch.Move.from_uci("f6g8"))
    assert b.is_repetition(3) and not b.is_repetition(5) and b.result(claim_draw=True) == "1/2-1/2"
    assert not b.is_game_over() and b.is_game_over(claim_draw=True)
    k = ch.Board("8/8/8/8/8/4k3/8/4K2R w K - 99 80")
    assert k.can_claim_fifty_moves() and not ch.Board("8/8/8/8/8/4k3/8/4K2R w K - 98 80").can_claim_fifty_moves()
    assert ch.Board("8/8/8/8/8/4k3/8/4K3 w - - 0 1").is_insufficient_material()
    assert ch.Board("8/8/8/8/8/4k3/8/4KB2 w - - 0 1").is_insufficient_material()
    assert not ch.Board("8/8/8/8/8/4k3/8/3NKN2 w - - 0 1").is_insufficient_material()
    assert ch.Board("7k/5Q2/6K1/8/8/8/8/8 b - - 0 1").is_stalemate()
    assert ch.Board("7k/6Q1/6K1/8/8/8/8/8 b - - 0 1").is_checkmate()
    assert ch.Board("8/8/8/8/8/4k3/8/4K2R w - - 150 90").is_seventyfive_moves()


def test_castling_rights_cleaning():
    b = ch.Board("r3k2r/8/8/8/8/8/8/R3K2R w KQkq - 0 1")
    assert all([b.has_kingside_castling_rights(True), b.has_queenside_castling_rights(True),
                b.has_kingside_castling_rights(False), b.has_queenside_castling_rights(False)])
    b.push(ch.Move.from_uci("h1h8"))   # captures the h8 rook: both h-side rights gone
    assert not b.has_kingside_castling_rights(True) and not b.has_kingside_castling_rights(False)
    assert b.has_queenside_castling_rights(True) and b.has_queenside_castling_rights(False)
    # FEN claims rights but the king is off e1: cleaned away
    assert not ch.Board("r3k2r/8/8/8/8/8/8/R2K3R w KQkq - 0 1").has_kingside_castling_rights(True)
