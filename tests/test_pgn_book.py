"""PGN opening book (azchess/selfplay/internal.py:39-69: chess.pgn.read_game + the board after each of the first 20 plies) read
by matrix0_amd/pgn_book.py on the host: the reference's own book file (tests/golden/main_eval_book.pgn = its
data/openings/main_eval_book.pgn) and PGN text rebuilt from the 125 games python-chess wrote under data/eval_games
(tests/golden/eval_games_san.json.gz), with comments, variations, NAGs and odd spacing mixed in.  Every position is compared
with the oracle's board after the same moves (Board.fen(): cleaned castling rights, legal-only en-passant square)."""
import gzip
import json
import os

import pytest

from matrix0_amd import engine as eng
from matrix0_amd import pgn_book
from oracle import chess_py as ch

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _oracle_fens(tokens_uci, start=ch.START_FEN, limit=20):
    b = ch.Board(start)
    out = []
    for u in tokens_uci[:limit]:
        b.push(ch.Move.from_uci(u))
        out.append(b.fen())
    return out


def test_reference_book_file():
    book = pgn_book.load_opening_book(os.path.join(GOLD, "main_eval_book.pgn"))
    lines = [["e2e4", "c7c5", "g1f3", "d7d6"], ["d2d4", "g8f6", "c2c4", "g7g6"], ["g1f3", "d7d5", "g2g3", "c7c5"]]
    want = [f for l in lines for f in _oracle_fens(l)]
    assert book == want and len(book) == 12
    assert pgn_book.load_opening_book("/nonexistent/book.pgn") == []


def test_eval_games_as_a_book_with_noise():
    games = json.load(gzip.open(os.path.join(GOLD, "eval_games_san.json.gz"), "rt"))
    text, want = [], []
    for gi, (toks, result) in enumerate(games):
        # the moves in UCI through the SAN matcher the parser itself uses is circular; use the oracle to resolve SAN instead:
        b = ch.Board()
        ucis = []
        for tok in toks:
            cand = {s: u for u, s in eng.san_legal(b.fen())}
            ucis.append(cand[tok])
            b.push(ch.Move.from_uci(cand[tok]))
        want += _oracle_fens(ucis)
        parts = []
        for i, tok in enumerate(toks):
            if i % 2 == 0:
                parts.append(f"{i // 2 + 1}." + ("" if gi % 3 == 0 else " "))          # "1.e4" and "1. e4"
            t = tok
            if gi % 4 == 1 and i == 3:
                t = tok + "!?"
            if gi % 5 == 2 and tok.startswith("O-O"):
                t = tok.replace("O", "0")
            parts.append(t)
            if i == 2 and gi % 2 == 0:
                parts.append("{ a comment with 1. e4 inside }")
            if i == 4 and gi % 3 == 1:
                parts.append("( 3... a6 $2 ( 3... h6 ) 4. a3 )")
            if i == 5:
                parts.append("$14")
            if i == 6 and gi % 7 == 0:
                parts.append("; rest of line comment 5. Qh5\n")
        text.append(f'[Event "g{gi}"]\n[Result "{result}"]\n\n' + " ".join(parts) + f" {result}\n")
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".pgn", delete=False) as f:
        f.write("\n".join(text))
        path = f.name
    try:
        book = pgn_book.load_opening_book(path)
    finally:
        os.unlink(path)
    assert len(book) == len(want) == sum(min(20, len(t)) for t, _ in games)
    assert book == want
    capped = None
    with tempfile.NamedTemporaryFile("w", suffix=".pgn", delete=False) as f:
        f.write("\n".join(text))
        path = f.name
    try:
        capped = pgn_book.load_opening_book(path, max_positions=45)
    finally:
        os.unlink(path)
    assert 45 <= len(capped) < 45 + 20 and capped == want[: len(capped)]       # the cap is tested once per game (internal.py:49)


def test_setup_fen_header_and_bad_token():
    fen = "4k3/8/8/8/8/8/4P3/4K3 w - - 0 1"
    heads = {"SetUp": "1", "FEN": fen}
    got = pgn_book.mainline_fens(heads, "1. e4 Kd7 2. e5 Kc6 Zz9 3. e6", 20)
    assert got == _oracle_fens(["e2e4", "e8d7", "e4e5", "d7c6"], start=fen)       # stops at the unreadable token
    # the FEN header is honoured with or without SetUp (python-chess's Game.board()), and "e8Q" reads as "e8=Q"
    assert pgn_book.mainline_fens({"FEN": fen}, "1. e4", 20) == _oracle_fens(["e2e4"], start=fen)
    pfen = "8/4P1k1/8/8/8/8/8/4K3 w - - 0 1"
    assert pgn_book.mainline_fens({"FEN": pfen}, "1. e8Q Kf6", 20) == _oracle_fens(["e7e8q", "g7f6"], start=pfen)
    assert pgn_book.mainline_fens({"FEN": pfen}, "1. e8=N+ Kf8", 20) == _oracle_fens(["e7e8n", "g7f8"], start=pfen)
    assert pgn_book.mainline_fens({}, "1. e4 e5 2. Nf3 *  3. Nc3", 20) == _oracle_fens(["e2e4", "e7e5", "g1f3"])
    assert pgn_book.mainline_fens({}, "1. e4 e5 2. Nf3 Nc6", 3) == _oracle_fens(["e2e4", "e7e5", "g1f3"])


def test_fen_after_matches_the_oracle_incl_en_passant_and_errors():
    b = ch.Board()
    seq = ["e2e4", "a7a6", "e4e5", "d7d5", "e5d6", "e7d6", "g1f3", "b8c6", "f1b5", "c8d7", "e1g1"]
    for i in range(len(seq) + 1):
        bb = ch.Board()
        for u in seq[:i]:
            bb.push(ch.Move.from_uci(u))
        assert eng.fen_after(ch.START_FEN, seq[:i]) == bb.fen(), i
    # a double push next to an enemy pawn that is pinned: the ep square is NOT printed (no legal capture)
    fen = "8/8/8/8/k2p3R/8/4P3/4K3 w - - 0 1"
    bb = ch.Board(fen); bb.push(ch.Move.from_uci("e2e4"))
    assert eng.fen_after(fen, ["e2e4"]) == bb.fen() and " - " in bb.fen()
    with pytest.raises(ValueError):
        eng.fen_after(ch.START_FEN, ["e2e5"])
