"""SAN generation (m0_san_legal_fen / m0_san_game, PGN output of arena games) against the reference's own PGN data:
125 games written by python-chess under data/eval_games (fixture: tests/golden/eval_games_san.json.gz, extracted by
tools/make_fixtures.py).  Every SAN token of every game must be produced for exactly one legal move, and replaying
the check / mate suffixes must agree with the oracle's board after the move."""
import gzip
import json
import os

import numpy as np

from matrix0_amd import engine as eng
from oracle import chess_py as ch

GOLD = os.path.join(os.path.dirname(__file__), "golden", "eval_games_san.json.gz")


def _uci_to_raw(u):
    f = (ord(u[0]) - 97) + 8 * (int(u[1]) - 1)
    t = (ord(u[2]) - 97) + 8 * (int(u[3]) - 1)
    p = " nbrq".index(u[4]) if len(u) > 4 else 0
    return f | (t << 6) | (p << 12)


def test_san_tokens_of_the_reference_pgns_replay():
    games = json.load(gzip.open(GOLD, "rt"))
    assert len(games) == 125
    n_tokens = 0
    kinds = set()
    for toks, result in games:
        b = ch.Board()
        raw = []
        for tok in toks:
            cand = eng.san_legal(b.fen())
            assert [u for u, _ in cand] == [m.uci() for m in b.legal_moves]      # same moves, same order as the oracle
            hits = [u for u, s in cand if s == tok]
            assert len(hits) == 1, (tok, b.fen(), cand)
            raw.append(_uci_to_raw(hits[0]))
            b.push(ch.Move.from_uci(hits[0]))
            n_tokens += 1
            kinds.update(c for c in tok if c in "x+#=O")
            # suffix semantics on the oracle's board ('#' mate, '+' check, none: no check)
            assert b.is_checkmate() == tok.endswith("#")
            assert (b.is_check() and not b.is_checkmate()) == tok.endswith("+")
        # (the data set's Result headers are not trustworthy -- game 0 ends "Be6# 1-0" with Black mating -- so the
        #  outcome is not compared)
        # the whole game's movetext is the tokens with move numbers
        text = eng.san_game(np.array(raw, np.uint16))
        assert [t for t in text.split() if not t.endswith(".")] == toks
    assert n_tokens == 7875
    assert kinds >= set("x+#=O")            # captures, checks, mates, promotions and castling all occur in the data
