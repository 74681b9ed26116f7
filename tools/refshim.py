"""Build-container harness that lets the REAL reference modules (azchess/encoding.py, mcts.py, draw.py,
selfplay/internal.py, arena.py ...) run without python-chess: registers a `chess` module backed by the oracle's
rules engine (oracle/chess_py.py + chess_oracle.c, the App. A.5 surface of SURVEY.md) and a synthetic `azchess`
package whose __path__ points at /root/reference/azchess (azchess/__init__.py, which drags in the orchestrator, is
skipped).  Used only by tools/gen_golden_*.py; never shipped to the GPU box (no /root/reference there).

Nothing of the reference is copied: its files are imported from where they lie and executed to produce golden
vectors (tests/golden/ref_*), which are data."""
from __future__ import annotations

import contextlib
import math
import os
import random as _random
import sys
import types

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from oracle import chess_py as ch           # noqa: E402
from oracle import mcts_ref as oref         # noqa: E402  (Stream / derive_seed: the counter-based draws)

REF = os.environ.get("M0_REFERENCE", "/root/reference")


class Piece:
    __slots__ = ("piece_type", "color")

    def __init__(self, piece_type, color):
        self.piece_type, self.color = piece_type, color


class Board(ch.Board):
    """oracle Board + the few python-chess members only the reference's callers touch."""

    def copy(self, stack=True):
        return Board(_handle=self._l.o_game_copy(self._g))

    def __deepcopy__(self, memo):
        return self.copy()

    @property
    def turn(self):
        return bool(self._p.contents.turn)

    @turn.setter
    def turn(self, v):
        self._p.contents.turn = int(bool(v))

    def piece_at(self, sq):
        pc = self.piece_code_at(sq)
        return None if pc == 0 else Piece((pc - 1) % 6 + 1, pc <= 6)


def install():
    if "chess" in sys.modules and getattr(sys.modules["chess"], "_m0_shim", False):
        return sys.modules["chess"]
    m = types.ModuleType("chess")
    m._m0_shim = True
    m.WHITE, m.BLACK = True, False
    m.PAWN, m.KNIGHT, m.BISHOP, m.ROOK, m.QUEEN, m.KING = range(1, 7)
    m.SQUARES = list(range(64))
    for s in range(64):
        setattr(m, ch.square_name(s).upper(), s)
    m.STARTING_FEN = ch.START_FEN
    m.square, m.square_rank, m.square_file, m.square_name = ch.square, ch.square_rank, ch.square_file, ch.square_name
    m.Move, m.Board, m.Piece = ch.Move, Board, Piece
    m.Bitboard = int
    m.Color = bool
    m.Square = int
    m.PieceType = int
    for sub in ("pgn", "polyglot", "syzygy", "engine", "svg"):
        sm = types.ModuleType("chess." + sub)
        setattr(m, sub, sm)
        sys.modules["chess." + sub] = sm
    sys.modules["chess"] = m
    pkg = types.ModuleType("azchess")
    pkg.__path__ = [os.path.join(REF, "azchess")]
    sys.modules["azchess"] = pkg
    return m


# ---- injected randomness: the reference's draws re-routed to the counter streams of oracle/mcts_ref.py ----
class Streams:
    """One game's four streams, keyed like csrc/host_rules.h (seed, game uid, purpose)."""

    def __init__(self, seed, game):
        self.jitter = oref.Stream(oref.derive_seed(seed, game, oref.PURPOSE_JITTER))
        self.noise = oref.Stream(oref.derive_seed(seed, game, oref.PURPOSE_NOISE))
        self.dirichlet = oref.Stream(oref.derive_seed(seed, game, oref.PURPOSE_DIRICHLET))
        self.game = oref.Stream(oref.derive_seed(seed, game, oref.PURPOSE_GAME))


@contextlib.contextmanager
def injected(streams: Streams):
    """random.random -> jitter stream (mcts.py:893-897); np.random.normal -> noise stream (mcts.py:180);
    np.random.dirichlet -> gamma draws from the dirichlet stream (mcts.py:966); random.randint / random.choice /
    np.random.choice -> the game stream (mcts.py:384, internal.py:376, 733)."""
    saved = (_random.random, _random.randint, _random.choice, np.random.normal, np.random.dirichlet, np.random.choice)

    def normal(loc, scale, size=None):
        shape = (size,) if isinstance(size, int) else tuple(size)
        n = int(np.prod(shape))
        z = np.array([streams.noise.normal() for _ in range(n)], dtype=np.float64)
        return (loc + scale * z).reshape(shape)

    def dirichlet(alpha):
        g = [streams.dirichlet.gamma(float(a)) for a in alpha]
        s = sum(g)
        return np.array([x / s for x in g], dtype=np.float64)

    def randint(lo, hi):
        u = streams.game.next()
        return lo + min(hi - lo, int(u * (hi - lo + 1)))

    def choice_py(seq):
        u = streams.game.next()
        return seq[min(len(seq) - 1, int(u * len(seq)))]

    def choice_np(a, p=None):
        u = streams.game.next()
        if p is None:
            n = a if isinstance(a, int) else len(a)
            k = min(n - 1, int(u * n))
            return k if isinstance(a, int) else a[k]
        cdf = np.cumsum(np.asarray(p, dtype=np.float64))
        cdf /= cdf[-1]
        k = int(min(len(cdf) - 1, np.searchsorted(cdf, u, side="right")))
        return k if isinstance(a, int) else a[k]

    _random.random = streams.jitter.next
    _random.randint = randint
    _random.choice = choice_py
    np.random.normal = normal
    np.random.dirichlet = dirichlet
    np.random.choice = choice_np
    try:
        yield
    finally:
        (_random.random, _random.randint, _random.choice, np.random.normal, np.random.dirichlet, np.random.choice) = saved
