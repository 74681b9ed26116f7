"""ORACLE (test infrastructure): ctypes view of oracle/chess_oracle.c with a small
python-chess-like surface (the subset the reference's hot path calls).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libchess_oracle.so")

START_FEN = "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1"
WHITE, BLACK = True, False
PAWN, KNIGHT, BISHOP, ROOK, QUEEN, KING = range(1, 7)


def build():
    src = os.path.join(_HERE, "chess_oracle.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


class OMove(C.Structure):
    _fields_ = [("from_", C.c_uint8), ("to", C.c_uint8), ("promo", C.c_uint8)]


class OPos(C.Structure):
    _fields_ = [("sq", C.c_int8 * 64), ("turn", C.c_int8), ("castling", C.c_uint64), ("ep", C.c_int8),
                ("halfmove", C.c_int32), ("fullmove", C.c_int32)]


_L = None


def L():
    global _L
    if _L is None:
        l = C.CDLL(build())
        l.o_game_new.restype = C.c_void_p
        l.o_game_new.argtypes = [C.c_char_p]
        l.o_game_free.argtypes = [C.c_void_p]
        l.o_game_copy.restype = C.c_void_p
        l.o_game_copy.argtypes = [C.c_void_p]
        l.o_game_push.argtypes = [C.c_void_p, OMove]
        l.o_game_pop.argtypes = [C.c_void_p]
        l.o_game_pos.restype = C.POINTER(OPos)
        l.o_game_pos.argtypes = [C.c_void_p]
        l.o_game_len.argtypes = [C.c_void_p]
        l.o_game_is_repetition.argtypes = [C.c_void_p, C.c_int]
        l.o_game_can_claim_threefold.argtypes = [C.c_void_p]
        l.o_game_outcome.argtypes = [C.c_void_p, C.c_int]
        l.o_perft.restype = C.c_uint64
        l.o_perft.argtypes = [C.POINTER(OPos), C.c_int]
        l.o_clean_castling.restype = C.c_uint64
        for f in ("o_gen_legal", "o_in_check", "o_is_checkmate", "o_is_stalemate", "o_is_insufficient",
                  "o_is_seventyfive", "o_can_claim_fifty", "o_has_legal_ep", "o_any_legal"):
            getattr(l, f).restype = C.c_int
        l.o_sizeof_pos.restype = C.c_size_t
        assert l.o_sizeof_pos() == C.sizeof(OPos), (l.o_sizeof_pos(), C.sizeof(OPos))
        _L = l
    return _L


def square(file, rank):
    return rank * 8 + file


def square_rank(sq):
    return sq >> 3


def square_file(sq):
    return sq & 7


def square_name(sq):
    return "abcdefgh"[sq & 7] + str((sq >> 3) + 1)


class Move:
    __slots__ = ("from_square", "to_square", "promotion")

    def __init__(self, from_square, to_square, promotion=None):
        self.from_square = int(from_square)
        self.to_square = int(to_square)
        self.promotion = promotion if promotion else None

    def __eq__(self, o):
        return isinstance(o, Move) and (self.from_square, self.to_square, self.promotion) == \
            (o.from_square, o.to_square, o.promotion)

    def __hash__(self):
        return hash((self.from_square, self.to_square, self.promotion))

    def uci(self):
        if self.from_square == self.to_square == 0 and self.promotion is None:
            return "0000"
        s = square_name(self.from_square) + square_name(self.to_square)
        if self.promotion:
            s += "  nbrq"[self.promotion]
        return s

    __str__ = uci

    def __repr__(self):
        return f"Move({self.uci()})"

    @classmethod
    def from_uci(cls, u):
        f = (ord(u[0]) - 97) + 8 * (int(u[1]) - 1)
        t = (ord(u[2]) - 97) + 8 * (int(u[3]) - 1)
        p = "  nbrq".index(u[4]) if len(u) > 4 else None
        return cls(f, t, p)

    @classmethod
    def null(cls):
        return cls(0, 0, None)

    def _c(self):
        return OMove(self.from_square, self.to_square, self.promotion or 0)


class Board:
    """Subset of chess.Board backed by the C oracle (with move stack)."""

    def __init__(self, fen: str = START_FEN, _handle=None):
        self._l = L()
        self._g = _handle if _handle is not None else self._l.o_game_new(fen.encode())
        if not self._g:
            raise ValueError(f"bad FEN: {fen}")

    def __del__(self):
        try:
            if self._g:
                self._l.o_game_free(self._g)
                self._g = None
        except Exception:
            pass

    @property
    def _p(self):
        return self._l.o_game_pos(self._g)

    @property
    def turn(self):
        return bool(self._p.contents.turn)

    @property
    def halfmove_clock(self):
        return int(self._p.contents.halfmove)

    @property
    def fullmove_number(self):
        return int(self._p.contents.fullmove)

    @property
    def ep_square(self):
        e = int(self._p.contents.ep)
        return None if e < 0 else e

    def copy(self, stack=True):
        return Board(_handle=self._l.o_game_copy(self._g))

    @property
    def legal_moves(self):
        buf = (OMove * 256)()
        n = self._l.o_gen_legal(self._p, buf)
        return [Move(buf[i].from_, buf[i].to, buf[i].promo or None) for i in range(n)]

    def is_legal(self, m):
        return m in self.legal_moves

    def push(self, m):
        self._l.o_game_push(self._g, m._c())

    def pop(self):
        self._l.o_game_pop(self._g)

    def ply(self):
        return self._l.o_game_len(self._g)

    def piece_code_at(self, sq):
        return int(self._p.contents.sq[sq])

    def piece_type_at(self, sq):
        pc = self.piece_code_at(sq)
        return None if pc == 0 else (pc - 1) % 6 + 1

    def pieces(self, piece_type, color):
        code = piece_type + (0 if color else 6)
        return [s for s in range(64) if self._p.contents.sq[s] == code]

    def piece_map(self):
        return {s: int(self._p.contents.sq[s]) for s in range(64) if self._p.contents.sq[s]}

    def has_kingside_castling_rights(self, color):
        return bool(self._l.o_has_kingside(self._p, int(bool(color))))

    def has_queenside_castling_rights(self, color):
        return bool(self._l.o_has_queenside(self._p, int(bool(color))))

    def is_check(self):
        return bool(self._l.o_in_check(self._p))

    def is_checkmate(self):
        return bool(self._l.o_is_checkmate(self._p))

    def is_stalemate(self):
        return bool(self._l.o_is_stalemate(self._p))

    def is_insufficient_material(self):
        return bool(self._l.o_is_insufficient(self._p))

    def is_seventyfive_moves(self):
        return bool(self._l.o_is_seventyfive(self._p))

    def is_fivefold_repetition(self):
        return self.is_repetition(5)

    def is_repetition(self, count=3):
        return bool(self._l.o_game_is_repetition(self._g, int(count)))

    def can_claim_threefold_repetition(self):
        return bool(self._l.o_game_can_claim_threefold(self._g))

    def can_claim_fifty_moves(self):
        return bool(self._l.o_can_claim_fifty(self._p))

    def outcome_code(self, claim_draw=False):
        return int(self._l.o_game_outcome(self._g, int(claim_draw)))

    def is_game_over(self, claim_draw=False):
        return self.outcome_code(claim_draw) != 0

    def result(self, claim_draw=False):
        oc = self.outcome_code(claim_draw)
        if oc == 0:
            return "*"
        if oc == 1:
            return "0-1" if self.turn else "1-0"
        return "1/2-1/2"

    def is_capture(self, m):
        if self.piece_code_at(m.to_square):
            return True
        return self.piece_type_at(m.from_square) == PAWN and m.to_square == self.ep_square and \
            (m.from_square & 7) != (m.to_square & 7)

    def _transposition_key(self):
        buf = (C.c_uint8 * 68)()
        self._l.o_tkey(self._p, buf)
        return bytes(buf)

    def perft(self, depth):
        return int(self._l.o_perft(self._p, depth))

    def fen(self):
        p = self._p.contents
        rows = []
        for r in range(7, -1, -1):
            row, e = "", 0
            for f in range(8):
                pc = p.sq[r * 8 + f]
                if pc == 0:
                    e += 1
                else:
                    if e:
                        row += str(e); e = 0
                    row += "PNBRQKpnbrqk"[pc - 1]
            if e:
                row += str(e)
            rows.append(row)
        cr = self._l.o_clean_castling(self._p)
        cs = ("K" if cr >> 7 & 1 else "") + ("Q" if cr & 1 else "") + ("k" if cr >> 63 & 1 else "") + \
             ("q" if cr >> 56 & 1 else "")
        ep = square_name(p.ep) if self._l.o_has_legal_ep(self._p) else "-"
        return f"{'/'.join(rows)} {'w' if p.turn else 'b'} {cs or '-'} {ep} {p.halfmove} {p.fullmove}"


# ---- azchess/encoding.py surface ----
def encode_board(board: Board) -> np.ndarray:
    out = np.zeros((19, 8, 8), dtype=np.float32)
    board._l.o_encode_board(board._p, out.ctypes.data_as(C.c_void_p))
    return out


def move_to_index(board: Board, move: Move) -> int:
    if not board.is_legal(move):
        raise ValueError(f"Illegal move: {move}")
    idx = board._l.o_move_to_index(board._p, move._c())
    if idx < 0:
        raise ValueError(f"Illegal move: {move}")
    return int(idx)


def get_legal_actions(board: Board) -> np.ndarray:
    m = np.zeros(4672, dtype=np.uint8)
    board._l.o_legal_mask(board._p, m.ctypes.data_as(C.c_void_p))
    return m.astype(bool)


def legal_moves_with_indices(board: Board):
    buf = (OMove * 256)()
    idx = (C.c_int32 * 256)()
    n = board._l.o_legal_moves_idx(board._p, buf, idx)
    return [Move(buf[i].from_, buf[i].to, buf[i].promo or None) for i in range(n)], [int(idx[i]) for i in range(n)]


RAY_DIRS = ((1, 0), (-1, 0), (0, 1), (0, -1), (1, 1), (1, -1), (-1, 1), (-1, -1))
KNIGHT_DELTAS = ((-2, -1), (-2, 1), (-1, -2), (-1, 2), (1, -2), (1, 2), (2, -1), (2, 1))


def decode_move(board: Board, action_idx: int) -> Move:
    """MoveEncoder.decode_move, encoding.py:174-229: index -> Move with auto-queen on the last rank and the
    same-destination / same-origin legal fallbacks; Move.null() when nothing fits."""
    if not (0 <= action_idx < 4672):
        raise ValueError("action_idx out of range")
    from_sq, off = divmod(action_idx, 73)
    fr, ff = from_sq >> 3, from_sq & 7

    def mk(dr, df, steps=1, promo=None):
        tr, tf = fr + dr * steps, ff + df * steps
        if not (0 <= tr < 8 and 0 <= tf < 8):
            return Move.null()
        p = promo
        if promo is None and board.piece_type_at(from_sq) == PAWN and tr in (0, 7):
            p = QUEEN
        return Move(from_sq, tr * 8 + tf, p)

    if off < 56:
        dr, df = RAY_DIRS[off // 7]
        mv = mk(dr, df, off % 7 + 1)
    elif off < 64:
        dr, df = KNIGHT_DELTAS[off - 56]
        mv = mk(dr, df)
    else:
        u = off - 64
        dirs = ((1, 0), (1, -1), (1, 1)) if board.turn else ((-1, 0), (-1, 1), (-1, -1))
        dr, df = dirs[u % 3]
        mv = mk(dr, df, 1, (KNIGHT, BISHOP, ROOK)[u // 3])
    legal = board.legal_moves
    if mv in legal:
        return mv
    for lm in legal:
        if lm.from_square == from_sq and lm.to_square == mv.to_square:
            return lm
    for lm in legal:
        if lm.from_square == from_sq:
            return lm
    return Move.null()
