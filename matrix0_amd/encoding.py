"""Device-side azchess/encoding.py (reference file:line in each docstring), FEN-addressed.

The reference takes python-chess Board objects; a caller that keeps python-chess passes
``board.fen()`` (and ``move.uci()``).  Everything is computed by the HIP kernels behind
m0_encode_fens (one wave per position); there is no CPU implementation in this package."""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence, Tuple

import numpy as np

from . import _lib
from .engine import _bind, move_to_uci

POLICY_SHAPE = (8, 8, 73)
LEGACY_POLICY_SIZE = 4672


def encode_fens(fens: Sequence[str], device_index: int = 0, want_moves: bool = True):
    """Batched: planes f32 [n,19,8,8], mask bool [n,4672], per-position (ucis, indices) in
    legal_moves order."""
    L = _bind()
    n = len(fens)
    arr = (C.c_char_p * n)(*[f.encode() for f in fens])
    planes = np.empty((n, 19, 8, 8), np.float32)
    mask = np.empty((n, 4672), np.uint8)
    nl = np.empty((n,), np.int32)
    mv = np.empty((n, 256), np.uint16)
    idx = np.empty((n, 256), np.int32)
    _lib.check(L.m0_encode_fens(int(device_index), arr, n, planes.ctypes.data_as(C.c_void_p), mask.ctypes.data_as(C.c_void_p),
                                nl.ctypes.data_as(C.c_void_p), mv.ctypes.data_as(C.c_void_p), idx.ctypes.data_as(C.c_void_p)),
               "m0_encode_fens")
    moves = None
    if want_moves:
        moves = [([move_to_uci(int(m)) for m in mv[i, : nl[i]]], idx[i, : nl[i]].tolist()) for i in range(n)]
    return planes, mask.astype(bool), moves


def encode_board(fen: str) -> np.ndarray:
    """encode_board, encoding.py:11-46 -> float32 [19,8,8]."""
    return encode_fens([fen], want_moves=False)[0][0]


def move_to_index(fen: str, uci: str) -> int:
    """move_to_index, encoding.py:114-150; ValueError for an illegal move (:120-121)."""
    out = C.c_int32(0)
    _lib.check(_bind().m0_move_to_index_fen(0, fen.encode(), uci.encode(), C.byref(out)), "move_to_index")
    return int(out.value)


class MoveEncoder:
    """MoveEncoder, encoding.py:153-253 (encode_move / get_legal_actions)."""

    def encode_move(self, fen: str, uci: str) -> int:
        return move_to_index(fen, uci)

    def decode_move(self, fen: str, action_idx: int) -> str:
        """encoding.py:174-229 -> UCI string ('0000' = Move.null())."""
        if not (0 <= int(action_idx) < 4672):
            raise ValueError("action_idx out of range")
        buf = C.create_string_buffer(8)
        _lib.check(_bind().m0_decode_move_fen(0, fen.encode(), int(action_idx), buf), "decode_move")
        return buf.value.decode()

    def get_legal_actions(self, fen: str) -> np.ndarray:
        return encode_fens([fen], want_moves=False)[1][0]


def build_horizontal_flip_permutation() -> np.ndarray:
    """encoding.py:310-347: E<->W, NE<->NW, SE<->SW per step; knight pairs; under-promo left/right."""
    perm = list(range(73))
    for step in range(7):
        for a, b in ((2, 3), (4, 5), (6, 7)):
            perm[a * 7 + step], perm[b * 7 + step] = perm[b * 7 + step], perm[a * 7 + step]
    for off in (0, 2, 4, 6):
        perm[56 + off], perm[57 + off] = perm[57 + off], perm[56 + off]
    for blk in (64, 67, 70):
        perm[blk + 1], perm[blk + 2] = perm[blk + 2], perm[blk + 1]
    return np.ascontiguousarray(perm, dtype=np.int64)


def build_rotate180_permutation() -> np.ndarray:
    """encoding.py:350-386: N<->S, E<->W, NE<->SW, NW<->SE per step; knight 180 pairs; under-promo l/r."""
    perm = list(range(73))
    for step in range(7):
        for a, b in ((0, 1), (2, 3), (4, 7), (5, 6)):
            perm[a * 7 + step], perm[b * 7 + step] = perm[b * 7 + step], perm[a * 7 + step]
    for a, b in ((56, 63), (57, 62), (58, 61), (59, 60)):
        perm[a], perm[b] = perm[b], perm[a]
    for blk in (64, 67, 70):
        perm[blk + 1], perm[blk + 2] = perm[blk + 2], perm[blk + 1]
    return np.ascontiguousarray(perm, dtype=np.int64)
