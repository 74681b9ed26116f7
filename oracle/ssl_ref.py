"""ORACLE (test infrastructure): per-square restatement of the reference's SSL target generators
(azchess/ssl_algorithms.py:51-143 threats, 256-343 pins, 345-433 forks, 435-517 control, 545-566 pieces),
as the worker calls them: one position (batch of 1) per recorded ply (selfplay/internal.py:460-482).

All geometry is in TENSOR space (row 0 = rank 8), exactly as the reference computes it -- including its quirks
(SURVEY B-5): "white" pawns attack toward higher row index; the pin detector ANDs a map that is non-zero only on
the candidate square with one that is non-zero only on the next square, so the pin target is identically zero.
Pinned by tests/golden/ssl_targets.npz (outputs of the real module, tools/gen_golden_ssl.py)."""
from __future__ import annotations

import numpy as np

KNIGHT = [(-2, -1), (-2, 1), (-1, -2), (-1, 2), (1, -2), (1, 2), (2, -1), (2, 1)]
KING = [(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)]
DIAG = [(-1, -1), (-1, 1), (1, -1), (1, 1)]
ORTHO = [(-1, 0), (1, 0), (0, -1), (0, 1)]


def _on(r, c):
    return 0 <= r < 8 and 0 <= c < 8


def _attack_counts(pl, occ, color):
    """Number of (piece, ray) attacks of `color` (0 white planes 0-5, 1 black planes 6-11) on every square."""
    o = 0 if color == 0 else 6
    P, N, B, R, Q, K = (pl[o + i] for i in range(6))
    att = np.zeros((8, 8), np.int32)
    pdr = 1 if color == 0 else -1                      # _shift(wp,+1,+-1) / _shift(bp,-1,+-1)
    for r in range(8):
        for c in range(8):
            for dc in (-1, 1):
                if P[r, c] and _on(r + pdr, c + dc):
                    att[r + pdr, c + dc] += 1
            for dr, dc in KNIGHT:
                if N[r, c] and _on(r + dr, c + dc):
                    att[r + dr, c + dc] += 1
            for dr, dc in KING:
                if K[r, c] and _on(r + dr, c + dc):
                    att[r + dr, c + dc] += 1
            for dirs, on in ((DIAG, B[r, c] + Q[r, c]), (ORTHO, R[r, c] + Q[r, c])):
                if not on:
                    continue
                for dr, dc in dirs:
                    rr, cc = r + dr, c + dc
                    while _on(rr, cc):
                        att[rr, cc] += int(on)
                        if occ[rr, cc]:
                            break
                        rr += dr; cc += dc
    return att


def targets(planes: np.ndarray) -> dict:
    """planes f32 [19,8,8] -> {'piece' [13,8,8], 'threat','pin','fork','control' [8,8]} (integer valued)."""
    pl = (planes[:12] > 0).astype(np.int32)
    occ = pl.sum(axis=0) > 0
    stm_white = planes[12, 0, 0] > 0.5
    piece = np.zeros((13, 8, 8), np.int64)
    piece[:12] = pl
    piece[12] = ~occ
    wa, ba = _attack_counts(pl, occ, 0), _attack_counts(pl, occ, 1)
    threat = np.clip(ba if stm_white else wa, 0, 1)
    control = np.sign(wa - ba)
    pin = np.zeros((8, 8), np.int32)                   # see module docstring
    o, e = (0, 6) if stm_white else (6, 0)
    enemy = pl[e:e + 6].sum(axis=0) > 0
    own = {k: pl[o + i] for i, k in enumerate("PNBRQK")}
    count = np.zeros((8, 8), np.int32)
    for r in range(8):
        for c in range(8):
            if own["N"][r, c]:
                count[r, c] += sum(1 for dr, dc in KNIGHT if _on(r + dr, c + dc) and enemy[r + dr, c + dc])
            if own["K"][r, c]:
                count[r, c] += sum(1 for dr, dc in KING if _on(r + dr, c + dc) and enemy[r + dr, c + dc])
            for dirs, on in ((DIAG, own["B"][r, c]), (ORTHO, own["R"][r, c]), (DIAG, own["Q"][r, c]), (ORTHO, own["Q"][r, c])):
                if not on:
                    continue
                for dr, dc in dirs:
                    rr, cc = r + dr, c + dc
                    while _on(rr, cc):
                        if occ[rr, cc]:
                            if enemy[rr, cc]:
                                count[r, c] += 1
                            break
                        rr += dr; cc += dc
    tactical = (own["N"] + own["B"] + own["R"] + own["Q"] + own["K"]) > 0
    fork = ((count >= 2) & tactical).astype(np.int32)
    return {"piece": piece, "threat": threat.astype(np.int32), "pin": pin, "fork": fork, "control": control.astype(np.int32)}
