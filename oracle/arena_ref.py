"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the decision functions of the reference's evaluation arena
(azchess/arena.py, azchess/elo.py).  Imported by tests/ only; the product (matrix0_amd/arena.py, csrc/host_rules.h)
must never import it.

  arena_choose_move   arena.py:73-106   temperature sampling over log-visits / most-visited move
  game_score          arena.py:110-126  score of the game from A's point of view
  wilson_interval     arena.py:272-278
  expected_score / update_elo   elo.py:10-22
"""
from __future__ import annotations

import math
from typing import Sequence, Tuple

import numpy as np


def arena_choose_move(visits: Sequence[int], temp: float, ply: int, temp_plies: int, u: float) -> int:
    """Index into the visit list (dict insertion order = legal-move order).  `u` replaces the uniform that
    np.random.choice(len(moves), p=probs) draws (inverse CDF, as numpy implements it)."""
    vis = np.array(list(visits), dtype=np.float32)
    if temp > 1e-3 and ply < temp_plies:
        logits = np.log(vis + np.float32(1e-8)) / np.float32(max(temp, 1e-3))
        probs = np.exp(logits - np.max(logits))
        s = probs.sum()
        if s <= 0 or not np.isfinite(s):
            return int(np.argmax(vis))
        probs = probs / s
        cdf = np.cumsum(probs.astype(np.float64))
        cdf /= cdf[-1]
        return int(min(np.searchsorted(cdf, u, side="right"), len(vis) - 1))
    # max(visits.items(), key=lambda kv: kv[1])[0]: the FIRST maximum in insertion order
    best = 0
    for i in range(1, len(vis)):
        if vis[i] > vis[best]:
            best = i
    return best


def game_score(result: str, a_is_white: bool) -> float:
    if result == "1-0":
        return 1.0 if a_is_white else 0.0
    if result == "0-1":
        return 0.0 if a_is_white else 1.0
    return 0.5


def wilson_interval(p: float, n: int, z: float = 1.96) -> Tuple[float, float]:
    if n == 0:
        return 0.0, 0.0
    denom = 1 + z * z / n
    center = (p + (z * z) / (2 * n)) / denom
    half = (z * ((p * (1 - p) / n) + (z * z) / (4 * n * n)) ** 0.5) / denom
    return max(0.0, center - half), min(1.0, center + half)


def expected_score(ra: float, rb: float) -> float:
    return 1.0 / (1.0 + 10.0 ** ((rb - ra) / 400.0))


def update_elo(ra: float, rb: float, sa: float, k: float = 20.0) -> Tuple[float, float]:
    delta = k * (sa - expected_score(ra, rb))
    return ra + delta, rb - delta
