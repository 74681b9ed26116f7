"""Arena decision functions: the product's host code (m0_arena_choose_move, matrix0_amd/arena.py) against
the oracle restatement of azchess/arena.py:73-126, 272-278 and elo.py:10-22, and the PGN writer's format."""
import numpy as np

from matrix0_amd import arena, engine as eng
from oracle import arena_ref as ref


def test_move_choice_matches_the_oracle():
    rng = np.random.default_rng(3)
    n_sampled = 0
    for trial in range(3000):
        k = int(rng.integers(1, 40))
        vis = rng.integers(0, 60, size=k).astype(np.int32)
        if trial % 7 == 0:
            vis[:] = int(rng.integers(0, 3))            # all equal (incl. all zero): first index / uniform softmax
        if trial % 11 == 0:
            vis[rng.integers(0, k)] = 800
        temp = float(rng.choice([0.0, 1e-4, 0.25, 1.0, 2.0]))
        temp_plies = int(rng.choice([0, 4, 30]))
        ply = int(rng.integers(0, 40))
        u = float(rng.random())
        want = ref.arena_choose_move(vis, temp, ply, temp_plies, u)
        got = eng.arena_choose_move(vis, temp, ply, temp_plies, u)
        assert got == want, (vis, temp, ply, temp_plies, u)
        n_sampled += int(temp > 1e-3 and ply < temp_plies)
    assert n_sampled > 300
    # most visited = FIRST maximum in move order (max(visits.items(), key=...))
    assert eng.arena_choose_move([3, 9, 9, 1], 0.0, 0, 0, 0.5) == 1


def test_wilson_elo_and_scores():
    for p, n in [(0.5, 100), (0.0, 10), (1.0, 10), (0.55, 200), (0.3, 0), (0.731, 37)]:
        assert np.allclose(arena.wilson_interval(p, n), ref.wilson_interval(p, n), rtol=0, atol=1e-15)
    lo, hi = arena.wilson_interval(0.5, 100)
    assert abs(lo - 0.4038) < 5e-4 and abs(hi - 0.5962) < 5e-4           # textbook value
    for ra, rb, sa in [(1500, 1500, 0.5), (1500, 1500, 0.75), (1620, 1480, 0.4)]:
        assert np.allclose(arena.update_elo(ra, rb, sa), ref.update_elo(ra, rb, sa))
    assert arena.update_elo(1500, 1500, 0.75) == (1505.0, 1495.0)
    for res in ("1-0", "0-1", "1/2-1/2"):
        for w in (True, False):
            assert arena.game_score(res, w) == ref.game_score(res, w)
    assert arena.result_string(1.0, True) == "1-0" and arena.result_string(-1.0, True) == "0-1"
    assert arena.result_string(0.0, True) == "1/2-1/2" and arena.result_string(1.0, False) == "1/2-1/2"


def test_pgn_file(tmp_path):
    # 1. f3 e5 2. g4 Qh4#
    def raw(u):
        return (ord(u[0]) - 97) + 8 * (int(u[1]) - 1) | ((ord(u[2]) - 97) + 8 * (int(u[3]) - 1)) << 6
    moves = np.array([raw("f2f3"), raw("e7e5"), raw("g2g4"), raw("d8h4")], np.uint16)
    path = arena.save_pgn(moves, "0-1", {"White": "A", "Black": "B", "Round": 3}, str(tmp_path), 7)
    assert path.endswith("game_0007.pgn")
    txt = open(path).read().splitlines()
    assert txt[:7] == ['[Event "?"]', '[Site "?"]', '[Date "????.??.??"]', '[Round "3"]', '[White "A"]', '[Black "B"]',
                       '[Result "0-1"]']
    assert txt[7] == "" and txt[8] == "1. f3 e5 2. g4 Qh4# 0-1"
