"""Evaluation matches on the GPU (azchess/arena.py:59-126): two networks in one engine, every search
evaluated by the network of the side to move, games replayed through the oracle's rules."""
import numpy as np
import pytest

from oracle import arena_ref
from oracle import chess_py as ch
from oracle import mcts_ref as ref
from oracle import net_ref

pytestmark = pytest.mark.gpu

NET = dict(planes=19, channels=32, blocks=2, attention_heads=2, policy_size=4672, norm="group", activation="silu",
           preact=True, policy_factor_rank=0, self_supervised=False)
CFG = {"seed": 7,
       "mcts": {"cpuct": 2.5, "cpuct_start": 3.0, "cpuct_end": 2.0, "cpuct_plies": 40, "dirichlet_plies": 30,
                "dirichlet_frac": 0.25, "selection_jitter": 0.0, "fpu_reduction": 0.1, "draw_penalty": -0.05,
                "legal_softmax": True, "inference_batch_size": 8, "playout_random_frac": 0.0},
       "selfplay": {"selection_jitter": 0.0},
       "eval": {"max_moves": 40},
       "draw": {"min_plies": 30, "window": 8, "min_unique": 4, "halfmove_cap": 100}}


def _backends(bias_b=None):
    from matrix0_amd.backend import M0Backend
    sd_a = net_ref.random_state_dict(NET, seed=1)
    sd_b = net_ref.random_state_dict(NET, seed=2)
    if bias_b:
        for idx in bias_b:
            sd_b["policy_fc.bias"][idx] += 60.0            # B "knows" exactly one move per colour
    return M0Backend.from_state_dict(NET, sd_a), M0Backend.from_state_dict(NET, sd_b)


def _replay(rec, max_moves):
    b = ch.Board()
    moves = []
    for i, u in enumerate(rec["played"]):
        m = ch.Move.from_uci(u)
        assert not b.is_game_over(claim_draw=True), (i, u)            # the loop condition held before every move
        assert m in b.legal_moves, (i, u)
        b.push(m)
        moves.append(m)
    return b, moves


def test_match_games_are_legal_and_scored_like_the_reference():
    from matrix0_amd import arena
    a, b = _backends()
    score = arena.play_match(a, b, 6, CFG, seed=11, num_sims=24, temp=1.0, temp_plies=6, concurrent_games=4,
                             leaves_per_step=8)
    st = arena.last_match_stats
    recs = st["records"]
    assert len(recs) == 6 and sorted(r["game_index"] for r in recs) == list(range(6))
    assert st["a_wins"] + st["b_wins"] + st["draws"] == 6
    assert score == st["a_wins"] + 0.5 * st["draws"] == st["score"]
    assert st["win_rate"] == score / 6 and (st["wilson_low"], st["wilson_high"]) == arena_ref.wilson_interval(score / 6, 6)
    for r in recs:
        board, moves = _replay(r, 40)
        n = len(r["played"])
        assert r["moves"] == n <= 40
        over = board.is_game_over(claim_draw=True)
        assert over or n == 40 or ref.should_adjudicate_draw(board, moves, CFG["draw"])
        res = board.result(claim_draw=True) if over else "1/2-1/2"
        assert r["result_str"] == res
        assert r["score_a"] == arena_ref.game_score(res, r["game_index"] % 2 == 0)
    a.close(); b.close()


def test_each_search_uses_the_network_of_the_side_to_move():
    """B's policy head is rigged to put all prior mass on a2a3 (as White) and a7a6 (as Black); A's is not.  With no
    temperature the most visited root move follows: B opens 1. a3 in the odd games and answers ... a6 in the even ones."""
    from matrix0_amd import arena
    start = ch.Board()
    i_w = ch.move_to_index(start, ch.Move.from_uci("a2a3"))
    a, b = _backends(bias_b=[i_w])
    arena.play_match(a, b, 4, CFG, seed=3, num_sims=32, temp=0.0, temp_plies=0, max_moves_override=1, concurrent_games=4,
                     leaves_per_step=8)
    first = {r["game_index"]: r["played"][0] for r in arena.last_match_stats["records"]}
    assert first[1] == "a2a3" and first[3] == "a2a3"            # B is White in odd games
    assert first[0] == first[2] and first[0] != "a2a3"          # A is White in even games, deterministic argmax
    a.close(); b.close()
    # and as Black: rig the reply to 1. <A's first move>
    bb = ch.Board(); bb.push(ch.Move.from_uci(first[0]))
    i_b = ch.move_to_index(bb, ch.Move.from_uci("a7a6"))
    a, b = _backends(bias_b=[i_w, i_b])
    arena.play_match(a, b, 2, CFG, seed=3, num_sims=32, temp=0.0, temp_plies=0, max_moves_override=2, concurrent_games=2,
                     leaves_per_step=8)
    recs = {r["game_index"]: r["played"] for r in arena.last_match_stats["records"]}
    assert recs[0][1] == "a7a6"                                 # game 0: A white, B black
    assert recs[1][0] == "a2a3"                                 # game 1: B white
    a.close(); b.close()


def test_match_is_deterministic_for_a_seed_and_pgn_out(tmp_path):
    from matrix0_amd import arena
    a, b = _backends()
    kw = dict(seed=5, num_sims=16, temp=1.0, temp_plies=10, max_moves_override=12, concurrent_games=3, leaves_per_step=8)
    arena.play_match(a, b, 3, CFG, pgn_out=str(tmp_path / "pgn"), **kw)
    g1 = {r["game_index"]: r["played"] for r in arena.last_match_stats["records"]}
    arena.play_match(a, b, 3, CFG, **kw)
    g2 = {r["game_index"]: r["played"] for r in arena.last_match_stats["records"]}
    assert g1 == g2
    files = sorted(p.name for p in (tmp_path / "pgn").iterdir())
    assert files == ["game_0000.pgn", "game_0001.pgn", "game_0002.pgn"]
    txt = (tmp_path / "pgn" / "game_0001.pgn").read_text()
    assert '[White "B"]' in txt and '[Black "A"]' in txt and "1. " in txt
    a.close(); b.close()


def test_match_with_per_side_tables_plays_legal_games():
    """`engine.compat.tt_merge` in play_match: one position table per side for the whole game (arena.py:157-158), node arenas sized by
    play_match itself.  (Move-for-move parity of this mode with the untouched reference: tests/test_golden_selfplay_gpu.py.)"""
    from matrix0_amd import arena
    a, b = _backends()
    cfg = dict(CFG, engine={"compat": {"tt_merge": True}})
    score = arena.play_match(a, b, 4, cfg, seed=21, num_sims=24, temp=1.0, temp_plies=4, max_moves_override=16, concurrent_games=4,
                             leaves_per_step=8)
    st = arena.last_match_stats
    assert len(st["records"]) == 4 and st["a_wins"] + st["b_wins"] + st["draws"] == 4 and score == st["score"]
    for r in st["records"]:
        board, moves = _replay(r, 16)
        assert 1 <= len(r["played"]) <= 16
    # more evaluations than the fresh-tree match of the same games: roots found in a side's table are evaluated again
    arena.play_match(a, b, 4, CFG, seed=21, num_sims=24, temp=1.0, temp_plies=4, max_moves_override=16, concurrent_games=4, leaves_per_step=8)
    assert st["evals"] != arena.last_match_stats["evals"]
    a.close(); b.close()


def test_per_side_tables_start_over_when_their_arena_is_full():
    """A side's arena half that cannot take another search's nodes drops its table and starts from a fresh root (the reference
    prunes an over-full table too): with the smallest arena the games still run to their end and stay legal."""
    from matrix0_amd import arena
    a, b = _backends()
    cfg = dict(CFG, engine={"compat": {"tt_merge": True}, "arena_nodes": 4096})
    arena.play_match(a, b, 2, cfg, seed=33, num_sims=32, temp=1.0, temp_plies=6, max_moves_override=30, concurrent_games=2,
                     leaves_per_step=8)
    st = arena.last_match_stats
    assert len(st["records"]) == 2
    for r in st["records"]:
        _replay(r, 30)
        assert len(r["played"]) >= 20                    # no search ended without visits (that would end the game early)
    a.close(); b.close()


def test_step_right_after_create_and_eval_cache_ignored_by_the_match_engine():
    """(a) Network B's forward runs on network A's stream: its workspace (allocated and cleared at the first forward, on B's own
    stream) must be ready whatever stream asks -- two wide networks created and stepped at once give the same match as a second
    run on warm workspaces.  (b) `eval_cache` is ignored by a match engine: two networks alternate in a game slot and the cache
    key carries no network id, so the switch must neither serve hits nor change a game."""
    from matrix0_amd import arena, engine as eng
    from matrix0_amd.backend import M0Backend
    wide = dict(NET, channels=320, blocks=2, attention_heads=20, policy_factor_rank=128)

    def match(a, b, cache):
        c = arena.arena_cfg_from_dict(CFG, games=4, num_sims=24, max_moves=8, temp=1.0, temp_plies=4, concurrent_games=4,
                                      leaves_per_step=8, seed=9)
        c.eval_cache = 1 if cache else 0
        e = eng.ArenaEngine(a, b, c)
        recs = {}
        while e.running():
            e.step(8)
            while (r := e.poll()) is not None:
                recs[r["game_index"]] = (r["played"], float(r["result"]))
        st = e.stats()
        e.close()
        return recs, st

    a = M0Backend.from_state_dict(wide, net_ref.random_state_dict(wide, seed=1))
    b = M0Backend.from_state_dict(wide, net_ref.random_state_dict(wide, seed=2))
    cold, st_cold = match(a, b, False)            # first forwards of both networks: workspaces allocated inside the first step
    warm, st_warm = match(a, b, False)
    cached, st_c = match(a, b, True)
    a.close(); b.close()
    assert sorted(cold) == [0, 1, 2, 3]
    assert cold == warm and st_cold["evals"] == st_warm["evals"]
    assert cached == warm and st_c["evals"] == st_warm["evals"] and st_c["evals_cached"] == 0
