"""Self-play logic pinned by outputs of the REAL reference (tools/gen_golden_selfplay.py):
  ref_selfplay.json.gz   sample_move_from_counts / game_result / should_adjudicate_draw / arena move choice cases
  ref_worker_*.npz       whole games written by the reference's selfplay_worker (NPZ + queue message + per-ply trace)
CPU: the oracle (oracle/mcts_ref.py decision functions, oracle/selfplay_ref.py game loop) and the product's host rules
through the C-ABI (m0_sample_move_index, m0_rules_probe, m0_arena_choose_move: no GPU needed)."""
import glob
import json
import os

import numpy as np
import pytest

from oracle import chess_py as ch
from oracle import mcts_ref as ref
from oracle import selfplay_ref
from tests.golden_ref import GOLD, load_json, load_npz, uci
from tests.hash_net import HashNet

S = load_json("ref_selfplay.json.gz")
WORKERS = sorted(os.path.basename(p)[len("ref_worker_"):-4] for p in glob.glob(os.path.join(GOLD, "ref_worker_*.npz")))


def _game_u(seed, uid, k=0):
    return ref.Stream(ref.derive_seed(seed, uid, ref.PURPOSE_GAME), ctr=k)


def test_sample_move_matches_reference():
    from matrix0_amd import engine as eng
    seed = S["sample_move"]["seed"]
    kinds = set()
    for c in S["sample_move"]["cases"]:
        draws = ref.sample_move_draws(c["visits"], c["temperature"])
        assert int(draws) == c["draws"]                                 # consumes a draw exactly when the reference does
        u = _game_u(seed, c["uid"]).next() if draws else 0.0
        assert ref.sample_move_index(c["visits"], c["temperature"], u) == c["chosen"], c
        assert eng.sample_move_index(c["visits"], c["temperature"], u) == c["chosen"], c      # host_rules.h via the C-ABI
        kinds.add((all(v == 0 for v in c["visits"]), c["temperature"] < 1e-3))
    assert len(kinds) >= 3


def test_game_result_matches_reference():
    from matrix0_amd import engine as eng
    cfg = eng.selfplay_cfg_from_dict({}, concurrent_games=1)
    for c in S["game_result"]:
        b = ch.Board(c["fen"])
        for u in c["moves"]:
            b.push(ch.Move.from_uci(u))
        assert ref.game_result(b) == c["game_result"], c["name"]
        assert b.is_game_over() == c["game_over"] and b.is_game_over(claim_draw=True) == c["game_over_claim"]
        assert b.result(claim_draw=True) == c["result_claim"]
        r = eng.rules_probe(cfg, c["fen"], c["moves"])
        assert r["game_over"] == c["game_over"] and r["game_over_claim"] == c["game_over_claim"], c["name"]
        if c["game_over_claim"]:
            assert r["result"] == c["game_result"], c["name"]


def test_adjudicate_draw_matches_reference():
    from matrix0_amd import engine as eng
    cfgs = S["adjudicate_draw"]["cfgs"]
    ecfgs = [eng.selfplay_cfg_from_dict({"draw": c}, concurrent_games=1) for c in cfgs]
    n_true = 0
    for g in S["adjudicate_draw"]["games"]:
        b = ch.Board(g["start"])
        moves = []
        for ply in range(len(g["flags"][0])):
            for ci, c in enumerate(cfgs):
                want = g["flags"][ci][ply]
                assert ref.should_adjudicate_draw(b, moves, c) == want, (g["start"], ply, ci)
                if ply % 3 == 0 or want:
                    assert eng.rules_probe(ecfgs[ci], g["start"], g["moves"][:ply])["adjudicate_draw"] == want, (g["start"], ply, ci)
                n_true += int(want)
            if ply < len(g["moves"]):
                m = ch.Move.from_uci(g["moves"][ply])
                moves.append(m)
                b.push(m)
    assert n_true > 100


def test_arena_choice_matches_reference():
    from matrix0_amd import engine as eng
    from oracle import arena_ref
    seed = S["arena_choice"]["seed"]
    for g in S["arena_choice"]["games"]:
        st = _game_u(seed, g["uid"])
        for ply, (vis, want) in enumerate(zip(g["visits"], g["chosen"])):
            sampling = g["temp"] > 1e-3 and ply < g["temp_plies"]
            u = st.next() if sampling else 0.0
            assert eng.arena_choose_move(vis, g["temp"], ply, g["temp_plies"], u) == want, (g["uid"], ply)
            assert arena_ref.arena_choose_move(vis, g["temp"], ply, g["temp_plies"], u) == want, (g["uid"], ply)
        assert st.ctr == g["draws"]


def _load_worker(name):
    z = load_npz(f"ref_worker_{name}.npz")
    g = {k: z[k] for k in z.files}
    g["meta"] = json.loads(str(g["meta_json"]))
    return g


def worker_cfg(meta):
    return {"seed": meta["seed"], "selfplay": meta["selfplay"], "mcts": meta["mcts"], "draw": {}, "openings": {}}


def check_game_against_golden(out, g, *, tol_v=1e-12, evals=None):
    """`out`: a finished game in the NPZ vocabulary (s, pi, z, legal_mask, moves, result, resigned, resigner, draw,
    avg_policy_entropy, avg_sims, trace{visits,chosen,v,sims}); `g`: the golden the reference worker wrote."""
    meta, msg = g["meta"], g["meta"]["message"]
    T = int(g["meta_moves"][0])
    assert out["moves"] == T == msg["moves"]
    off = np.concatenate([[0], np.cumsum(g["trace_nchild"])])
    for t in range(T):
        assert list(out["trace"]["visits"][t]) == g["trace_visits"][off[t]:off[t + 1]].tolist(), f"visit counts at ply {t}"
        assert out["trace"]["sims"][t] == int(g["trace_sims"][t])
        assert out["trace"]["chosen"][t] == int(g["trace_chosen"][t]), f"move choice at ply {t}"
        assert abs(out["trace"]["v"][t] - float(g["trace_v"][t])) <= tol_v
    assert np.array_equal(out["s"], g["s"])
    assert np.array_equal(out["pi"], g["pi"])
    assert np.array_equal(out["legal_mask"], g["legal_mask"])
    np.testing.assert_allclose(out["z"], g["z"], rtol=0, atol=max(tol_v, 1e-7))
    assert abs(out["result"] - float(g["meta_result"][0])) <= max(tol_v, 1e-7)
    assert bool(out["resigned"]) == bool(g["meta_resigned"][0]) == msg["resigned"]
    assert out["resigner"] == msg["resigner"]
    assert bool(out["draw"]) == bool(g["meta_draw"][0]) == msg["draw"]
    assert abs(out["avg_policy_entropy"] - msg["avg_policy_entropy"]) < 1e-6
    assert abs(out["avg_sims"] - msg["avg_sims"]) < 1e-6
    if evals is not None:
        assert evals == meta["evals"]


@pytest.mark.parametrize("name", WORKERS)
def test_oracle_replays_reference_worker_games(name):
    """oracle/selfplay_ref.play_game against the game the reference's selfplay_worker produced with the same streams and
    evaluator: every visit count, move, plane, target, mask and message field."""
    g = _load_worker(name)
    meta = g["meta"]
    net = HashNet(**meta["net"])
    out = selfplay_ref.play_game(worker_cfg(meta), net.infer_np, meta["seed"], 0, book=meta["book"] or None,
                                 use_tt=False, tree_reuse=False, virtual_loss_active=bool(meta.get("virtual_loss_active", False)),
                                 numerics="reference")
    check_game_against_golden(out, g, evals=net.calls)
    assert out["streams"] == meta["draws"]


def test_worker_goldens_cover_every_ending():
    ends = {}
    for name in WORKERS:
        g = _load_worker(name)
        ends[name] = (int(g["meta_resigned"][0]), float(g["meta_result"][0]), int(g["meta_moves"][0]))
    assert any(r for r, _, _ in ends.values())                          # a resignation
    assert any(z == 1.0 and not r for r, z, _ in ends.values())         # White mates
    assert any(z == -1.0 and not r for r, z, _ in ends.values())        # Black mates
    assert any(z == 0.0 for _, z, _ in ends.values())                   # a draw by rule
    assert any(z not in (0.0, 1.0, -1.0) for _, z, _ in ends.values())  # cut by length / heuristic: z = last root value
    assert "ssl_piece" in _load_worker("lengthcap")


# ---- whole arena games of the reference's _arena_run_one_game with real searches (tools/gen_golden_selfplay.py::gen_arena) ----
ARENA = load_json("ref_arena.json.gz")


@pytest.mark.parametrize("gi", range(len(ARENA["games"])))
def test_oracle_replays_reference_arena_games(gi):
    """oracle/arena_ref.play_game against the game the reference's own loop played, with its transposition table patched out
    (fresh root per search) and untouched (one table per side for the whole game): two evaluators, the side to move's
    searcher at every ply, visit counts, move choice (sampled / most visited), draw adjudication, length cap, result,
    evaluation counts per network and stream positions."""
    from oracle import arena_ref
    g = ARENA["games"][gi]
    na, nb = HashNet(**g["net_a"]), HashNet(**g["net_b"])
    out = arena_ref.play_game(g["uid"], g["mcts"], na.infer_np, nb.infer_np, ARENA["seed"], sims=g["sims"], max_moves=g["max_moves"],
                              temp=g["temp"], temp_plies=g["temp_plies"], draw_cfg=g["draw"], use_tt=g["tt"] == "on")
    assert out["plies"] == g["plies"] and out["result"] == g["result"] and out["score"] == g["score"]
    assert out["final_fen"] == g["final_fen"]
    for t, (got, want) in enumerate(zip(out["trace"], g["trace"])):
        assert got["side"] == want["side"] and got["fen"] == want["fen"], t
        assert got["moves"] == want["moves"] and got["visits"] == want["visits"], t
        assert abs(got["root_q"] - want["root_q"]) < 1e-12, t
    assert [t["chosen"] for t in out["trace"]] == g["chosen"]
    assert (out["evals_a"], out["evals_b"]) == (g["evals_a"], g["evals_b"])
    assert out["draws"] == g["draws"]


def test_arena_goldens_cover_both_colours_sampling_argmax_and_adjudication():
    gs = ARENA["games"]
    assert {g["uid"] % 2 for g in gs} == {0, 1}
    assert any(g["temp"] <= 1e-3 for g in gs) and any(g["temp"] > 1e-3 and g["temp_plies"] > 0 for g in gs)
    assert any(g["plies"] < g["max_moves"] for g in gs) and any(g["plies"] == g["max_moves"] for g in gs)
    assert all({t["side"] for t in g["trace"]} == {"A", "B"} for g in gs)
    assert {g["tt"] for g in gs} == {"on", "off"}
    # the decisive branch of arena.py:112-120: A mates as White, B mates as White, B mates as Black
    assert {(g["result"], g["score"], g["uid"] % 2) for g in gs} >= {("1-0", 1.0, 0), ("1-0", 0.0, 1), ("0-1", 0.0, 0), ("1/2-1/2", 0.5, 0)}
    # the table changes the games: more evaluations (a root found in the table is evaluated again) and other visit counts
    off = {g["uid"]: g for g in gs if g["tt"] == "off"}
    on = {g["uid"]: g for g in gs if g["tt"] == "on"}
    assert all(on[u]["evals_a"] >= off[u]["evals_a"] for u in off) and sum(on[u]["evals_a"] > off[u]["evals_a"] for u in off) >= 4
    assert any([t["visits"] for t in on[u]["trace"]] != [t["visits"] for t in off[u]["trace"]] for u in off)
