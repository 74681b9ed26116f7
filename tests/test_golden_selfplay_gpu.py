"""WHOLE GAMES on the HIP engine compared directly with the games the reference's own selfplay_worker produced
(tests/golden/ref_worker_*.npz, tools/gen_golden_selfplay.py): same seed, same counter streams, same evaluator (tests/hash_net.py
behind the external-evaluator step of the C-ABI), reference mode = a fresh tree per move, no virtual loss -- except `vl_on_batch96`, played by the reference with its own
virtual-loss lines executed (tools/gen_golden_mcts.py::VLOn) and by the engine with virtual_loss_active=1 -- (its transposition
table patched out: with the table on the reference cannot get past move 2, see ref_mcts.json.gz::tt_across_moves).

Everything a shard contains must agree: planes, policy targets, legal masks, value targets, result, resignation, draw flag,
move count, entropy / simulation averages, SSL target maps -- plus the visit counts and the chosen move of every ply."""
import numpy as np
import pytest

from tests.hash_net import HashNet
from tests.test_golden_selfplay import WORKERS, _load_worker, worker_cfg

pytestmark = pytest.mark.gpu


def _play(meta, net):
    from matrix0_amd import engine as eng
    from matrix0_amd.selfplay import detect_value_from_white
    cfg_dict = worker_cfg(meta)
    vfw = bool(cfg_dict["mcts"].get("value_from_white", False)) or detect_value_from_white(net)     # internal.py:245-247
    cfg_dict["mcts"] = dict(cfg_dict["mcts"], value_from_white=vfw)
    cfg = eng.selfplay_cfg_from_dict(cfg_dict, concurrent_games=1, total_games=1, first_game_index=0,
                                     virtual_loss_active=bool(meta.get("virtual_loss_active", False)), ssl_targets=meta["ssl"], compat={"fresh_tree_per_move": True})
    e = eng.SelfplayEngine(None, cfg)
    if meta["book"]:
        e.set_openings(meta["book"])
    rec = None
    for _ in range(200000):
        if not e.running():
            break
        planes = e.ext_select()
        lg, v = net.infer_np(planes) if planes.shape[0] else (np.zeros((0, 4672), np.float32), np.zeros((0,), np.float32))
        e.ext_expand(lg, v)
        rec = rec or e.poll()
    rec = rec or e.poll()
    assert rec is not None, "the game did not finish"
    st = e.stats()
    e.close()
    return rec, st


@pytest.mark.parametrize("name", WORKERS)
def test_engine_plays_the_reference_workers_game(name):
    g = _load_worker(name)
    meta, msg = g["meta"], g["meta"]["message"]
    net = HashNet(**meta["net"])
    rec, st = _play(meta, net)
    T = int(g["meta_moves"][0])
    assert rec["moves"] == T
    if T:
        assert np.array_equal(rec["s"], g["s"]), "planes"
        assert np.array_equal(rec["legal_mask"], g["legal_mask"]), "legal masks"
        # pi = child.n / total per ply: identical visit counts <=> identical float32 targets
        assert np.array_equal(rec["pi"], g["pi"]), "policy targets (visit counts)"
        np.testing.assert_allclose(rec["search_values"], g["trace_v"].astype(np.float32), rtol=0, atol=1e-6)
        np.testing.assert_allclose(rec["z"], g["z"], rtol=0, atol=1e-6)
    assert abs(rec["result"] - float(g["meta_result"][0])) < 1e-6
    assert rec["resigned"] == bool(g["meta_resigned"][0]) and rec["resigner"] == msg["resigner"]
    assert rec["draw"] == bool(g["meta_draw"][0])
    assert abs(rec["avg_policy_entropy"] - float(g["meta_avg_policy_entropy"][0])) < 1e-5
    assert abs(rec["avg_sims"] - float(g["meta_avg_sims"][0])) < 1e-4
    # the moves actually played after the opening plies = the reference's sampled moves (the last choice of a resigned
    # game is made but not played)
    n_open = len(rec["played"]) - (T - (1 if rec["resigned"] else 0))
    from tests.golden_ref import uci
    want_moves = [uci(int(c)) for c in g["trace_chosen_move"]][: T - (1 if rec["resigned"] else 0)]
    assert rec["played"][n_open:] == want_moves
    assert net.calls == meta["evals"], "network evaluations"
    if meta["ssl"]:
        for task in ("piece", "threat", "pin", "fork", "control"):
            assert np.array_equal(rec["ssl"][task], g[f"ssl_{task}"]), task


# ---- whole arena games: the reference's own _arena_run_one_game with real searches (tests/golden/ref_arena.json.gz) ----
from tests.golden_ref import load_json, uci as _uci      # noqa: E402

ARENA = load_json("ref_arena.json.gz")


@pytest.mark.parametrize("gi", range(len(ARENA["games"])))
def test_match_engine_plays_the_reference_arena_game(gi):
    """m0_arena_create_ext + the two-evaluator step of the C-ABI against the game the reference's arena loop played
    (arena.py:59-126; one MCTS object per side; its table patched out, and untouched): the side to move's evaluator at every ply, visit counts of
    every search (as the float32 policy target n / total), root values, the sampled / most visited move, adjudication and
    length cap, the result, and the number of evaluations each network was asked for."""
    from matrix0_amd import engine as eng
    g = ARENA["games"][gi]
    cfg_dict = {"seed": ARENA["seed"], "mcts": dict(g["mcts"]), "draw": dict(g["draw"]),
                "selfplay": {"num_simulations": g["sims"], "max_game_len": g["max_moves"], "opening_random_plies": 0}}
    # tt = "on": the untouched reference, one table per side for the whole game = the match engine with compat.tt_merge (the
    # node arena must then hold everything a side creates in a game)
    cfg = eng.selfplay_cfg_from_dict(cfg_dict, concurrent_games=1, total_games=1, first_game_index=g["uid"],
                                     virtual_loss_active=False, record_games=True, compat={"tt_merge": g["tt"] == "on"},
                                     arena_nodes=400000 if g["tt"] == "on" else 0)
    cfg.arena_temp, cfg.arena_temp_plies = float(g["temp"]), int(g["temp_plies"])
    e = eng.ArenaExtEngine(cfg)
    na, nb = HashNet(**g["net_a"]), HashNet(**g["net_b"])
    z0 = (np.zeros((0, 4672), np.float32), np.zeros((0,), np.float32))
    rec = None
    for _ in range(100000):
        if not e.running():
            break
        pa, pb = e.arena_ext_select()
        la, va = na.infer_np(pa) if pa.shape[0] else z0
        lb, vb = nb.infer_np(pb) if pb.shape[0] else z0
        e.arena_ext_expand(la, va, lb, vb)
        rec = rec or e.poll()
    rec = rec or e.poll()
    e.close()
    assert rec is not None and rec["game_index"] == g["uid"]
    T = g["plies"]
    assert rec["moves"] == T and len(rec["played"]) == T
    legal0 = [t["moves"][k] for t, k in zip(g["trace"], g["chosen"])]
    assert rec["played"] == [_uci(c) for c in legal0]
    for t, want in enumerate(g["trace"]):
        tot = float(sum(want["visits"]))
        pi = np.zeros(4672, np.float32)
        for i, n in zip(want["idx"], want["visits"]):
            pi[i] = np.float32(n / tot)
        assert np.array_equal(rec["pi"][t], pi), f"visit counts at ply {t} (side {want['side']})"
        assert abs(float(rec["search_values"][t]) - want["root_q"]) < 1e-6, t
    # arena.py:112-126: a finished game is scored by board.result(claim_draw=True), an unfinished / adjudicated one is 1/2-1/2;
    # the engine reports the result from White's point of view (+1 / 0 / -1) -- three of the goldens end in mate
    assert rec["result"] == {"1-0": 1.0, "0-1": -1.0, "1/2-1/2": 0.0}[g["result"]]
    from matrix0_amd import arena as m0arena
    res_str = m0arena.result_string(float(rec["result"]), True)
    assert res_str == g["result"] and m0arena.game_score(res_str, g["uid"] % 2 == 0) == g["score"]
    assert (na.calls, nb.calls) == (g["evals_a"], g["evals_b"])
