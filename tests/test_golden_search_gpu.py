"""The HIP search kernels compared DIRECTLY with traces of the real reference code (tests/golden/ref_mcts.json.gz, produced by
running /root/reference/azchess/mcts.py: see tools/gen_golden_mcts.py) -- no oracle in between.  Same counter streams, same
bit-reproducible evaluator (tests/hash_net.py) fed through the split-step C-ABI.  The reference's transposition table is
either patched out (tree-only: the engine's default structure) or left on (the engine's compat.tt_merge).  Its run() never
hands _select an in-flight dict, so ref_mcts.json.gz is compared with virtual_loss_active=0; ref_mcts_vl.json.gz holds the
traces in which it does (the reference's own virtual-loss lines executed) and is compared with virtual_loss_active=1, the
mode bench.py times.

Integer results (move order, policy indices, visit counts, simulations) must be identical; float32 priors within 1e-6
(the tolerance of the reference's own tests/test_mcts_logits.py); float64 q / root value within 1e-9."""
import numpy as np
import pytest

from tests.golden_ref import load_json, uci
from tests.hash_net import HashNet

pytestmark = pytest.mark.gpu
G = load_json("ref_mcts.json.gz")
FENS, BASE = G["fens"], G["base_mcts"]


def _engine(mcts, seed, L, compat=None, sims=96, vl=False):
    from matrix0_amd import engine as eng
    # num_simulations sizes the node arena (one search of new children): give it the search length
    cfg = eng.selfplay_cfg_from_dict({"seed": seed, "mcts": dict(mcts, inference_batch_size=L), "selfplay": {"num_simulations": sims}},
                                     concurrent_games=1, virtual_loss_active=vl, compat=compat)
    return eng.SelfplayEngine(None, cfg)


def _search(e, net):
    for _ in range(100000):
        planes = e.search_select()
        lg, v = net.infer_np(planes) if planes.shape[0] else (np.zeros((0, 4672), np.float32), np.zeros((0,), np.float32))
        e.search_expand(lg, v)
        r = e.search_result(0)
        if r["finished"]:
            return r
    raise AssertionError("search did not finish")


def _compare(res, want, prior_tol=1e-6):
    assert res["moves"] == [uci(c) for c in want["moves"]]
    assert res["idx"].tolist() == want["idx"]
    assert res["n"].tolist() == want["n"], (res["n"].tolist(), want["n"])
    np.testing.assert_allclose(res["prior"], want["prior"], rtol=0, atol=prior_tol)
    np.testing.assert_allclose(res["q"], want["q"], rtol=0, atol=1e-9)
    assert res["root_n"] == want["root_n"]
    assert abs(res["root_q"] - want["root_q"]) < 1e-9


@pytest.mark.parametrize("tt", ["off", "on"])
def test_whole_searches_match_reference_traces(tt):
    """MCTS.run: 96-simulation searches on 10 positions (sharp and flat policies, with and without Dirichlet noise), a
    300-simulation search at the reference's batch of 96, full-softmax mode, both cpuct schedules, two 1600-simulation
    searches (BASELINE configs[4] search length), playout cap, value_from_white, MCTS._prune_children (top-K, minimum prior,
    both), and the in-process-model branch whose non-root priors are raw legal logits over their sum
    (Node._expand_with_legal_priors).  tt="on": the reference's transposition table untouched (search graph is a DAG),
    against the engine's position table (compat.tt_merge)."""
    n = 0
    for case in G["runs"]:
        if case["tt"] != tt or case["repeats"] != 1:
            continue
        mcts = dict(BASE, **case["mcts_extra"])
        want = case["results"][0]
        e = _engine(mcts, case["seed"], case["L"], sims=want["sims"],
                    compat={"tt_merge": tt == "on", "raw_legal_priors": bool(case.get("model_path", False))})
        net = HashNet(**case["net"])
        e.search_begin(0, FENS[case["fen"]], want["sims"], case["dirichlet"], case["uid"])
        res = _search(e, net)
        _compare(res, want)
        assert net.calls == case["evals"], case["name"]        # same number of network evaluations as the reference made
        pi = np.zeros(4672, np.float32)
        tot = int(res["n"].sum())
        for i, k in zip(res["idx"], res["n"]):
            pi[i] = np.float32(int(k) / tot)                      # _policy_from_root, mcts.py:828-837
        nz = np.nonzero(pi)[0]
        assert nz.tolist() == want["pi_idx"] and [float(pi[j]) for j in nz] == want["pi_val"]
        e.close()
        n += 1
    assert n >= 30


@pytest.mark.parametrize("tt", ["off", "on"])
def test_whole_searches_with_virtual_loss_match_reference_traces(tt):
    """bench.py's search mode (virtual_loss_active = 1) against the reference's OWN virtual-loss lines: tests/golden/
    ref_mcts_vl.json.gz holds MCTS.run traces in which _collect_leaf_position hands _select the batch's in-flight dict
    (mcts.py:851, 889-890, 922-923; tools/gen_golden_mcts.py::VLOn).  96 ... 1 600 simulations, 16 / 32 / 96 leaves per batch
    (incl. the bench shape: 800 simulations in batches of 96), virtual_loss 0 / 0.3 / 1 / 3, terminal leaves inside a batch,
    pruning.  Identical visit counts, evaluation counts and policy targets."""
    GV = load_json("ref_mcts_vl.json.gz")
    assert GV["fens"] == FENS
    n = 0
    for case in GV["runs"]:
        if case["tt"] != tt:
            continue
        mcts = dict(GV["base_mcts"], **case["mcts_extra"])
        want = case["results"][0]
        e = _engine(mcts, case["seed"], case["L"], sims=want["sims"], vl=True, compat={"tt_merge": tt == "on"})
        net = HashNet(**case["net"])
        e.search_begin(0, FENS[case["fen"]], want["sims"], case["dirichlet"], case["uid"])
        res = _search(e, net)
        _compare(res, want)
        assert net.calls == case["evals"], case["name"]
        pi = np.zeros(4672, np.float32)
        tot = int(res["n"].sum())
        for i, k in zip(res["idx"], res["n"]):
            pi[i] = np.float32(int(k) / tot)
        nz = np.nonzero(pi)[0]
        assert nz.tolist() == want["pi_idx"] and [float(pi[j]) for j in nz] == want["pi_val"]
        e.close()
        n += 1
    assert n >= 22


def test_reused_root_is_evaluated_again_unless_cached():
    """compat.root_reinfer (mcts.py:359-371, LRUCache mcts.py:44-59): a root taken over from the previous search costs one
    more evaluation, except when the position was a reused root before (nn_cache hit).  Same visit counts either way; the
    evaluation count follows the oracle, which is pinned to the reference on this (same_board_x3 in ref_mcts.json.gz)."""
    from oracle import chess_py as ch
    from oracle import mcts_ref as ref
    mcts = dict(BASE)
    for reinfer in (False, True):
        e = _engine(mcts, 77, 8, sims=48, compat={"root_reinfer": reinfer})
        net = HashNet(seed=12, sharp=8.0)
        o_net = HashNet(seed=12, sharp=8.0)
        o = ref.MCTS(ref.MCTSConfig.from_dict(dict(mcts, inference_batch_size=8, use_tt=False, virtual_loss_active=False,
                                                   numerics="engine")), o_net.infer_np, seed=77, game=5)
        b = ch.Board(FENS[1])
        e.search_begin(0, FENS[1], 48, True, 5)
        for ply in range(4):
            res = _search(e, net)
            vc, _, _ = o.run(b, num_simulations=48, ply=ply)
            assert res["n"].tolist() == list(vc.values())
            slot = int(np.argmax(res["n"]))
            mv = ch.Move.from_uci(res["moves"][slot])
            o.note_move_played(mv)
            b.push(mv)
            if ply < 3:
                e.search_advance(0, slot, 48, True)
        # the oracle always re-evaluates a reused root (reference behaviour); the engine only with the switch
        assert net.calls == (o_net.calls if reinfer else o_net.calls - 3), (reinfer, net.calls, o_net.calls)
        e.close()


def test_expand_priors_match_reference_traces():
    """Node._expand (mcts.py:135-225): legal-only and full softmax, entropy noise on and off, flat / sharp / very sharp
    logits; for non-finite logits the reference dies with UnboundLocalError where its branch says 'uniform', which is what
    the kernel computes."""
    seed = G["expand"]["seed"]
    for c in G["expand"]["cases"]:
        mcts = dict(BASE, legal_softmax=c["legal_only"], enable_entropy_noise=c["noise"])
        e = _engine(mcts, seed, 4)
        net = HashNet(seed=c["net_seed"], sharp=c["sharp"], poison=c["poison"])
        e.search_begin(0, FENS[c["fen"]], 1, False, c["uid"])
        res = _search(e, net)
        if c["raised"]:
            np.testing.assert_allclose(res["prior"], 1.0 / len(res["prior"]), rtol=1e-6)
        else:
            assert res["moves"] == [uci(m) for m in c["moves"]] and res["idx"].tolist() == c["idx"]
            np.testing.assert_allclose(res["prior"], c["prior"], rtol=0, atol=1e-6)
        e.close()


def test_dirichlet_matches_reference_traces():
    """_add_dirichlet (mcts.py:955-992) for alpha below and above 1: priors after the root noise."""
    seed = G["dirichlet"]["seed"]
    for c in G["dirichlet"]["cases"]:
        mcts = dict(BASE, dirichlet_alpha=c["alpha"], dirichlet_frac=c["frac"], enable_entropy_noise=False, legal_softmax=True)
        e = _engine(mcts, seed, 4)
        e.search_begin(0, FENS[c["fen"]], 1, True, c["uid"])
        res = _search(e, HashNet(seed=9, sharp=8.0))
        np.testing.assert_allclose(res["prior"], c["after"], rtol=0, atol=1e-6)       # `before` is float32-rounded in both
        np.testing.assert_allclose(res["prior"] - np.array(c["before"]) * (1 - c["frac"]),
                                   np.array(c["after"]) - np.array(c["before"]) * (1 - c["frac"]), rtol=0, atol=2e-7)
        e.close()
