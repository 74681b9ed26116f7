"""The oracle (oracle/mcts_ref.py) pinned to traces of the REAL reference search code: tests/golden/ref_mcts.json.gz was
produced by running /root/reference/azchess/mcts.py itself (tools/gen_golden_mcts.py) with the shared counter streams
and the bit-reproducible HashNet evaluator.  Integer results exact; float64 statistics to 1e-12; float32 priors to 1e-7
(torch.softmax may differ by an ulp between CPU models)."""
import numpy as np
import pytest

from oracle import chess_py as ch
from oracle import mcts_ref as ref
from tests.golden_ref import load_json
from tests.hash_net import HashNet

G = load_json("ref_mcts.json.gz")
FENS = G["fens"]
BASE = G["base_mcts"]


def _mv(code):
    return ch.Move(code & 63, (code >> 6) & 63, ((code >> 12) & 7) or None)


def test_cpuct_schedules_match_reference():
    for tab in G["cpuct_at"]["tables"]:
        o = ref.MCTS(ref.MCTSConfig.from_dict(dict(BASE, **tab["cfg"])), None)
        got = [o.cpuct_at(p) for p in G["cpuct_at"]["plies"]]
        assert got == tab["values"]


def test_backpropagate_matches_reference():
    for case in G["backpropagate"]:
        nodes = []
        for n, w in case["before"]:
            nd = ref.Node(); nd.n = n; nd.w = w; nd.q = w / n if n else 0.0
            nodes.append(nd)
        for v in case["values"]:
            ref.MCTS.backpropagate(nodes, v)
        assert [[nd.n, nd.w, nd.q] for nd in nodes] == case["after"]


@pytest.mark.parametrize("numerics,tol", [("reference", 1e-7), ("engine", 1e-6)])
def test_expand_priors_match_reference(numerics, tol):
    seed = G["expand"]["seed"]
    n_raised = 0
    for c in G["expand"]["cases"]:
        b = ch.Board(FENS[c["fen"]])
        moves, idxs = ch.legal_moves_with_indices(b)
        lg, _ = HashNet(seed=c["net_seed"], sharp=c["sharp"], poison=c["poison"]).infer_np(ch.encode_board(b))
        noise = ref.Stream(ref.derive_seed(seed, c["uid"], ref.PURPOSE_NOISE))
        pri = ref.legal_priors(lg[0], idxs, c["legal_only"], c["noise"], noise, numerics=numerics)
        if c["raised"]:
            # the reference's non-finite branch (mcts.py:147-149) dies with UnboundLocalError at :214; the oracle and the
            # engine implement the branch as written (uniform priors)
            n_raised += 1
            assert c["raised"] == "UnboundLocalError" and not c["finite"]
            np.testing.assert_allclose(pri, 1.0 / len(moves), rtol=1e-6)
            continue
        assert [m.from_square | (m.to_square << 6) | ((m.promotion or 0) << 12) for m in moves] == c["moves"]
        assert idxs == c["idx"]
        np.testing.assert_allclose(pri.astype(np.float64), c["prior"], rtol=0, atol=tol)
        assert noise.ctr == c["noise_draws"]
    assert n_raised > 0


def test_dirichlet_matches_reference():
    for c in G["dirichlet"]["cases"]:
        o = ref.MCTS(ref.MCTSConfig.from_dict(dict(BASE, dirichlet_alpha=c["alpha"], dirichlet_frac=c["frac"])), None,
                     seed=G["dirichlet"]["seed"], game=c["uid"])
        root = ref.Node()
        for i, p in enumerate(c["before"]):
            root.children[i] = ref.Node(prior=p)
        o.add_dirichlet(root)
        np.testing.assert_allclose([x.prior for x in root.children.values()], c["after"], rtol=1e-12, atol=0)
        o.add_dirichlet(root)
        np.testing.assert_allclose([x.prior for x in root.children.values()], c["after2"], rtol=1e-12, atol=0)
        assert o.dirichlet.ctr == c["draws"]


def test_select_one_level_matches_reference():
    for c in G["select_one_level"]["cases"]:
        b = ch.Board(FENS[c["fen"]])
        o = ref.MCTS(ref.MCTSConfig.from_dict(dict(BASE, selection_jitter=c["jitter"], fpu_reduction=c["fpu_reduction"],
                                                   use_tt=False)), None, seed=G["select_one_level"]["seed"], game=c["uid"])
        parent = ref.Node(); parent.n = c["parent_n"]; parent.q = c["parent_q"]; parent.expanded = True
        for m, (n, q, p) in zip(b.legal_moves, c["children"]):
            k = ref.Node(prior=p, move=m, parent=parent); k.n = n; k.q = q
            parent.children[m] = k
        node, path, _ = o.select(b.copy(), parent, None)
        assert list(parent.children.values()).index(path[1]) == c["chosen"]
        assert o.jitter.ctr == c["draws"]


class _Snap:
    pass


def _snap(root):
    """Statistics of a root and its children as they are now (the same Node objects are updated by later runs)."""
    s = _Snap()
    s.n = root.n
    s.children = {}
    for m, c in root.children.items():
        k = _Snap()
        k.move, k.move_idx, k.n, k.prior, k.q = c.move, c.move_idx, c.n, c.prior, c.q
        s.children[m] = k
    return s


def _oracle_run(case, numerics="reference"):
    cfg = ref.MCTSConfig.from_dict(dict(BASE, **case["mcts_extra"], inference_batch_size=case["L"], use_tt=(case["tt"] == "on"),
                                        virtual_loss_active=False, numerics=numerics,
                                        raw_legal_priors=bool(case.get("model_path", False))))
    net = HashNet(**case["net"])
    o = ref.MCTS(cfg, net.infer_np, seed=case["seed"], game=case["uid"])
    game = ref.Stream(ref.derive_seed(case["seed"], case["uid"], ref.PURPOSE_GAME))
    b = ch.Board(FENS[case["fen"]])
    outs = []
    for r in range(case["repeats"]):
        sims_override = None
        if cfg.playout_random_frac > 0:
            sims_override = ref.playout_cap(case["sims"], cfg.playout_random_frac, game.next())
        vc, pi, rq = o.run(b, num_simulations=case["sims"], ply=(0 if case["dirichlet"] else 1000), sims_override=sims_override)
        outs.append((_snap(o._last_root), vc, pi, rq, o._last_sims_run))
    return o, net, outs


def _check_root(root, vc, pi, rq, sims, want, prior_tol):
    kids = list(root.children.values())
    assert [c.move.from_square | (c.move.to_square << 6) | ((c.move.promotion or 0) << 12) for c in kids] == want["moves"]
    assert [c.move_idx for c in kids] == want["idx"]
    assert [c.n for c in kids] == want["n"]
    np.testing.assert_allclose([c.prior for c in kids], want["prior"], rtol=0, atol=prior_tol)
    np.testing.assert_allclose([c.q for c in kids], want["q"], rtol=0, atol=1e-12)
    assert root.n == want["root_n"] and sims == want["sims"]
    assert abs(rq - want["root_q"]) < 1e-12
    nz = np.nonzero(pi)[0]
    assert nz.tolist() == want["pi_idx"]
    assert [float(pi[j]) for j in nz] == want["pi_val"]


@pytest.mark.parametrize("tt", ["off", "on"])
def test_whole_searches_match_reference(tt):
    """MCTS.run of the reference (tree-only with the table patched out, and untouched with the table on) against the
    oracle: identical visit counts, policy target, evaluation count and stream positions."""
    n = 0
    for case in G["runs"]:
        if case["tt"] != tt:
            continue
        o, net, outs = _oracle_run(case)
        for (root, vc, pi, rq, sims), want in zip(outs, case["results"]):
            _check_root(root, vc, pi, rq, sims, want, 1e-7)
        assert net.calls == case["evals"], case["name"]
        assert (o.jitter.ctr, o.noise.ctr, o.dirichlet.ctr) == (case["draws"]["jitter"], case["draws"]["noise"], case["draws"]["dirichlet"])
        if tt == "on":
            assert len(o.tt) == case["tt_entries"]
        n += 1
    assert n >= 25


def test_engine_numerics_reproduce_the_reference_trajectories():
    """The float64 softmax the HIP expand kernel uses (numerics="engine") is within an ulp of torch's float32 softmax;
    on the golden searches it must lead to the very same visit counts (tree-only cases = what the GPU test compares)."""
    for case in G["runs"]:
        if case["tt"] != "off":
            continue
        o, net, outs = _oracle_run(case, numerics="engine")
        for (root, vc, pi, rq, sims), want in zip(outs, case["results"]):
            _check_root(root, vc, pi, rq, sims, want, 1e-6)


def test_table_across_moves_reference_raises_oracle_shows_zero_visits():
    """Recorded behaviour of the reference: the second run() of a game on one MCTS object raises 'zero visits' when the
    table is on (see tools/gen_golden_mcts.py).  The oracle's use_tt mode shows the same state (all root children n == 0)."""
    tr = G["tt_across_moves"]
    assert tr[0]["total_visits"] == 64 and tr[1].get("raised") == "RuntimeError" and "zero visits" in tr[1]["message"]
    o = ref.MCTS(ref.MCTSConfig.from_dict(dict(BASE, inference_batch_size=8, use_tt=True)), HashNet(seed=3, sharp=8.0).infer_np,
                 seed=1, game=0)
    b = ch.Board()
    vc, _, _ = o.run(b, num_simulations=64, ply=0)
    assert sum(vc.values()) == 64
    b.push(max(vc, key=vc.get))
    vc, _, _ = o.run(b, num_simulations=64, ply=1)
    assert sum(vc.values()) == 0 and o._last_root.n == 64


# ---- virtual loss: the reference's own lines (mcts.py:851, 889-890, 922-923), executed by tools/gen_golden_mcts.py::VLOn ----
GV = load_json("ref_mcts_vl.json.gz")


def test_select_one_level_with_inflight_counts_matches_reference():
    """MCTS._select(board, root, inflight_counts=d) of the reference on hand-set statistics and a hand-set dict: which child
    wins under the penalty `count * virtual_loss`, and that the chosen edge's count goes up by one."""
    assert GV["fens"] == FENS
    n_pen = 0
    for c in GV["select_one_level"]["cases"]:
        b = ch.Board(FENS[c["fen"]])
        o = ref.MCTS(ref.MCTSConfig.from_dict(dict(BASE, selection_jitter=c["jitter"], fpu_reduction=c["fpu_reduction"],
                                                   virtual_loss=c["virtual_loss"], use_tt=False)), None,
                     seed=GV["select_one_level"]["seed"], game=c["uid"])
        parent = ref.Node(); parent.n = c["parent_n"]; parent.q = c["parent_q"]; parent.expanded = True
        kids = []
        for m, (n, q, p) in zip(b.legal_moves, c["children"]):
            k = ref.Node(prior=p, move=m, parent=parent); k.n = n; k.q = q
            parent.children[m] = k
            kids.append(k)
        infl = {k: v for k, v in zip(kids, c["inflight"]) if v}
        node, path, _ = o.select(b.copy(), parent, infl)
        assert kids.index(path[1]) == c["chosen"], c["uid"]
        assert [infl.get(k, 0) for k in kids] == c["inflight_after"]
        assert o.jitter.ctr == c["draws"]
        # without the dict the oracle must pick differently in some cases, or the cases pin nothing
        o2 = ref.MCTS(ref.MCTSConfig.from_dict(dict(BASE, selection_jitter=c["jitter"], fpu_reduction=c["fpu_reduction"],
                                                    use_tt=False)), None, seed=GV["select_one_level"]["seed"], game=c["uid"])
        _, path2, _ = o2.select(b.copy(), parent, None)
        n_pen += kids.index(path2[1]) != c["chosen"]
    assert n_pen >= 10


def _oracle_run_vl(case, numerics="reference"):
    cfg = ref.MCTSConfig.from_dict(dict(BASE, **case["mcts_extra"], inference_batch_size=case["L"], use_tt=(case["tt"] == "on"),
                                        virtual_loss_active=True, numerics=numerics))
    net = HashNet(**case["net"])
    o = ref.MCTS(cfg, net.infer_np, seed=case["seed"], game=case["uid"])
    vc, pi, rq = o.run(ch.Board(FENS[case["fen"]]), num_simulations=case["sims"], ply=(0 if case["dirichlet"] else 1000))
    return o, net, (_snap(o._last_root), vc, pi, rq, o._last_sims_run)


@pytest.mark.parametrize("tt", ["off", "on"])
@pytest.mark.parametrize("numerics,tol", [("reference", 1e-7), ("engine", 1e-6)])
def test_whole_searches_with_virtual_loss_match_reference(tt, numerics, tol):
    """MCTS.run with the batch's in-flight dict handed to _select (VLOn): 96 ... 1 600 simulations, batches of 16 / 32 / 96
    (bench.py's search shape: 800 simulations, 96 leaves per batch), virtual_loss 0 / 0.3 / 1 / 3, Dirichlet on, terminal
    leaves inside a batch, pruning; tree-only and with the table on.  The oracle with virtual_loss_active=True must give
    identical visit counts, policy targets, evaluation counts and stream positions."""
    n = n_differs = 0
    for case in GV["runs"]:
        if case["tt"] != tt:
            continue
        o, net, (root, vc, pi, rq, sims) = _oracle_run_vl(case, numerics)
        _check_root(root, vc, pi, rq, sims, case["results"][0], tol)
        assert net.calls == case["evals"], case["name"]
        assert (o.jitter.ctr, o.noise.ctr, o.dirichlet.ctr) == (case["draws"]["jitter"], case["draws"]["noise"], case["draws"]["dirichlet"])
        n += 1
        n_differs += bool(case["differs_from_vl_off"])
    assert n >= 22 and n_differs >= n - 1          # every case but virtual_loss = 0 is a search the penalty changed
