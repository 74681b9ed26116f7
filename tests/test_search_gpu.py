"""GPU parity of the tree kernels (select / expand / backup / Dirichlet / re-root) against the oracle
restatement of azchess/mcts.py, with identical injected random streams and an identical evaluator.

Integer results (visit counts, move order, policy indices) must be identical; fp64 statistics within
1e-9; float32 priors within 1e-6 (tests/test_mcts_logits.py tolerance)."""
import numpy as np
import pytest

from oracle import chess_py as ch
from oracle import mcts_ref as ref
from tests.fake_net import FakeNet

pytestmark = pytest.mark.gpu

MCTS = {"cpuct": 2.5, "cpuct_start": 3.0, "cpuct_end": 2.0, "cpuct_plies": 40, "dirichlet_alpha": 0.3,
        "dirichlet_frac": 0.25, "dirichlet_plies": 30, "selection_jitter": 0.05, "fpu_reduction": 0.1,
        "draw_penalty": -0.05, "virtual_loss": 1.0, "legal_softmax": True, "enable_entropy_noise": True,
        "no_instant_backtrack": True, "playout_random_frac": 0.0}

FENS = [ch.START_FEN,
        "r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1",
        "8/2p5/3p4/KP5r/1R3p1k/8/4P1P1/8 w - - 0 1",
        "r1bq1rk1/pp2bppp/2n1pn2/2pp4/3P1B2/2PBPN2/PP1N1PPP/R2QK2R b KQ - 3 8",
        "6k1/5ppp/8/8/8/8/5PPP/3R2K1 w - - 0 1",          # mate in 1 available: terminal leaves
        "7k/5Q2/5K2/8/8/8/8/8 w - - 0 1"]               # stalemate / mate leaves near the root


def _engine(G, L, sims, vl_active=True, legal_softmax=True, noise=True):
    from matrix0_amd import engine as eng
    m = dict(MCTS, inference_batch_size=L, legal_softmax=legal_softmax, enable_entropy_noise=noise)
    cfg = eng.selfplay_cfg_from_dict({"seed": 1234, "mcts": m, "selfplay": {"num_simulations": sims}},
                                     concurrent_games=G, virtual_loss_active=vl_active)
    return eng.SelfplayEngine(None, cfg), m


def _run_engine_search(e, net, G):
    for _ in range(10000):
        planes = e.search_select()
        if planes.shape[0] == 0 and all(e.search_result(g)["finished"] for g in range(G)):
            break
        lg, v = net.infer_np(planes) if planes.shape[0] else (np.zeros((0, 4672), np.float32), np.zeros((0,), np.float32))
        e.search_expand(lg, v)
        if all(e.search_result(g)["finished"] for g in range(G)):
            break
    return [e.search_result(g) for g in range(G)]


def _oracle(fen, uid, sims, L, net, m, vl_active, dirichlet, ply=0, numerics="engine"):
    cfg = ref.MCTSConfig.from_dict(dict(m, use_tt=False, virtual_loss_active=vl_active, inference_batch_size=L,
                                        dirichlet_plies=(30 if dirichlet else 0), numerics=numerics))
    o = ref.MCTS(cfg, net.infer_np, seed=1234, game=uid)
    b = ch.Board(fen)
    vc, pi, rq = o.run(b, num_simulations=sims, ply=ply)
    return o, b, vc, pi, rq


def _compare(res, o, vc, rq):
    root = o._last_root
    kids = list(root.children.values())
    assert res["moves"] == [c.move.uci() for c in kids]
    assert res["idx"].tolist() == [c.move_idx for c in kids]
    assert res["n"].tolist() == [c.n for c in kids], (res["n"].tolist(), [c.n for c in kids])
    np.testing.assert_allclose(res["prior"], [c.prior for c in kids], rtol=0, atol=1e-6)
    np.testing.assert_allclose(res["q"], [c.q for c in kids], rtol=0, atol=1e-6)
    assert abs(res["root_q"] - rq) < 1e-6
    assert res["root_n"] == root.n


@pytest.mark.parametrize("vl_active,dirichlet,sharp", [(True, True, 8.0), (False, False, 8.0), (True, True, 0.5),
                                                       (True, False, 30.0)])
def test_search_matches_oracle(vl_active, dirichlet, sharp):
    G, L, sims = len(FENS), 8, 96
    e, m = _engine(G, L, sims, vl_active=vl_active)
    net = FakeNet(seed=3, sharp=sharp)
    for g, fen in enumerate(FENS):
        e.search_begin(g, fen, sims, dirichlet, 100 + g)
    results = _run_engine_search(e, net, G)
    for g, fen in enumerate(FENS):
        o, b, vc, pi, rq = _oracle(fen, 100 + g, sims, L, FakeNet(seed=3, sharp=sharp), m, vl_active, dirichlet)
        _compare(results[g], o, vc, rq)


def test_full_softmax_mode_and_no_noise():
    G, L, sims = 2, 4, 40
    e, m = _engine(G, L, sims, legal_softmax=False, noise=False)
    net = FakeNet(seed=5, sharp=12.0)
    for g in range(G):
        e.search_begin(g, FENS[g], sims, False, 7 + g)
    results = _run_engine_search(e, net, G)
    for g in range(G):
        o, b, vc, pi, rq = _oracle(FENS[g], 7 + g, sims, L, FakeNet(seed=5, sharp=12.0), m, True, False)
        _compare(results[g], o, vc, rq)


class _PoisonedNet:
    """FakeNet whose logits rows carry one non-finite value (inf, -inf or nan, at an index that is usually not even a
    legal move) for positions chosen by a hash of the input: Node._expand then uses uniform priors (mcts.py:147-149)."""

    def __init__(self, seed, sharp):
        self.net = FakeNet(seed=seed, sharp=sharp)

    def infer_np(self, planes):
        lg, v = self.net.infer_np(planes)
        lg = np.array(lg, np.float32, copy=True)
        for i in range(lg.shape[0]):
            h = int(np.asarray(planes[i]).sum() * 7919) % 5
            if h == 0:
                lg[i, 4671] = np.inf
            elif h == 1:
                lg[i, 17] = np.nan
            elif h == 2:
                lg[i, 2300] = -np.inf
        return lg, v


def test_nonfinite_logits_give_uniform_priors_as_in_the_reference():
    G, L, sims = 3, 8, 64
    e, m = _engine(G, L, sims)
    net = _PoisonedNet(seed=9, sharp=6.0)
    for g in range(G):
        e.search_begin(g, FENS[g], sims, True, 40 + g)
    results = _run_engine_search(e, net, G)
    for g in range(G):
        o, b, vc, pi, rq = _oracle(FENS[g], 40 + g, sims, L, _PoisonedNet(seed=9, sharp=6.0), m, True, True)
        _compare(results[g], o, vc, rq)


def test_tree_reuse_across_moves_matches_oracle():
    """Play 4 plies, each a fresh search on the re-rooted (compacted) subtree; visits accumulate as in the
    reference (TT-reused root, mcts.py:342-371) and Dirichlet re-applies to already-noised priors."""
    G, L, sims = 2, 8, 64
    e, m = _engine(G, L, sims)
    net = FakeNet(seed=9, sharp=8.0)
    oracles = []
    for g in range(G):
        e.search_begin(g, FENS[g], sims, True, 50 + g)
        cfg = ref.MCTSConfig.from_dict(dict(m, use_tt=False, virtual_loss_active=True, inference_batch_size=L, numerics="engine"))
        oracles.append((ref.MCTS(cfg, FakeNet(seed=9, sharp=8.0).infer_np, seed=1234, game=50 + g), ch.Board(FENS[g])))
    for ply in range(4):
        results = _run_engine_search(e, net, G)
        for g in range(G):
            o, b = oracles[g]
            vc, pi, rq = o.run(b, num_simulations=sims, ply=ply)
            _compare(results[g], o, vc, rq)
            # play the most visited move (first max), alternate with the least visited expanded child
            n = results[g]["n"]
            slot = int(np.argmax(n)) if ply % 2 == 0 else int(np.argmin(np.where(n > 0, n, 10**9)))
            mv = ch.Move.from_uci(results[g]["moves"][slot])
            o.note_move_played(mv)
            b.push(mv)
            e.search_advance(g, slot, sims, True)


@pytest.mark.parametrize("sharp,legal_softmax", [(8.0, True), (0.5, True), (30.0, True), (12.0, False)])
def test_expand_priors_vs_reference_numerics(sharp, legal_softmax):
    """Node._expand priors from the HIP kernel vs the reference-faithful float32 path (torch.softmax, numpy
    float32 sums): tolerance 1e-6 absolute, the bound tests/test_mcts_logits.py uses."""
    G = len(FENS)
    e, m = _engine(G, 4, 1, legal_softmax=legal_softmax)
    net = FakeNet(seed=11, sharp=sharp)
    for g, fen in enumerate(FENS):
        e.search_begin(g, fen, 1, False, 300 + g)
    results = _run_engine_search(e, net, G)
    for g, fen in enumerate(FENS):
        b = ch.Board(fen)
        moves, idxs = ch.legal_moves_with_indices(b)
        lg, _ = FakeNet(seed=11, sharp=sharp).infer_np(ch.encode_board(b)[None])
        noise = ref.Stream(ref.derive_seed(1234, 300 + g, ref.PURPOSE_NOISE))
        want = ref.legal_priors(lg[0], idxs, legal_softmax, True, noise, numerics="reference")
        np.testing.assert_allclose(results[g]["prior"], want.astype(np.float64), rtol=0, atol=1e-6)
        assert abs(results[g]["prior"].sum() - 1.0) < 1e-5


def test_c_base_cpuct_schedule_and_long_search():
    """KataGo-style cpuct (mcts.py:929-934) and a 1600-simulation search with Dirichlet + virtual loss
    (BASELINE configs[4] search shape) against the oracle."""
    from matrix0_amd import engine as eng
    G, L, sims = 2, 16, 1600
    m = dict(MCTS, inference_batch_size=L, cpuct_c_base=19652.0, cpuct_c_init=1.25)
    cfg = eng.selfplay_cfg_from_dict({"seed": 1234, "mcts": m, "selfplay": {"num_simulations": sims}}, concurrent_games=G)
    assert cfg.use_c_base == 1
    e = eng.SelfplayEngine(None, cfg)
    net = FakeNet(seed=21, sharp=10.0)
    for g in range(G):
        e.search_begin(g, FENS[g], sims, True, 900 + g)
    results = _run_engine_search(e, net, G)
    for g in range(G):
        o, b, vc, pi, rq = _oracle(FENS[g], 900 + g, sims, L, FakeNet(seed=21, sharp=10.0), m, True, True)
        _compare(results[g], o, vc, rq)
        assert sum(results[g]["n"]) == sims


def test_pruning_config_is_validated():
    from matrix0_amd import engine as eng
    cfg = eng.selfplay_cfg_from_dict({"mcts": {"max_children": 8, "min_child_prior": 0.01}}, concurrent_games=1)
    assert cfg.max_children == 8 and abs(cfg.min_child_prior - 0.01) < 1e-12
    with pytest.raises(ValueError):
        eng.selfplay_cfg_from_dict({"mcts": {"max_children": 1000}}, concurrent_games=1)
    with pytest.raises(ValueError, match="compat"):
        eng.selfplay_cfg_from_dict({}, concurrent_games=1, compat={"no_such_switch": True})
