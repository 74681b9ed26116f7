"""`engine.eval_cache`: (a) a leaf reached more than once in one pass (96 descents over a young tree land on the same
unexpanded node many times) shares ONE batch row, (b) per-game evaluation cache (csrc/tree.h EvalCache): a leaf whose position
the game evaluated in an earlier pass is expanded from the stored value + legal logits.  The 320-wide forward is bitwise batch invariant, so a fresh evaluation would return the very same
numbers: the games must be identical, bit for bit, with the cache on and off -- moves, visit distributions, values -- and every
simulation that was a network evaluation without the cache is either an evaluation or a cache hit with it."""
import numpy as np
import pytest

from oracle import net_ref

pytestmark = pytest.mark.gpu

NET = dict(planes=19, channels=320, blocks=3, attention_heads=20, policy_size=4672, norm="group", activation="silu",
           preact=True, policy_factor_rank=128, self_supervised=False)
CFG = {"seed": 77,
       "mcts": {"cpuct": 2.5, "cpuct_start": 3.0, "cpuct_end": 2.0, "cpuct_plies": 40, "dirichlet_plies": 30,
                "selection_jitter": 0.05, "fpu_reduction": 0.1, "draw_penalty": -0.05, "legal_softmax": True,
                "inference_batch_size": 32, "playout_random_frac": 0.05},
       "selfplay": {"num_simulations": 200, "max_game_len": 30, "min_resign_plies": 50, "resign_threshold": -0.85,
                    "opening_random_plies": 4, "temperature_start": 1.2, "temperature_end": 0.3, "temperature_moves": 40}}


def _play(be, eval_cache, **kw):
    from matrix0_amd import engine as eng
    e = eng.SelfplayEngine(be, eng.selfplay_cfg_from_dict(CFG, concurrent_games=6, total_games=8, eval_cache=eval_cache, **kw))
    games = {}
    for _ in range(4000):
        e.step(4)
        while (r := e.poll()) is not None:
            games[r["game_index"]] = r
        if not e.running():
            break
    st = e.stats()
    e.close()
    return games, st


@pytest.mark.parametrize("mode", ["tree_reuse_vl", "fresh_tree_no_vl"])
def test_games_are_identical_with_and_without_the_cache(mode):
    from matrix0_amd.backend import M0Backend
    be = M0Backend.from_state_dict(NET, net_ref.random_state_dict(NET, seed=9))
    kw = {} if mode == "tree_reuse_vl" else dict(virtual_loss_active=False, compat={"fresh_tree_per_move": True})
    off, st_off = _play(be, False, **kw)
    on, st_on = _play(be, True, **kw)
    be.close()
    assert sorted(off) == sorted(on) == list(range(8))
    for i in range(8):
        a, b = off[i], on[i]
        assert a["played"] == b["played"] and a["result"] == b["result"], i
        for k in ("pi", "z", "s", "legal_mask", "search_values"):
            assert np.array_equal(a[k], b[k]), (i, k)
    assert st_off["evals_cached"] == 0 and st_on["evals_cached"] > 0
    assert st_on["sims"] == st_off["sims"] and st_on["plies"] == st_off["plies"]
    assert st_on["evals"] + st_on["evals_cached"] == st_off["evals"]
    # a fresh tree per move re-expands what the previous search already knew: far more hits than with the subtree kept
    print(f"eval cache [{mode}]: {int(st_on['evals_cached'])} of {int(st_off['evals'])} leaf evaluations served from the cache "
          f"({100.0 * st_on['evals_cached'] / st_off['evals']:.1f} %)")
    assert st_on["evals_cached"] > 0.005 * st_off["evals"]


def test_cache_is_off_where_the_payload_cannot_serve_the_expansion():
    """Full-policy softmax and the table / raw-prior compat modes need more than the legal logits: the switch is ignored there."""
    from matrix0_amd.backend import M0Backend
    from matrix0_amd import engine as eng
    be = M0Backend.from_state_dict(NET, net_ref.random_state_dict(NET, seed=9))
    cfgd = {k: (dict(v) if isinstance(v, dict) else v) for k, v in CFG.items()}
    cfgd["mcts"] = dict(cfgd["mcts"], legal_softmax=False)
    cfgd["selfplay"] = dict(cfgd["selfplay"], max_game_len=3)
    e = eng.SelfplayEngine(be, eng.selfplay_cfg_from_dict(cfgd, concurrent_games=2, total_games=2, eval_cache=True))
    while e.running():
        e.step(8)
    assert e.stats()["evals_cached"] == 0
    e.close(); be.close()


def test_full_size_worker_configuration_is_identical_with_and_without_the_cache():
    """The configuration the drop-in worker and bench.py ship with: R24-320, 256 concurrent games, 800 simulations per move at
    96 leaves per tree and pass, virtual loss on, subtree reuse -- two searched plies per game, cache on against off: the same
    moves, the same visit distributions, and every evaluation accounted for."""
    from matrix0_amd.backend import M0Backend
    from matrix0_amd import engine as eng
    net = dict(NET, blocks=24)
    be = M0Backend.from_state_dict(net, net_ref.random_state_dict(net, seed=12))
    cfgd = {k: (dict(v) if isinstance(v, dict) else v) for k, v in CFG.items()}
    cfgd["mcts"] = dict(cfgd["mcts"], inference_batch_size=96, playout_random_frac=0.0)
    cfgd["selfplay"] = dict(cfgd["selfplay"], num_simulations=800, max_game_len=2, opening_random_plies=0)

    def play(cache):
        e = eng.SelfplayEngine(be, eng.selfplay_cfg_from_dict(cfgd, concurrent_games=256, total_games=256, eval_cache=cache))
        games = {}
        for _ in range(400):
            e.step(1)
            while (r := e.poll()) is not None:
                games[r["game_index"]] = r
            if not e.running():
                break
        st = e.stats()
        e.close()
        return games, st

    off, st_off = play(False)
    on, st_on = play(True)
    be.close()
    assert sorted(off) == sorted(on) == list(range(256))
    for i in range(256):
        assert off[i]["played"] == on[i]["played"], i
        for k in ("pi", "z", "s", "legal_mask", "search_values"):
            assert np.array_equal(off[i][k], on[i][k]), (i, k)
    assert st_on["sims"] == st_off["sims"] and st_on["plies"] == st_off["plies"] == 512
    assert st_on["evals"] + st_on["evals_cached"] == st_off["evals"] and st_on["evals_cached"] > 0
    print(f"eval cache [R24-320, 256 x 800 x 2 plies]: {int(st_on['evals_cached'])} of {int(st_off['evals'])} served from the cache")
