"""The path bench.py times -- m0_selfplay_step: select_kernel writes the leaves' fp16 NHWC planes on the device
(encode_nhwc), the network reads them in place, its logits / values stay on the device for expand_kernel -- pinned to the
split step that every golden test goes through (m0_selfplay_ext_select -> host planes f32 -> infer_np -> ext_expand).
The 320-wide network is bitwise batch-invariant, so the two must produce the same games bit for bit; a wrong channel,
square order or rounding in encode_nhwc, or a row mix-up between select, forward and expand, changes a visit count
somewhere.  Reference: encoding.py:11-46 (encode_board), mcts.py:570-621 (leaf planes -> infer_np -> expansion)."""
import numpy as np
import pytest

from oracle import net_ref

pytestmark = pytest.mark.gpu

NET = dict(planes=19, channels=320, blocks=3, attention_heads=20, policy_size=4672, norm="group", activation="silu",
           preact=True, policy_factor_rank=128, self_supervised=False)
CFG = {"seed": 4242,
       "mcts": {"cpuct": 2.5, "cpuct_start": 3.0, "cpuct_end": 2.0, "cpuct_plies": 40, "dirichlet_plies": 30,
                "selection_jitter": 0.05, "fpu_reduction": 0.1, "draw_penalty": -0.05, "legal_softmax": True,
                "inference_batch_size": 24, "playout_random_frac": 0.05},
       "selfplay": {"num_simulations": 96, "max_game_len": 24, "min_resign_plies": 50, "resign_threshold": -0.85,
                    "opening_random_plies": 6, "temperature_start": 1.2, "temperature_end": 0.3, "temperature_moves": 40}}


def _records_equal(a, b, tag):
    assert a["played"] == b["played"] and a["moves"] == b["moves"], tag
    assert a["result"] == b["result"] and a["resigned"] == b["resigned"] and a["draw"] == b["draw"], tag
    for k in ("pi", "z", "s", "legal_mask", "search_values"):
        assert np.array_equal(a[k], b[k]), (tag, k)


@pytest.mark.parametrize("mode", ["default", "vl_off_fresh_tree"])
def test_fused_step_equals_split_step_bit_for_bit(mode):
    from matrix0_amd.backend import M0Backend
    from matrix0_amd import engine as eng
    sd = net_ref.random_state_dict(NET, seed=7)
    be = M0Backend.from_state_dict(NET, sd)
    kw = dict(concurrent_games=5, total_games=8)
    if mode == "vl_off_fresh_tree":
        kw.update(virtual_loss_active=False, compat={"fresh_tree_per_move": True})
    # fused: m0_selfplay_step
    e = eng.SelfplayEngine(be, eng.selfplay_cfg_from_dict(CFG, **kw))
    fused = {}
    for _ in range(4000):
        e.step(4)
        while (r := e.poll()) is not None:
            fused[r["game_index"]] = r
        if not e.running():
            break
    st_f = e.stats()
    e.close()
    # split: ext_select -> infer_np -> ext_expand, and at every step the device batch select wrote vs the host planes
    e = eng.SelfplayEngine(be, eng.selfplay_cfg_from_dict(CFG, **kw))
    split = {}
    checked_rows = 0
    for _ in range(16000):
        planes = e.ext_select()
        if planes.shape[0]:
            dev = e.last_batch_nhwc()
            assert dev.shape[0] == planes.shape[0]
            assert np.array_equal(dev.view(np.uint16), eng.planes_to_nhwc(planes).view(np.uint16)), "select's device batch != planes"
            checked_rows += planes.shape[0]
            lg, vv = be.infer_np(planes)
        else:
            lg, vv = np.zeros((0, 4672), np.float32), np.zeros((0,), np.float32)
        e.ext_expand(lg, vv)
        while (r := e.poll()) is not None:
            split[r["game_index"]] = r
        if not e.running():
            break
    st_s = e.stats()
    e.close()
    be.close()
    assert sorted(fused) == sorted(split) == list(range(8))
    assert checked_rows > 5000
    for k in ("evals", "sims", "plies", "games_finished", "steps"):
        assert st_f[k] == st_s[k], k
    for i in range(8):
        _records_equal(fused[i], split[i], i)


def test_ext_select_refuses_a_small_buffer_before_it_touches_the_trees():
    """A planes buffer below the worst case is refused BEFORE select runs: the engine stays usable."""
    import ctypes as C
    from matrix0_amd import engine as eng
    e = eng.SelfplayEngine(None, eng.selfplay_cfg_from_dict(CFG, concurrent_games=2, total_games=2))
    rows = C.c_int(0)
    small = np.zeros((3, 19, 8, 8), np.float32)
    assert e._L.m0_selfplay_ext_select(e._h, C.byref(rows), small.ctypes.data_as(C.c_void_p), 3) == -1
    assert e._L.m0_selfplay_ext_select(e._h, C.byref(rows), None, 1000) == -1
    planes = e.ext_select()                       # still usable: nothing was reserved by the refused calls
    assert planes.shape[0] == 2
    e.close()
