"""Host-side decision functions of the engine (exposed through the C-ABI, no GPU needed) vs the
oracle restatement of selfplay/internal.py / draw.py / mcts.py:378-387, and C-ABI surface checks."""
import re
import os

import numpy as np
import pytest

from oracle import chess_py as ch
from oracle import mcts_ref as ref
from matrix0_amd import engine as eng
from matrix0_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "m0_engine.h")).read()
    names = sorted(set(re.findall(r"\b(m0_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 25
    L = _lib.lib()
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/m0_engine.h but not exported"


def test_create_without_gpu_fails_loudly_or_works():
    """No CPU fallback: on a box without a HIP device creation must raise, never silently degrade."""
    import torch
    from matrix0_amd.backend import M0Backend
    cfg = dict(channels=32, blocks=1, attention_heads=2, norm="group", activation="silu", preact=True)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no HIP device"):
            M0Backend(cfg)


def test_playout_cap_and_temperature():
    rng = np.random.default_rng(0)
    for _ in range(500):
        sims = int(rng.integers(1, 2000)); frac = float(rng.choice([0.0, 0.05, 0.3])); u = float(rng.random())
        assert eng.playout_cap(sims, frac, u) == ref.playout_cap(sims, frac, u)
        fm = int(rng.integers(0, 120))
        assert eng.temperature_for(fm, 1.2, 0.3, 40) == ref.temperature_for(fm, 1.2, 0.3, 40)
    assert eng.temperature_for(5, 1.0, 0.1, 0) == 0.1


def test_sample_move_index_matches_oracle():
    rng = np.random.default_rng(1)
    same = 0
    N = 3000
    for _ in range(N):
        n = int(rng.integers(1, 219))
        visits = rng.integers(0, 800, size=n).astype(np.int32)
        if rng.random() < 0.1:
            visits[:] = 0
        T = float(rng.choice([0.0, 0.3, 0.5, 0.975, 1.2]))
        u = float(rng.random())
        same += eng.sample_move_index(visits, T, u) == ref.sample_move_index(visits.tolist(), T, u)
    assert same >= N - 2     # powf vs numpy's float32 power may differ by an ulp on a CDF boundary


def test_rules_probe_vs_oracle_on_random_playouts():
    cfg = eng.selfplay_cfg_from_dict({"selfplay": {"draw": {"min_plies": 30, "window": 8, "min_unique": 4, "halfmove_cap": 100}}},
                                     concurrent_games=1)
    dcfg = {"min_plies": 30, "window": 8, "min_unique": 4, "halfmove_cap": 100}
    rng = np.random.default_rng(3)
    checked = 0
    for g in range(25):
        b = ch.Board()
        ucis, moves = [], []
        for ply in range(220):
            if b.is_game_over():
                break
            lm = b.legal_moves
            # bias toward shuffling so repetitions / 50-move claims occur
            m = lm[int(rng.integers(len(lm)))] if rng.random() < 0.7 else lm[0]
            b.push(m); ucis.append(m.uci()); moves.append(m)
            if ply % 7 == 0 or b.is_game_over(claim_draw=True):
                r = eng.rules_probe(cfg, ch.START_FEN, ucis)
                assert r["game_over"] == b.is_game_over()
                assert r["game_over_claim"] == b.is_game_over(claim_draw=True)
                assert r["checkmate"] == b.is_checkmate() and r["stalemate"] == b.is_stalemate()
                assert r["insufficient"] == b.is_insufficient_material()
                assert r["can_claim_fifty"] == b.can_claim_fifty_moves()
                assert r["repetition3"] == b.is_repetition(3) and r["fivefold"] == b.is_repetition(5)
                assert r["can_claim_threefold"] == b.can_claim_threefold_repetition()
                assert r["adjudicate_draw"] == ref.should_adjudicate_draw(b, moves, dcfg)
                assert r["result"] == ref.game_result(b)
                checked += 1
    assert checked > 200


def test_config_mapping_matches_reference_yaml_semantics():
    cfg = {"seed": 7,
           "mcts": {"num_simulations": 300, "cpuct": 2.5, "cpuct_start": 3.0, "cpuct_end": 2.0, "cpuct_plies": 40,
                    "dirichlet_plies": 30, "selection_jitter": 0.05, "fpu_reduction": 0.1, "draw_penalty": -0.05,
                    "legal_softmax": True, "inference_batch_size": 96, "playout_random_frac": 0.05},
           "selfplay": {"num_simulations": 800, "max_game_len": 200, "min_resign_plies": 50, "resign_threshold": -0.85,
                        "opening_random_plies": 12, "temperature_start": 1.2, "temperature_end": 0.3, "temperature_moves": 40,
                        "draw": {"min_plies": 30, "window": 8, "min_unique": 4, "halfmove_cap": 100}}}
    c = eng.selfplay_cfg_from_dict(cfg, concurrent_games=4)
    assert c.num_simulations == 800            # selfplay section overrides mcts (internal.py:291)
    assert (c.cpuct_start, c.cpuct_end, c.cpuct_plies) == (3.0, 2.0, 40)
    assert c.dirichlet_plies == 30 and c.legal_softmax == 1 and c.inference_batch_size == 96
    assert c.draw_enabled == 0 and c.draw_halfmove_cap == 100 and c.seed == 7
    assert c.resign_consecutive_bad == 5 and c.resign_window == 4


def test_engine_section_switches_map_to_the_cabi_fields():
    """`engine.eval_cache` / `engine.tail_split` of config.yaml (engine-only switches, off unless asked for) -> m0_selfplay_cfg."""
    base = eng.selfplay_cfg_from_dict({}, concurrent_games=4)
    assert base.eval_cache == 0 and base.tail_split == 0
    for given, want in ((False, 0), (True, 1), (1, 1), (2, 2), ("halves", 2), (0, 0)):
        assert eng.selfplay_cfg_from_dict({}, concurrent_games=4, tail_split=given).tail_split == want, given
        assert eng.selfplay_cfg_from_dict({"engine": {"tail_split": given}}, concurrent_games=4).tail_split == want, given
    # the keyword wins over the config section
    assert eng.selfplay_cfg_from_dict({"engine": {"tail_split": "halves", "eval_cache": True}}, concurrent_games=4,
                                      tail_split=False, eval_cache=False).tail_split == 0
    assert eng.selfplay_cfg_from_dict({"engine": {"eval_cache": True}}, concurrent_games=4).eval_cache == 1
