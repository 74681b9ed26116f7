"""Pin oracle/net_ref.py (fp32 CPU restatement) against golden vectors produced by
the real reference PolicyValueNet (tools/gen_golden_net.py).  Tolerances from
SURVEY App. A.3: max|dlogit| <= 1e-4, |dv| <= 1e-5."""
import numpy as np
import pytest
import torch

from oracle import net_ref
from tests.golden_util import NET_CASES, SEEDED_NET_CASES, load_net_golden, load_seeded_net_golden


@pytest.mark.parametrize("name", NET_CASES)
def test_oracle_matches_reference_golden(name):
    cfg, sd, x, p_ref, v_ref, ssl_ref = load_net_golden(name)
    p, v, ssl = net_ref.forward(sd, cfg, torch.from_numpy(x), return_ssl=bool(ssl_ref))
    assert p.shape == (x.shape[0], 4672) and v.shape == (x.shape[0],)
    assert np.abs(p.numpy() - p_ref).max() <= 1e-4
    assert np.abs(v.numpy() - v_ref).max() <= 1e-5
    for t, ref in ssl_ref.items():
        assert np.abs(ssl[t].numpy() - ref).max() <= 1e-4, t


@pytest.mark.parametrize("name", SEEDED_NET_CASES)
def test_oracle_matches_reference_module_at_full_width(name):
    """320 channels, 20 heads, GroupNorm(20 groups), factorised policy head, five SSL heads: the reference module's own numbers
    (seeded weights, tools/gen_golden_net.py seeded) -- the oracle is pinned at the width the benchmark runs, not only at 32 / 64."""
    cfg, sd, x, p_ref, v_ref, ssl_ref = load_seeded_net_golden(name)
    p, v, ssl = net_ref.forward(sd, cfg, torch.from_numpy(x), return_ssl=True)
    assert np.abs(p.numpy() - p_ref).max() <= 1e-4
    assert np.abs(v.numpy() - v_ref).max() <= 1e-5
    assert set(ssl_ref) == {"piece", "threat", "pin", "fork", "control"}
    for t, ref in ssl_ref.items():
        assert np.abs(ssl[t].numpy() - ref).max() <= 1e-4, t


def test_param_shapes_match_golden_state_dict():
    for name in NET_CASES:
        cfg, sd, *_ = load_net_golden(name)
        shapes = net_ref.param_shapes(cfg)
        assert set(shapes) == set(sd), (name, set(shapes) ^ set(sd))
        for k, s in shapes.items():
            assert tuple(sd[k].shape) == tuple(s), (name, k)


def test_r24_320_param_count():
    """the documented '53M' config instantiates to 57,562,210 parameters."""
    cfg = dict(planes=19, channels=320, blocks=24, attention_heads=20, policy_size=4672, norm="group",
               activation="silu", preact=True, policy_factor_rank=128, self_supervised=True,
               ssl_tasks=["piece", "threat", "pin", "fork", "control"])
    n = 0
    for k, s in net_ref.param_shapes(cfg).items():
        c = 1
        for d in s:
            c *= d
        n += c
    assert n == 57_562_210
