"""SSL loss heads (azchess/model/resnet.py:892-1130) pinned by tests/golden/ssl_loss.npz, which the REAL reference
module produced (tools/gen_golden_ssl_loss.py: PolicyValueNet.get_enhanced_ssl_loss on a five-head network with targets from
the reference's ssl_algorithms.py).  CPU: the host function on the reference's own head outputs; GPU: the same loss computed
from the HIP network's head outputs (fp16 forward) against the reference's value."""
import json

import numpy as np
import pytest

from matrix0_amd import ssl_loss
from tests.golden_ref import load_npz

TASKS = ("piece", "threat", "pin", "fork", "control")


@pytest.fixture(scope="module")
def gold():
    z = load_npz("ssl_loss.npz")
    g = {k: z[k] for k in z.files}
    g["cases"] = json.loads(str(g["cases_json"]))
    g["cfg"] = json.loads(str(g["cfg_json"]))
    return g


def test_loss_matches_the_reference_on_its_own_head_outputs(gold):
    heads = {t: gold[f"head_{t}"] for t in TASKS}
    targets = {t: gold[f"target_{t}"] for t in TASKS}
    for c in gold["cases"]:
        tg = dict(targets)
        if c.get("piece_as_index"):
            tg["piece"] = np.argmax(targets["piece"], axis=1)
        if c.get("threat_zero"):
            tg["threat"] = np.zeros_like(targets["threat"])
        got = ssl_loss.enhanced_ssl_loss(heads, tg, c["tasks"], c["weights"])
        assert abs(got["total"] - c["total"]) <= 2e-6 * max(1.0, abs(c["total"])), (c["tasks"], got, c["total"])
        for t, want in c["single"].items():
            assert abs(got[t] - want) <= 2e-6 * max(1.0, abs(want)), (t, got[t], want)
        assert set(got) - {"total"} <= set(c["tasks"])


def test_target_shapes_and_degenerate_cases():
    rng = np.random.default_rng(0)
    out = {"piece": rng.normal(size=(2, 13, 8, 8)).astype(np.float32), "control": rng.normal(size=(2, 3, 8, 8)).astype(np.float32),
           "threat": rng.normal(size=(2, 1, 8, 8)).astype(np.float32)}
    onehot = np.zeros((2, 13, 8, 8), np.float32)
    idx = rng.integers(0, 13, size=(2, 8, 8))
    np.put_along_axis(onehot, idx[:, None], 1.0, axis=1)
    a = ssl_loss.task_loss("piece", out["piece"], onehot)
    b = ssl_loss.task_loss("piece", out["piece"], idx)
    assert a == b and a > 0
    assert ssl_loss.task_loss("threat", out["control"], np.zeros((2, 8, 8))) is None        # wrong head width
    assert ssl_loss.task_loss("king_safety", out["control"], np.zeros((2, 8, 8))) is None   # no such head in self-play shards
    # a task that is not enabled, or has no target, contributes nothing
    r = ssl_loss.enhanced_ssl_loss(out, {"piece": idx, "threat": np.ones((2, 8, 8))}, ["piece"])
    assert set(r) == {"piece", "total"} and r["total"] == r["piece"]
    # control classes by threshold
    ctl = np.array([[-1.0, -0.4, 0.0, 0.6]]).reshape(1, 2, 2)
    logits = np.zeros((1, 3, 2, 2), np.float32)
    logits[0, 0, 0, 0] = logits[0, 1, 0, 1] = logits[0, 1, 1, 0] = logits[0, 2, 1, 1] = 20.0
    assert ssl_loss.task_loss("control", logits, ctl) < 1e-6


@pytest.mark.gpu
def test_loss_from_the_hip_heads_matches_the_reference(gold):
    """The five SSL heads of the HIP forward on the golden positions -> the same loss as the reference's torch fp32 module, within
    what fp16 activations allow (head logits agree to 1e-2: tests/test_net_gpu.py; the mean losses to 2e-3)."""
    from matrix0_amd.backend import M0Backend
    from tests.golden_util import load_net_golden
    cfg, sd, _, _, _, _ = load_net_golden("gn_silu_preact")          # the network the reference computed the losses with
    be = M0Backend.from_state_dict(cfg, sd)
    _, _, heads = be.infer_np_ssl(gold["x"])
    be.close()
    targets = {t: gold[f"target_{t}"] for t in TASKS}
    for t in TASKS:
        assert np.abs(heads[t] - gold[f"head_{t}"]).max() <= 1e-2, t
    c = gold["cases"][0]
    got = ssl_loss.enhanced_ssl_loss(heads, targets, c["tasks"], c["weights"])
    assert abs(got["total"] - c["total"]) <= 2e-3 * c["total"], (got, c["total"])
    for t, want in c["single"].items():
        assert abs(got[t] - want) <= 2e-3 * max(1.0, want), (t, got[t], want)
