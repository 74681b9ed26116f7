"""The shared-memory inference server with the HIP network behind it: a spawned server process owns the
GPU network; clients in this process get exactly what an in-process M0Backend returns."""
import multiprocessing as mp

import numpy as np
import pytest

from oracle import net_ref
from tests.golden_util import load_net_golden

pytestmark = pytest.mark.gpu


def test_spawned_server_serves_the_hip_network():
    from matrix0_amd import inference_server as srv
    from matrix0_amd.backend import M0Backend
    cfg, sd, x, *_ = load_net_golden("gn_silu_preact")
    sd_np = {k: (v.numpy() if hasattr(v, "numpy") else np.asarray(v)) for k, v in sd.items()}
    ctx = mp.get_context("spawn")
    res = [srv.setup_shared_memory_for_worker(w, int(cfg.get("planes", 19)), 4672, 16) for w in range(2)]
    for r in res:
        r["request_event"], r["response_event"] = ctx.Event(), ctx.Event()
    stop, ready = ctx.Event(), ctx.Event()
    p = ctx.Process(target=srv.run_inference_server, args=("cuda:0", cfg, sd_np, stop, ready, res))
    p.start()
    try:
        assert ready.wait(timeout=300)
        local = M0Backend.from_state_dict(cfg, sd)
        xs = np.asarray(x, np.float32)
        n = min(len(xs), 6)
        want_p, want_v = local.infer_np(xs[:n])
        for w in range(2):
            got_p, got_v = srv.InferenceClient(res[w]).infer_np(xs[:n])
            assert np.array_equal(got_p, want_p) and np.array_equal(got_v, want_v)
        one_p, one_v = srv.InferenceClient(res[0]).infer_np(xs[0])
        assert one_p.shape == (1, 4672) and np.array_equal(one_p[0], want_p[0])
        local.close()
    finally:
        stop.set()
        p.join(timeout=60)
        if p.is_alive():
            p.terminate()
    assert p.exitcode == 0
