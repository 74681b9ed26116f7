"""Shared-memory inference protocol (azchess/selfplay/inference.py:18-35, 101-575, 585-680): a server process and
several clients exchanging real shared-memory tensors and multiprocessing Events, with a deterministic backend."""
import multiprocessing as mp
import threading

import numpy as np
import pytest

from matrix0_amd import inference_server as srv


def _expected(x, policy_size=4672):
    s = x.reshape(x.shape[0], -1).sum(axis=1).astype(np.float32)
    return (np.outer(s, np.arange(policy_size, dtype=np.float32) / policy_size).astype(np.float32),
            np.tanh(s / 100.0).astype(np.float32))


def test_resource_layout():
    r = srv.setup_shared_memory_for_worker(0, 19, 4672, 96)
    assert set(r) == {"request_tensor", "response_policy_tensor", "response_value_tensor", "request_event", "response_event",
                      "batch_size_tensor"}
    assert tuple(r["request_tensor"].shape) == (96, 19, 8, 8) and r["request_tensor"].is_shared()
    assert tuple(r["response_policy_tensor"].shape) == (96, 4672) and tuple(r["response_value_tensor"].shape) == (96, 1)
    assert tuple(r["batch_size_tensor"].shape) == (1,) and str(r["batch_size_tensor"].dtype) == "torch.int32"


def test_server_and_clients_over_shared_memory():
    ctx = mp.get_context("spawn")
    workers = 3
    res = [srv.setup_shared_memory_for_worker(w, 19, 4672, 32) for w in range(workers)]
    stop, ready = ctx.Event(), ctx.Event()
    # Events created by setup_shared_memory_for_worker come from the default context; recreate them in the spawn context
    for r in res:
        r["request_event"], r["response_event"] = ctx.Event(), ctx.Event()
    p = ctx.Process(target=srv.run_inference_server,
                    args=("cuda:0", {"policy_size": 4672}, {}, stop, ready, res), kwargs={"backend_factory": "tests.fake_backend:make"})
    p.start()
    try:
        assert ready.wait(timeout=60)
        errors = []

        def client(w):
            try:
                rng = np.random.default_rng(w)
                c = srv.InferenceClient(res[w])
                for n in (1, 7, 32, 3):
                    x = rng.random((n, 19, 8, 8)).astype(np.float32)
                    p_, v_ = c.infer_np(x if n > 1 else x[0])
                    pe, ve = _expected(x)
                    assert p_.shape == (n, 4672) and v_.shape == (n,)
                    np.testing.assert_allclose(p_, pe, rtol=1e-6)
                    np.testing.assert_allclose(v_, ve, rtol=1e-6)
            except Exception as e:                      # noqa: BLE001
                errors.append((w, repr(e)))

        ts = [threading.Thread(target=client, args=(w,)) for w in range(workers)]
        [t.start() for t in ts]
        [t.join(timeout=120) for t in ts]
        assert not errors, errors
        with pytest.raises(ValueError):
            srv.InferenceClient(res[0]).infer_np(np.zeros((2, 3), np.float32))
    finally:
        stop.set()
        p.join(timeout=30)
        if p.is_alive():
            p.terminate()
    assert p.exitcode == 0


@pytest.mark.skipif(not __import__("os").path.isdir("/root/reference/azchess"), reason="the reference exists in the build container only")
def test_reference_inference_client_drives_our_server():
    """The reference's OWN `InferenceClient` and `setup_shared_memory_for_worker` (azchess/selfplay/inference.py:18-35, 585-680,
    imported where they lie through tools/refshim.py) against this package's `run_inference_server`: the unmodified Matrix0
    worker side of the protocol -- resource dict, batch-size tensor, event hand-shake, the (C,H,W) convenience form, a batch
    at the resource capacity -- gets its answers from our server process.  Build container only (no reference on the GPU box)."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import refshim
    refshim.install()
    import logging
    logging.disable(logging.CRITICAL)
    import azchess.selfplay.inference as rinf
    ctx = mp.get_context("spawn")
    workers = 2
    res = [rinf.setup_shared_memory_for_worker(w, 19, 4672, 96) for w in range(workers)]
    assert set(res[0]) == set(srv.setup_shared_memory_for_worker(0, 19, 4672, 96))
    stop, ready = ctx.Event(), ctx.Event()
    for r in res:                                   # Events must come from the context that spawns the server process
        r["request_event"], r["response_event"] = ctx.Event(), ctx.Event()
    p = ctx.Process(target=srv.run_inference_server,
                    args=("cuda:0", {"policy_size": 4672}, {}, stop, ready, res), kwargs={"backend_factory": "tests.fake_backend:make"})
    p.start()
    try:
        assert ready.wait(timeout=60)
        errors = []

        def client(w):
            try:
                rng = np.random.default_rng(10 + w)
                c = rinf.InferenceClient(res[w])
                for n in (1, 8, 96, 33, 1):
                    x = rng.random((n, 19, 8, 8)).astype(np.float32)
                    p_, v_ = c.infer_np(x if n > 1 else x[0])
                    pe, ve = _expected(x)
                    assert p_.shape == (n, 4672) and v_.shape == (n,)
                    np.testing.assert_allclose(p_, pe, rtol=1e-6)
                    np.testing.assert_allclose(v_, ve, rtol=1e-6)
            except Exception as e:                      # noqa: BLE001
                errors.append((w, repr(e)))

        ts = [threading.Thread(target=client, args=(w,)) for w in range(workers)]
        [t.start() for t in ts]
        [t.join(timeout=120) for t in ts]
        assert not errors, errors
    finally:
        logging.disable(logging.NOTSET)
        stop.set()
        p.join(timeout=30)
        if p.is_alive():
            p.terminate()
    assert p.exitcode == 0
