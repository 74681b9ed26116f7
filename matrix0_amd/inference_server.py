"""Shared-memory inference server speaking the reference's protocol (azchess/selfplay/inference.py:18-35,
101-575, 585-680), with the HIP network behind it: unmodified Matrix0 workers, arena workers or the web UI keep their
`InferenceClient` and get their evaluations from the MI355X.

Per worker one resource dict (inference.py:18-35):
    request_tensor          f32 [max_batch, planes, 8, 8]   shared memory, written by the client
    response_policy_tensor  f32 [max_batch, policy_size]    written by the server (raw logits)
    response_value_tensor   f32 [max_batch, 1]
    batch_size_tensor       i32 [1]
    request_event / response_event   multiprocessing.Event
Client (inference.py:585-680): copy the positions in, set batch_size, set request_event, wait for response_event
(timeout scaled with the batch, two retries), copy the answers out, clear response_event.  Server: set
`server_ready_event` once the weights are on the device; then, until `stop_event`: collect every worker whose
request_event is set, clear it, evaluate all their positions in ONE forward, write each worker's slice back, set its
response_event.  A forward that fails (e.g. non-finite outputs, which M0Backend.infer_np raises on) is logged and not
answered: the client times out and raises, as with the reference server.
"""
from __future__ import annotations

import importlib
import logging
import os
import time
from multiprocessing import Event
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

log = logging.getLogger(__name__)

DEFAULT_BACKEND_FACTORY = "matrix0_amd.backend:M0Backend.from_state_dict"


def setup_shared_memory_for_worker(worker_id: int, planes: int, policy_size: int, max_batch_size: int) -> Dict[str, Any]:
    """Same keys, shapes and dtypes as inference.py:18-35."""
    return {
        "request_tensor": torch.zeros((max_batch_size, planes, 8, 8), dtype=torch.float32).share_memory_(),
        "response_policy_tensor": torch.zeros((max_batch_size, policy_size), dtype=torch.float32).share_memory_(),
        "response_value_tensor": torch.zeros((max_batch_size, 1), dtype=torch.float32).share_memory_(),
        "request_event": Event(),
        "response_event": Event(),
        "batch_size_tensor": torch.tensor([0], dtype=torch.int32).share_memory_(),
    }


def _device_index(device) -> int:
    if isinstance(device, int):
        return device
    s = str(device)
    return int(s.split(":", 1)[1]) if ":" in s and s.split(":", 1)[1].isdigit() else 0


def _resolve(path: str):
    mod, _, attr = path.partition(":")
    obj = importlib.import_module(mod)
    for part in attr.split("."):
        obj = getattr(obj, part)
    return obj


def run_inference_server(device, model_cfg: dict, model_state_dict: Optional[Dict[str, Any]], stop_event: Any,
                         server_ready_event: Any, shared_memory_resources: List[Dict[str, Any]],
                         backend_factory: str = DEFAULT_BACKEND_FACTORY, poll_seconds: float = 0.001) -> None:
    """Same positional arguments as the reference's run_inference_server (inference.py:101-108).
    `backend_factory` ("module:callable", called as f(model_cfg, state_dict, device_index=...)) exists for tests."""
    factory = _resolve(backend_factory)
    if model_state_dict is None:
        raise ValueError("run_inference_server needs the weights (model_state_dict)")
    backend = factory(dict(model_cfg), model_state_dict, device_index=_device_index(device))
    server_ready_event.set()
    served = 0
    try:
        while not stop_event.is_set():
            ready = [i for i, res in enumerate(shared_memory_resources) if res["request_event"].is_set()]
            if not ready:
                time.sleep(poll_seconds)
                continue
            parts, owners = [], []
            for i in ready:
                res = shared_memory_resources[i]
                n = int(res["batch_size_tensor"].item())
                res["request_event"].clear()
                if n <= 0:                                   # spurious wake-up: ignored (inference.py:361-366)
                    continue
                cap = int(res["request_tensor"].shape[0])
                if n > cap:                                  # clamp to capacity (inference.py:367-374)
                    n = cap
                    res["batch_size_tensor"][0] = cap
                parts.append(res["request_tensor"][:n].numpy())
                owners.append((i, n))
            if not parts:
                continue
            x = np.ascontiguousarray(np.concatenate(parts, axis=0), dtype=np.float32)
            try:
                policy, value = backend.infer_np(x)
            except Exception as e:                           # not answered: the clients time out (reference behaviour)
                log.error("inference failed for %d positions of workers %s: %s", x.shape[0], [o for o, _ in owners], e)
                continue
            off = 0
            for i, n in owners:
                res = shared_memory_resources[i]
                res["response_policy_tensor"][:n] = torch.from_numpy(policy[off:off + n])
                res["response_value_tensor"][:n, 0] = torch.from_numpy(np.ascontiguousarray(value[off:off + n]))
                off += n
                res["response_event"].set()
            served += off
    finally:
        close = getattr(backend, "close", None)
        if close:
            close()
        log.info("inference server stopped after %d positions", served)


class InferenceClient:
    """Client side, for callers that do not bring the reference's (inference.py:585-680): same `infer_np`."""

    def __init__(self, resources: Dict[str, Any]):
        self.res = resources

    def infer_np(self, arr_batch: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        if arr_batch.ndim == 3:
            arr_batch = np.expand_dims(arr_batch, 0)
        if arr_batch.ndim != 4:
            raise ValueError("InferenceClient expects (B,C,H,W) or (C,H,W)")
        n = int(arr_batch.shape[0])
        if arr_batch.dtype != np.float32:
            arr_batch = arr_batch.astype(np.float32, copy=False)
        fast = os.environ.get("MATRIX0_FAST_TIMEOUTS", "1").lower() in ("1", "true", "yes")
        base = 5.0 if fast else 12.0
        timeout = base * (2.0 if n == 1 else 1.5 if n <= 8 else 1.0 if n <= 32 else 1.0 + n / 64.0)
        timeout = min(timeout, 15.0 if fast else 30.0)
        self.res["request_tensor"][:n] = torch.from_numpy(np.ascontiguousarray(arr_batch))
        self.res["batch_size_tensor"][0] = n
        self.res["request_event"].set()
        for attempt in range(3):
            if self.res["response_event"].wait(timeout=timeout):
                policy = self.res["response_policy_tensor"][:n].numpy().copy()
                value = self.res["response_value_tensor"][:n].numpy().copy()
                self.res["response_event"].clear()
                return policy, value.flatten()
            if attempt < 2:
                self.res["request_event"].clear()
                self.res["response_event"].clear()
                time.sleep(0.1)
                self.res["request_event"].set()          # ask again (the reference leaves the retry to its caller)
        raise TimeoutError(f"Inference timeout after {timeout}s for batch size {n}")
