"""Inference-backend seam: a drop-in for the object the reference passes as
``MCTS(cfg, model, device, inference_backend=obj)`` (azchess/mcts.py:270-283), whose only
method is ``infer_np`` (azchess/mcts.py:618-621, 1021-1023; client side
azchess/selfplay/inference.py:585-645).

    backend = M0Backend.from_state_dict(model_cfg_dict, state_dict)      # weights: torch or numpy
    policy_logits, value = backend.infer_np(np.float32[B,19,8,8])       # -> f32 [B,4672], f32 [B]

All arithmetic runs in libm0engine.so on the MI355X; torch is used only to read
checkpoints (weight I/O).
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import Dict, Optional, Tuple

import numpy as np

from . import _lib


class M0Backend:
    def __init__(self, model_cfg: dict, device_index: int = 0):
        self._L = _lib.lib()
        self.model_cfg = dict(model_cfg)
        self._cfg = _lib.net_cfg_from_dict(model_cfg)
        self._h = self._L.m0_net_create(C.byref(self._cfg), int(device_index))
        if not self._h:
            raise RuntimeError(f"m0_net_create failed: {_lib.last_error()}")
        self._finalized = False
        self.ssl_tasks = [t for t in _lib.SSL_ORDER if self._cfg.ssl_tasks & _lib.SSL_BITS[t]] \
            if self._cfg.self_supervised else []

    # ---- weight I/O -------------------------------------------------------------------------
    def load_state_dict(self, sd: Dict[str, "np.ndarray"]) -> None:
        """Feed every state-dict entry (reference key names) then repack.  Accepts torch tensors
        or numpy arrays.  Mirrors load_state_dict(strict=False): unknown keys are ignored by
        the engine; missing keys raise at finalize (the reference silently re-initialises them,
        resnet.py:1418-1440 -- a self-play run on half-random weights is not something to
        reproduce silently)."""
        for k, v in sd.items():
            if hasattr(v, "detach"):
                v = v.detach().cpu().float().numpy()
            a = np.ascontiguousarray(np.asarray(v), dtype=np.float32)
            shape = (C.c_int64 * max(1, a.ndim))(*a.shape)
            _lib.check(self._L.m0_net_load_weight(self._h, k.encode(), a.ctypes.data_as(C.c_void_p), 0, shape, a.ndim),
                       f"m0_net_load_weight({k})")
        _lib.check(self._L.m0_net_finalize(self._h), "m0_net_finalize")
        self._finalized = True

    @classmethod
    def from_state_dict(cls, model_cfg: dict, sd, device_index: int = 0) -> "M0Backend":
        b = cls(model_cfg, device_index)
        b.load_state_dict(sd)
        return b

    @classmethod
    def from_checkpoint(cls, model_cfg: dict, ckpt_path: str, device_index: int = 0) -> "M0Backend":
        """Checkpoint dict layout of the reference: prefer model_ema, then model, then
        model_state_dict, then the raw dict (selfplay/internal.py:172-174, orchestrator.py:373-388)."""
        import torch  # weight I/O only
        state = torch.load(ckpt_path, map_location="cpu", weights_only=False)
        sd = state
        if isinstance(state, dict):
            for key in ("model_ema", "model", "model_state_dict"):
                if key in state and isinstance(state[key], dict):
                    sd = state[key]
                    break
        return cls.from_state_dict(model_cfg, sd, device_index)

    # ---- the seam ---------------------------------------------------------------------------
    def infer_np(self, arr: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        p, v, _ = self._infer(arr, False)
        return p, v

    def infer_np_ssl(self, arr: np.ndarray):
        """(policy, value, {task: f32 [B,ch,8,8]}) -- forward(return_ssl=True), resnet.py:738-745."""
        return self._infer(arr, True)

    def _infer(self, arr: np.ndarray, want_ssl: bool):
        if not self._finalized:
            raise RuntimeError("weights not loaded")
        a = np.asarray(arr)
        if a.ndim == 3:
            a = a[None]
        if a.ndim != 4 or a.shape[1:] != (self._cfg.planes, 8, 8):
            raise ValueError(f"expected input [B,{self._cfg.planes},8,8], got {a.shape}")  # mcts.py:1030-1040
        a = np.ascontiguousarray(a, dtype=np.float32)
        B = a.shape[0]
        pol = np.empty((B, _lib.POLICY_SIZE), dtype=np.float32)
        val = np.empty((B,), dtype=np.float32)
        sslc = self._L.m0_net_ssl_channels(self._h)
        ssl = np.empty((B, sslc, 8, 8), dtype=np.float32) if (want_ssl and sslc > 0) else None
        rc = self._L.m0_net_infer(self._h, a.ctypes.data_as(C.c_void_p), B, pol.ctypes.data_as(C.c_void_p),
                                  val.ctypes.data_as(C.c_void_p),
                                  ssl.ctypes.data_as(C.c_void_p) if ssl is not None else None)
        _lib.check(rc, "m0_net_infer")
        out = None
        if ssl is not None:
            out, off = {}, 0
            for t in self.ssl_tasks:
                n = _lib.SSL_CH[t]
                out[t] = ssl[:, off:off + n]
                off += n
        return pol, val, out

    # ---- measurement helpers ----------------------------------------------------------------
    def param_count(self) -> int:
        return int(self._L.m0_net_param_count(self._h))

    def flops_per_position(self, with_ssl: bool = False) -> float:
        return float(self._L.m0_net_flops_per_position(self._h, int(with_ssl)))

    def bench_forward(self, batch: int, iters: int, with_ssl: bool = False) -> float:
        ms = C.c_float(0)
        _lib.check(self._L.m0_net_bench_forward(self._h, int(batch), int(iters), int(with_ssl), C.byref(ms)),
                   "m0_net_bench_forward")
        return float(ms.value)

    def profile_enable(self, on: bool = True) -> None:
        _lib.check(self._L.m0_net_profile_enable(self._h, int(on)), "m0_net_profile_enable")

    def profile_get(self, reset: bool = False):
        """(ms, algorithmic FLOP, launches) of the 3x3 conv kernel, HIP events on its launch stream."""
        ms, fl, n = C.c_double(0), C.c_double(0), C.c_int64(0)
        tms, tn = C.c_double(0), C.c_int64(0)
        _lib.check(self._L.m0_net_profile_get_tail(self._h, C.byref(tms), C.byref(tn)), "m0_net_profile_get_tail")
        self.last_tail_profile = (tms.value, tn.value)      # (ms, launches) of the convs with a fused block tail
        _lib.check(self._L.m0_net_profile_get(self._h, C.byref(ms), C.byref(fl), C.byref(n), int(reset)), "m0_net_profile_get")
        return ms.value, fl.value, n.value

    @property
    def handle(self):
        return self._h

    def close(self):
        if getattr(self, "_h", None):
            self._L.m0_net_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
