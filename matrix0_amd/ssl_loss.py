"""SSL loss heads on the host: PolicyValueNet.get_enhanced_ssl_loss / _compute_task_loss of the reference
(azchess/model/resnet.py:892-1130, called from training/train.py:285-305) restated for the HIP network's head OUTPUTS.

The reference computes the heads inside the loss (features -> ssl_heads[task]); here the heads already ran on the device
(m0_net_infer's `ssl` output / M0Backend.infer_np_ssl), so the loss is a function of (head outputs, targets):

    piece    13-way cross-entropy over the 64 squares of every position, mean over positions x squares; targets one-hot
             [B,13,8,8] (arg-max taken, resnet.py:930-933) or class indices [B,8,8], clamped to 0..12
    threat / pin / fork   binary cross-entropy with logits, targets clamped to [0, 1], mean (resnet.py:1062-1076)
    control  3-way cross-entropy, targets -1 / 0 / +1 -> classes 0 / 1 / 2 by the thresholds -0.5 / +0.5 (:1078-1094)
    total  = piece + sum_t ssl_<t>_weight * loss_t over the tasks that are enabled (cfg.ssl_tasks), have a target and a head;
             a task whose loss is not finite or not > 0 is left out (:961-964, :988-996)

Monitoring-side code (no gradients: training itself stays in the reference's trainer); pinned by tests/golden/ssl_loss.npz,
which the reference module produced (tools/gen_golden_ssl_loss.py)."""
from __future__ import annotations

from typing import Dict, Iterable, Mapping, Optional

import numpy as np

ADVANCED = ("threat", "pin", "fork", "control")


def _log_softmax(z: np.ndarray) -> np.ndarray:
    z = z.astype(np.float32)
    m = z.max(axis=-1, keepdims=True)
    e = z - m
    return e - np.log(np.exp(e).sum(axis=-1, keepdims=True, dtype=np.float32))


def _cross_entropy_mean(logits_nchw: np.ndarray, classes: np.ndarray) -> float:
    c = logits_nchw.shape[1]
    flat = np.transpose(logits_nchw, (0, 2, 3, 1)).reshape(-1, c)
    ls = _log_softmax(flat)
    idx = classes.reshape(-1).astype(np.int64)
    return float(-ls[np.arange(flat.shape[0]), idx].sum(dtype=np.float32) / np.float32(flat.shape[0]))


def _bce_with_logits_mean(logits: np.ndarray, targets: np.ndarray) -> float:
    z = logits.reshape(-1).astype(np.float32)
    t = np.clip(targets.reshape(-1).astype(np.float32), 0.0, 1.0)
    # max(z, 0) - z t + log(1 + exp(-|z|)): the numerically stable form torch uses
    loss = np.maximum(z, 0) - z * t + np.log1p(np.exp(-np.abs(z)))
    return float(loss.sum(dtype=np.float32) / np.float32(z.shape[0]))


def task_loss(task: str, output: np.ndarray, targets: np.ndarray) -> Optional[float]:
    """_compute_task_loss (resnet.py:1058-1130) / the piece branch of get_enhanced_ssl_loss; None = wrong head shape."""
    output = np.asarray(output, np.float32)
    targets = np.asarray(targets)
    if output.ndim != 4:
        return None
    if task == "piece":
        if targets.ndim == 4:
            targets = np.argmax(targets, axis=1)
        return _cross_entropy_mean(output, np.clip(targets.astype(np.int64), 0, 12)) if output.shape[1] == 13 else None
    if task in ("threat", "pin", "fork"):
        return _bce_with_logits_mean(output, targets) if output.shape[1] == 1 else None
    if task == "control":
        if output.shape[1] != 3:
            return None
        t = targets.reshape(-1).astype(np.float32)
        cls = np.where(t < -0.5, 0, np.where(t > 0.5, 2, 1))
        return _cross_entropy_mean(output, cls)
    return None


def enhanced_ssl_loss(outputs: Mapping[str, np.ndarray], targets: Mapping[str, np.ndarray], ssl_tasks: Iterable[str] = ("piece",),
                      weights: Optional[Mapping[str, float]] = None) -> Dict[str, float]:
    """{"total": ..., "<task>": weighted term, ...} for head outputs [B,C,8,8] and targets as the self-play shards store them
    (ssl_piece [B,13,8,8], ssl_threat / ssl_pin / ssl_fork / ssl_control [B,8,8])."""
    ssl_tasks = list(ssl_tasks)
    weights = dict(weights or {})
    out: Dict[str, float] = {}
    total = np.float32(0.0)
    if "piece" in targets and "piece" in outputs and "piece" in ssl_tasks:
        v = task_loss("piece", outputs["piece"], targets["piece"])
        if v is not None and np.isfinite(v) and v > 0:
            out["piece"] = float(v)
            total = np.float32(total + np.float32(v))
    for t in ADVANCED:
        if t in targets and t in ssl_tasks and t in outputs:
            v = task_loss(t, outputs[t], targets[t])
            if v is not None and np.isfinite(v) and v > 0:
                w = np.float32(float(weights.get(t, 1.0)) * np.float32(v))
                out[t] = float(w)
                total = np.float32(total + w)
    out["total"] = float(total)
    return out
