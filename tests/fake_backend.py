"""Deterministic stand-in for M0Backend in the CPU test of the shared-memory inference server."""
import numpy as np


class FakeBackend:
    def __init__(self, cfg, sd, device_index=0):
        self.policy_size = int(cfg.get("policy_size", 4672))
        self.calls = 0

    def infer_np(self, x):
        self.calls += 1
        s = x.reshape(x.shape[0], -1).sum(axis=1).astype(np.float32)
        policy = np.outer(s, np.arange(self.policy_size, dtype=np.float32) / self.policy_size).astype(np.float32)
        value = np.tanh(s / 100.0).astype(np.float32)
        if np.any(s > 1e6):
            raise ValueError("network produced non-finite policy logits")
        return policy, value

    def close(self):
        pass


def make(cfg, sd, device_index=0):
    return FakeBackend(cfg, sd, device_index)
