"""Deterministic stand-in evaluator for search parity tests: a fixed random projection of the planes.
`sharp` controls how peaked the policy is (flat policies trigger the reference's entropy noise)."""
import numpy as np


class FakeNet:
    def __init__(self, seed=0, sharp=8.0):
        rng = np.random.default_rng(seed)
        # float64 weights + float64 accumulation, rounded once to float32: the result for a position does not
        # depend on the batch it is evaluated in (float32 BLAS results do, by an ulp, via blocking)
        self.wp = (rng.standard_normal((19 * 64, 4672)) * sharp / 8.0).astype(np.float32).astype(np.float64)
        self.wv = (rng.standard_normal((19 * 64,)) * 0.2).astype(np.float32).astype(np.float64)
        self.calls = 0

    def infer_np(self, x):
        x = np.asarray(x, dtype=np.float32)
        if x.ndim == 3:
            x = x[None]
        f = x.reshape(x.shape[0], -1).astype(np.float64)
        self.calls += x.shape[0]
        return (f @ self.wp).astype(np.float32), np.tanh(f @ self.wv).astype(np.float32)
