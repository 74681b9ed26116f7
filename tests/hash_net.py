"""Bit-reproducible stand-in evaluator for golden search traces.

Every output is a pure function of the input planes computed with 64-bit integer hashing and ONE float32
rounding per value, so the same logits come out on any machine (the float64-GEMM FakeNet of fake_net.py is only
reproducible on one machine).  tools/gen_golden_mcts.py runs the reference's MCTS against this evaluator; the parity
tests feed the same evaluator to the oracle and to the HIP engine.

  logits[j] = ((mix64(h ^ (j+1)*GOLDEN) >> 40) / 2^24 - 0.5) * sharp          (float32, uniform in +-sharp/2)
  value     = sign * ((mix64(h2) >> 40) / 2^24 - 0.5) * 2 * vscale + vbias*sign
     h  = position hash over all 19 planes,
     h2 = position hash over the planes without the side-to-move plane,
     sign = +1 white to move / -1 black to move when `stm_oriented` (a side-to-move oriented value head: the same
            position with the other side to move gets the opposite value, which is what the reference's
            _detect_value_from_white probe looks for, selfplay/internal.py:203-243), else +1.
"""
import numpy as np

MASK = np.uint64(0xFFFFFFFFFFFFFFFF)
GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def _mix64(x):
    x = x.astype(np.uint64, copy=True)
    x ^= x >> np.uint64(30); x *= np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(27); x *= np.uint64(0x94D049BB133111EB)
    x ^= x >> np.uint64(31)
    return x


_J = (np.arange(1, 4673, dtype=np.uint64) * GOLDEN)
_W = _mix64(np.arange(1, 19 * 64 + 1, dtype=np.uint64) * np.uint64(0xD6E8FEB86659FD93)) | np.uint64(1)
_W_NOSTM = _W.copy()
_W_NOSTM[12 * 64:13 * 64] = 0


class HashNet:
    def __init__(self, seed=0, sharp=8.0, vscale=0.9, vbias=0.0, stm_oriented=True, poison=False, boost_white=None,
                 boost_black=None):
        self.seed = np.uint64(seed)
        self.sharp = np.float32(sharp)
        self.vscale = np.float32(vscale)
        self.vbias = np.float32(vbias)
        self.stm_oriented = stm_oriented
        self.poison = poison          # some positions get one non-finite logit (Node._expand's uniform fallback)
        self.calls = 0
        self.boost = [[(int(i), np.float32(b)) for i, b in (bb or [])] for bb in (boost_black, boost_white)]   # [black, white]

    def infer_np(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.ndim == 3:
            x = x[None]
        B = x.shape[0]
        self.calls += B
        with np.errstate(over="ignore"):
            words = x.reshape(B, -1).view(np.uint32).astype(np.uint64)
            h = _mix64((words * _W[None, :]).sum(axis=1, dtype=np.uint64) + self.seed * GOLDEN)
            h2 = _mix64((words * _W_NOSTM[None, :]).sum(axis=1, dtype=np.uint64) + (self.seed + np.uint64(77)) * GOLDEN)
            k = (_mix64(h[:, None] ^ _J[None, :]) >> np.uint64(40)).astype(np.float32)       # 24-bit integers: exact
        logits = (k * np.float32(1.0 / 16777216.0) - np.float32(0.5)) * self.sharp
        kv = (h2 >> np.uint64(40)).astype(np.float32)
        base = (kv * np.float32(1.0 / 16777216.0) - np.float32(0.5)) * np.float32(2.0) * self.vscale + self.vbias
        if self.stm_oriented:
            white = x[:, 12, 0, 0] > 0.5
            value = np.where(white, base, -base).astype(np.float32)
        else:
            value = base.astype(np.float32)
        if self.boost[0] or self.boost[1]:
            wtm = x[:, 12, 0, 0] > 0.5
            for side in (0, 1):
                rows = np.nonzero(wtm == bool(side))[0]
                for i, b in self.boost[side]:
                    logits[rows, i] = logits[rows, i] + b
        if self.poison:
            sel = (h % np.uint64(5)).astype(np.int64)
            for i in range(B):
                if sel[i] == 0:
                    logits[i, 4671] = np.inf
                elif sel[i] == 1:
                    logits[i, 17] = np.nan
                elif sel[i] == 2:
                    logits[i, 2300] = -np.inf
        return np.ascontiguousarray(logits, dtype=np.float32), value
