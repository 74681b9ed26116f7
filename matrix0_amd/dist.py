"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in
CPU tests).  Games are independent units, so the only collectives are the optional weight broadcast at start-up
 and the scalar reductions of the benchmark clock/counters -- nothing on the data path."""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

from .weights import param_shapes


def shard_games(total_games: int, rank: int, world: int) -> Tuple[int, int]:
    """(first_game_index, n_games) of this rank: contiguous blocks, remainder to the low ranks; game indices
    (hence the per-game random streams, seed + index) are global, so a run is reproducible for any world size."""
    base, rem = divmod(int(total_games), int(world))
    n = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, n


def is_gemm_weight(name: str, shape) -> bool:
    """Tensors the engine packs as fp16 MFMA operands (csrc/net.hip pack_gemm / pack_attn_block): conv and fully
    connected weight matrices.  Everything else (norm gains and biases, squeeze-excite, positional encoding, relative
    position bias, the logit scale) is consumed in fp32."""
    return name.endswith(".weight") and len(shape) >= 2 and ".se_fc" not in name


def broadcast_state_dict(sd: Optional[Dict[str, torch.Tensor]], model_cfg: dict, src: int = 0,
                         device: Optional[torch.device] = None) -> Dict[str, torch.Tensor]:
    """Rank `src` holds the state dict; every rank returns the same tensors.  Two flat blobs in two broadcasts: the GEMM
    weights as fp16 -- exactly the values the kernels compute with, 114 MB for R24-320 -- and the small fp32 remainder
    (0.5 MB).  With backend "nccl" the blobs live on the GPU and move over xGMI (RCCL); the engine's weight loader takes
    host memory (m0_net_load_weight), so each rank copies its blob down once afterwards."""
    shapes = param_shapes(model_cfg)
    names = list(shapes.keys())
    half = [k for k in names if is_gemm_weight(k, shapes[k])]
    full = [k for k in names if not is_gemm_weight(k, shapes[k])]
    numel = {k: (int(np.prod(shapes[k])) if len(shapes[k]) else 1) for k in names}
    dev = device if device is not None else torch.device("cpu")
    blob16 = torch.empty(sum(numel[k] for k in half), dtype=torch.float16, device=dev)
    blob32 = torch.empty(sum(numel[k] for k in full), dtype=torch.float32, device=dev)
    if dist.get_rank() == src:
        if sd is None:
            raise ValueError("source rank needs the state dict")
        blob16.copy_(torch.cat([sd[k].reshape(-1).half() for k in half]).to(dev))
        blob32.copy_(torch.cat([sd[k].reshape(-1).float() for k in full]).to(dev))
    dist.broadcast(blob16, src=src)
    dist.broadcast(blob32, src=src)
    out = {}
    for blob, keys in ((blob16.cpu().float(), half), (blob32.cpu(), full)):
        off = 0
        for k in keys:
            out[k] = blob[off:off + numel[k]].reshape(shapes[k]).clone()
            off += numel[k]
    return {k: out[k] for k in names}


def reduce_clock_and_counters(seconds: float, counters: np.ndarray, device: Optional[torch.device] = None):
    """(max over ranks of seconds, sum over ranks of counters): the benchmark contract (value = all ranks' units /
    slowest rank's time)."""
    dev = device if device is not None else torch.device("cpu")
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=dev)
    c = torch.tensor(np.asarray(counters, dtype=np.float64), device=dev)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), c.cpu().numpy()


# ---- the same broadcast without torch.distributed: the C-ABI entry points of csrc/capi_dist.hip (RCCL loaded by the library) ----
class CAbiDist:
    """One communicator per process (m0_dist_create).  `id128` comes from `CAbiDist.unique_id()` on rank 0 and reaches the other
    ranks by any out-of-band means -- a file on a shared path in `CAbiDist.from_file`."""

    def __init__(self, rank: int, world: int, id128: bytes, device_index: int = 0):
        import ctypes as C
        from . import _lib
        L = _lib.lib()
        L.m0_dist_create.restype = C.c_void_p
        L.m0_dist_create.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int]
        L.m0_dist_destroy.argtypes = [C.c_void_p]
        L.m0_net_broadcast_weights.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        if len(id128) != 128:
            raise ValueError("id128 must be the 128 bytes of m0_dist_unique_id")
        self._L = L
        self._h = L.m0_dist_create(int(rank), int(world), bytes(id128), int(device_index))
        if not self._h:
            raise RuntimeError("m0_dist_create: " + L.m0_last_error().decode())
        self.rank, self.world = int(rank), int(world)

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C
        from . import _lib
        L = _lib.lib()
        buf = C.create_string_buffer(128)
        _lib.check(L.m0_dist_unique_id(buf), "m0_dist_unique_id")
        return buf.raw

    @classmethod
    def from_file(cls, path: str, rank: int, world: int, device_index: int = 0, timeout_s: float = 120.0) -> "CAbiDist":
        """Rank 0 writes the id to `path` (atomically), the others wait for it."""
        import os
        import time
        if rank == 0:
            tmp = path + ".tmp"
            with open(tmp, "wb") as f:
                f.write(cls.unique_id())
            os.replace(tmp, path)
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > timeout_s:
                raise TimeoutError(f"no communicator id at {path}")
            time.sleep(0.05)
        with open(path, "rb") as f:
            return cls(rank, world, f.read(), device_index)

    def broadcast_weights(self, backend, root: int = 0) -> None:
        """Every packed device buffer of `backend` (an M0Backend whose weights are finalized) becomes the root's."""
        from . import _lib
        _lib.check(self._L.m0_net_broadcast_weights(backend._h, self._h, int(root)), "m0_net_broadcast_weights")

    def close(self) -> None:
        if self._h:
            self._L.m0_dist_destroy(self._h)
            self._h = None
