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
