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


def broadcast_state_dict(sd: Optional[Dict[str, torch.Tensor]], model_cfg: dict, src: int = 0,
                         device: Optional[torch.device] = None) -> Dict[str, torch.Tensor]:
    """Rank `src` holds the state dict (fp32); every rank returns an identical copy.  One flat fp32 blob
    (230 MB for R24-320) in a single broadcast: on xGMI a ring/tree broadcast of that size is ~ms."""
    shapes = param_shapes(model_cfg)
    names = list(shapes.keys())
    sizes = [int(np.prod(shapes[k])) if len(shapes[k]) else 1 for k in names]
    dev = device if device is not None else torch.device("cpu")
    blob = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
    if dist.get_rank() == src:
        if sd is None:
            raise ValueError("source rank needs the state dict")
        blob.copy_(torch.cat([sd[k].reshape(-1).float() for k in names]).to(dev))
    dist.broadcast(blob, src=src)
    flat = blob.cpu()
    out, off = {}, 0
    for k, n in zip(names, sizes):
        out[k] = flat[off:off + n].reshape(shapes[k]).clone()
        off += n
    return out


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
