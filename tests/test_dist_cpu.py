"""N>1 path on CPU: two processes over gloo exercise the game sharding, the weight broadcast and the clock/counter
reduction that bench.py and the multi-GPU worker launcher use (RCCL in production)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from matrix0_amd import dist as m0dist
from matrix0_amd import weights

CFG = dict(planes=19, channels=32, blocks=2, attention_heads=2, norm="group", activation="silu", preact=True,
           policy_factor_rank=8, self_supervised=True, ssl_tasks=["piece"])


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sd = weights.random_state_dict(CFG, seed=5) if rank == 0 else None
    got = m0dist.broadcast_state_dict(sd, CFG, src=0)
    ref = weights.random_state_dict(CFG, seed=5)
    ok = all(torch.equal(got[k], ref[k]) for k in ref) and set(got) == set(ref)
    first, n = m0dist.shard_games(11, rank, world)
    tmax, tot = m0dist.reduce_clock_and_counters(1.0 + rank, np.array([n, 10.0 * (rank + 1)]))
    torch.save({"ok": ok, "first": first, "n": n, "tmax": tmax, "tot": tot}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo(tmp_path):
    world = 2
    port = 29000 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(tmp_path, f"r{i}.pt"), weights_only=False) for i in range(world)]
    assert all(x["ok"] for x in r)
    assert (r[0]["first"], r[0]["n"]) == (0, 6) and (r[1]["first"], r[1]["n"]) == (6, 5)
    for x in r:
        assert x["tmax"] == 2.0 and x["tot"].tolist() == [11.0, 30.0]


def test_shard_games_partitions_exactly():
    for total in (0, 1, 7, 256, 2048):
        for world in (1, 2, 3, 8):
            spans = [m0dist.shard_games(total, r, world) for r in range(world)]
            assert sum(n for _, n in spans) == total
            pos = 0
            for first, n in spans:
                assert first == pos
                pos += n
