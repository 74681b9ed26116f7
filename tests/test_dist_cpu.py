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
    # GEMM weights travel as fp16 (the values the kernels use), everything else as fp32
    shapes = weights.param_shapes(CFG)
    ok = set(got) == set(ref) and all(
        torch.equal(got[k], ref[k].half().float() if m0dist.is_gemm_weight(k, shapes[k]) else ref[k]) for k in ref)
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


def test_workers_with_one_config_land_on_different_gpus_and_games():
    """The orchestrator gives every worker the SAME cfg_dict (orchestrator.py:490-496): placement comes from proc_id."""
    from matrix0_amd.selfplay import worker_placement
    eng_cfg = {}
    spots = [worker_placement(i, 32, eng_cfg, n_devices=8) for i in range(8)]
    assert [d for d, _ in spots] == list(range(8))
    assert [f for _, f in spots] == [32 * i for i in range(8)]
    assert worker_placement(9, 32, eng_cfg, n_devices=8) == (1, 288)            # more workers than GPUs: round-robin
    assert worker_placement(3, 32, {"device_index": 5, "first_game_index": 1000}, n_devices=8) == (5, 1000)
    assert worker_placement(3, 32, eng_cfg, n_devices=0)[0] == 0                  # no GPU visible: index 0, creation fails loudly later


def test_worker_returns_at_once_for_zero_games_and_rejects_unimplemented_sections():
    from matrix0_amd import selfplay
    assert selfplay.selfplay_worker(0, {"model": {}}, None, 0) is None            # `for g in range(0)`: nothing to play
    assert selfplay.selfplay_worker(0, {"model": {}}, None, -3) is None
    with pytest.raises(NotImplementedError, match="polyglot"):
        selfplay.check_unsupported_sections({"openings": {"polyglot": "book.bin", "max_plies": 8}})
    with pytest.raises(NotImplementedError, match="tablebases"):
        selfplay.check_unsupported_sections({"tablebases": {"enabled": True, "path": "tb"}})
    selfplay.check_unsupported_sections({"selfplay": {"book_path": "openings.pgn"}})     # PGN books are read (matrix0_amd/pgn_book.py)
    selfplay.check_unsupported_sections({"openings": {"polyglot": "", "max_plies": 0}, "tablebases": {"enabled": False},
                                         "selfplay": {"book_path": "x.pgn"}, "engine": {"opening_fens": ["8/8/8/8/8/8/8/K1k5 w - - 0 1"]}})
