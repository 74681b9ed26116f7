#!/usr/bin/env python3
"""Self-play throughput benchmark (BASELINE.json metric: self-play games/sec at 800 sims/move,
ResNet-24 "53M" (R24-320), 1/2/4/8 MI355X).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one searched ply for every resident game: ceil(sims / leaves) passes of the hot path, each pass =
for every resident game tree select up to `leaves` leaves (PUCT + virtual loss), encode them into the network input,
run the R24-320 forward on the whole batch (games x leaves positions), expand + back up; searches that complete
(800 simulations per move, +-5 % playout cap) play their move and start the next search.  `leaves` defaults to the
reference's own leaf batch, mcts.inference_batch_size = 96 (config.yaml:157): a ply is then 9 passes of up to 24 576
positions.  The timed region is EXACTLY `--steps` searched plies per resident game: passes are issued until the engine's
ply counter has advanced by steps x games (the games search in step, so the region starts and ends right after a round of
moves and the result does not depend on the window length).  Inputs are resident in HBM; weights are random-init R24-320
(synthetic: no checkpoint, no dataset).  One process per GPU; games shard across GPUs with no data-path collective
(RCCL is used only to broadcast the weights from rank 0), so scaling is "weak".

games/sec: games finished in the timed region / time when at least MIN_FINISHED games finish there; otherwise the
MEASURED plies/sec (searched and played plies counted by the engine) divided by the mean game length in plies from
profiles/game_length.json (measured by running whole generations of games to completion with this engine/config;
the basis is named in the JSON line).

`value` is measured with every leaf going through the network (`engine.eval_cache` off).  The drop-in worker's default has the
cache on -- a leaf reached more than once in a pass shares one batch row and a per-game cache serves positions evaluated in
earlier passes, ~5 % of the leaf evaluations; every simulation is still played and the games are bit-identical
(tests/test_eval_cache_gpu.py) -- so the same K plies are timed a second time with it on and reported in `eval_cache`
(`value_with_eval_cache`); `--no-eval-cache` skips that region, `--eval-cache-in-value` measures `value` itself with it on.
A third, supplementary region times the same K plies with `engine.tail_split = "halves"` (opt-in: every pass as two half batches
side by side; `half_split.value_with_half_split`; `--no-half-split-region` skips it).  Neither extra region enters `value` or `roofline`.

Extra objects in the JSON line:
  roofline      dominant kernel = 3x3 320->320 implicit-GEMM conv (MFMA-bound); achieved = algorithmic FLOP /
                launch time from HIP events around every launch on the launch stream, summed over the timed region
  cpu_baseline  BASELINE configs[0] on this box's host cores: W = cores / threads worker processes, each playing one
                self-play game at 64 sims/move with the CPU oracle (oracle/selfplay_ref.py + oracle/mcts_ref.py, reference
                search semantics: batches of <= 96 leaves, no virtual loss) and the torch-CPU fp32 R24-320 forward, for a
                bounded time; rank 0 at N=1 only, run BEFORE the GPU is touched so the two do not share the host
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

R24_320 = dict(planes=19, channels=320, blocks=24, attention_heads=20, policy_size=4672, norm="group",
               activation="silu", preact=True, droppath=0.1, policy_factor_rank=128, self_supervised=True,
               ssl_tasks=["piece", "threat", "pin", "fork", "control"])

# config.yaml:26-43, 128-163, num_simulations = 800 per the metric
SELFPLAY_CFG = {
    "seed": 1234,
    "mcts": {"cpuct": 2.5, "cpuct_start": 3.0, "cpuct_end": 2.0, "cpuct_plies": 40, "dirichlet_alpha": 0.3,
             "dirichlet_frac": 0.25, "dirichlet_plies": 30, "selection_jitter": 0.05, "fpu_reduction": 0.1,
             "draw_penalty": -0.05, "legal_softmax": True, "no_instant_backtrack": True, "value_from_white": False,
             "virtual_loss": 1.0, "playout_random_frac": 0.05, "enable_entropy_noise": True,
             "inference_batch_size": 96},
    "selfplay": {"num_simulations": 800, "max_game_len": 200, "min_resign_plies": 50, "resign_threshold": -0.85,
                 "opening_random_plies": 12, "temperature_start": 1.2, "temperature_end": 0.3, "temperature_moves": 40,
                 "draw": {"min_plies": 30, "window": 8, "min_unique": 4, "halfmove_cap": 100}},
}

PEAK_FP16_DENSE = 2.5e15          # MI355X dense fp16/bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
MIN_FINISHED = 32
DEFAULT_PLIES_PER_GAME = 150.0


def game_length_basis():
    p = os.path.join(ROOT, "profiles", "game_length.json")
    try:
        d = json.load(open(p))
        return float(d["mean_plies_per_game"]), f"profiles/game_length.json ({d.get('games', '?')} games)"
    except Exception:
        return DEFAULT_PLIES_PER_GAME, "default (no calibration file)"


def generation_passes_per_ply():
    """Passes of the hot path per searched ply and game over a WHOLE generation played to completion (games finish at different
    times and their searches drift apart, so a pass serves searches at different stages): tools/calibrate_game_length.py."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "game_length.json")))
        return d.get("passes_per_ply"), d.get("games_per_s_end_to_end")
    except Exception:
        return None, None


def host_cores() -> int:
    """CPUs this process may use: scheduler affinity, capped by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


CPU_CFG = {  # BASELINE configs[0]: 1 self-play game, 64 MCTS sims/move, random-init ResNet-24, device=cpu
    "seed": 1234, "mcts": dict(SELFPLAY_CFG["mcts"]),
    "selfplay": dict(SELFPLAY_CFG["selfplay"], num_simulations=64),
}


def _cpu_worker(idx: int, threads: int, seconds: float, out_q):
    """One reference-style worker process: one game at a time (selfplay/internal.py:326), 64 sims/move."""
    import numpy as np
    import torch
    torch.set_num_threads(threads)
    from matrix0_amd.weights import random_state_dict
    from oracle import net_ref, selfplay_ref
    sd = random_state_dict(R24_320, seed=0, varied=True)

    def infer(x):
        with torch.no_grad():
            p, v, _ = net_ref.forward(sd, R24_320, torch.from_numpy(np.ascontiguousarray(x)))
        return p.numpy(), v.numpy()

    t0 = time.perf_counter()
    out = selfplay_ref.play_game(CPU_CFG, infer, CPU_CFG["seed"], idx, use_tt=False, tree_reuse=False,
                                 virtual_loss_active=False, numerics="reference", value_from_white=False, max_seconds=seconds)
    dt = time.perf_counter() - t0
    out_q.put({"worker": idx, "plies": int(out["moves"]), "sims": int(sum(out["trace"]["sims"])), "evals": int(out["evals"]),
               "secs": dt, "finished": not out["timed_out"]})


def cpu_baseline(seconds: float, plies_per_game: float, threads: int = 4):
    import multiprocessing as mp
    cores = host_cores()
    threads = max(1, min(threads, cores))
    W = max(1, cores // threads)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    t0 = time.perf_counter()
    procs = [ctx.Process(target=_cpu_worker, args=(i, threads, seconds, q)) for i in range(W)]
    for p in procs:
        p.start()
    res = []
    deadline = t0 + seconds * 6 + 300          # a worker checks its own clock once per searched ply; first import can take minutes
    while len(res) < len(procs) and time.perf_counter() < deadline:
        try:
            res.append(q.get(timeout=2.0))
        except Exception:
            if all(p.exitcode is not None for p in procs) and q.empty():
                break                          # every worker is gone (a crash reports nothing): do not wait out the deadline
    for p in procs:                            # nothing of the baseline may still run when the GPU leg starts
        p.join(timeout=5)
        if p.is_alive():
            p.terminate()
            p.join(timeout=10)
            if p.is_alive():
                p.kill()
                p.join()
    wall = time.perf_counter() - t0
    if not res:
        return {"value": None, "unit": "games/s", "cores": cores, "kind": "port", "sample": "no worker reported"}
    span = max(r["secs"] for r in res)
    sims = sum(r["sims"] for r in res)
    plies = sum(r["plies"] for r in res)
    evals = sum(r["evals"] for r in res)
    sims_per_s = sims / span
    return {"value": sims_per_s / (800.0 * plies_per_game), "unit": "games/s", "cores": int(cores), "workers": len(res),
            "threads_per_worker": threads, "kind": "port", "sims_per_s": round(sims_per_s, 2), "evals_per_s": round(evals / span, 2),
            "plies_per_s_at_64_sims": round(plies / span, 4), "games_per_s_at_64_sims": round(plies / span / plies_per_game, 6),
            "sample": f"{len(res)} worker processes x {threads} torch threads on {cores} host cores, each one self-play game at "
                      f"64 sims/move (BASELINE configs[0]) for {seconds:.0f} s: {plies} plies, {sims} simulations, {evals} fp32 "
                      f"R24-320 evaluations in {span:.1f} s (wall incl. process start {wall:.1f} s); oracle = CPU restatement of the "
                      f"reference worker (tree-only: with its transposition table on the reference raises on the 2nd move); "
                      f"value = simulations/s / (800 x {plies_per_game:.1f} plies per game)"}


def main():
    # stdout carries exactly ONE line, the JSON result.  Native libraries write there too (RCCL prints a version banner at the
    # first communicator), so file descriptor 1 points at stderr for the whole run and the line goes to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed steps; one step = one searched ply per resident game")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--games", type=int, default=256, help="concurrent games per GPU (BASELINE configs[1])")
    ap.add_argument("--leaves", type=int, default=96,
                    help="leaves per tree and pass = mcts.inference_batch_size of the reference's config.yaml (96)")
    ap.add_argument("--sims", type=int, default=800)
    ap.add_argument("--ssl", action="store_true", help="run the 5 SSL heads in every evaluation (configs[3])")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--cpu-threads", type=int, default=4, help="torch threads per CPU-baseline worker process")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-eval-cache", action="store_true",
                    help="skip the second timed region (the same plies with engine.eval_cache on, reported in `eval_cache`)")
    ap.add_argument("--eval-cache-in-value", action="store_true",
                    help="measure `value` itself with engine.eval_cache on (the drop-in worker's default) instead of off")
    ap.add_argument("--streams", type=int, default=1,
                    help="independent engines (own network instance and HIP stream) sharing the games of a GPU; 2 gives "
                         "+3..4 %% games/s, but per-launch kernel timings then include the other stream's kernels, so the "
                         "roofline of the default run is taken with 1")
    ap.add_argument("--tail-split", action="store_true",
                    help="engine.tail_split: evaluate a pass as a whole number of workgroup rounds + a concurrent tail on a second "
                         "instance over the same weights (bit-identical games; +0.4..0.5 %% measured, and the event-bracketed conv "
                         "times then include the tail's workgroups, so the default bench keeps one forward per pass)")
    ap.add_argument("--no-half-split-region", action="store_true",
                    help="skip the supplementary third timed region (the same plies with engine.tail_split = 'halves', reported in `half_split`)")
    ap.add_argument("--half-split", action="store_true",
                    help="engine.tail_split = 'halves': a pass as two half batches on two instances over the same weights, side by side "
                         "(bit-identical games; +1.3 %% measured: one half's attention blocks run beside the other half's convs).  The "
                         "halves' kernels overlap, so the event-bracketed conv times -- and `roofline` -- no longer describe one kernel "
                         "on the whole chip; the default bench keeps one forward per pass")
    ap.add_argument("--cu-split", action="store_true",
                    help="experiment (with --streams 2): the two engines' streams run on complementary CU halves of every XCD")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world

    # CPU baseline first (rank 0, N=1): its worker processes are started before this process touches the GPU
    cpu = None
    if rank == 0 and world == 1 and args.gpus == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.cpu_seconds, game_length_basis()[0], args.cpu_threads)

    import numpy as np
    import torch
    import torch.distributed as dist
    from matrix0_amd.weights import random_state_dict
    from matrix0_amd.backend import M0Backend
    from matrix0_amd import engine as eng

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    # M0_FORCE_DIST=1: take the RCCL path (init, weight broadcast, reductions) even with one rank -- lets the
    # one-GPU box exercise exactly the code the multi-GPU runs use
    distributed = world > 1 or (os.environ.get("M0_FORCE_DIST") == "1" and "RANK" in os.environ)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    # weights: rank 0 synthesises, RCCL broadcast (the optional weight broadcast of the north star)
    from matrix0_amd import dist as m0dist
    sd = random_state_dict(R24_320, seed=0, varied=True) if rank == 0 else None
    if distributed:
        sd = m0dist.broadcast_state_dict(sd, R24_320, src=0, device=torch.device("cuda", local_rank))
    # --cu-split (experiment, with --streams 2): the two engines' streams get complementary halves of the CUs of every XCD
    # (hipExtStreamCreateWithCUMask through M0_NET_CU_MASK, read when a network is created)
    cu_masks = ["ffffffff,ffffffff,ffffffff,ffffffff,0,0,0,0", "0,0,0,0,ffffffff,ffffffff,ffffffff,ffffffff"] if args.cu_split else []

    def make_backend():
        if cu_masks:
            os.environ["M0_NET_CU_MASK"] = cu_masks.pop(0)
        try:
            return M0Backend.from_state_dict(R24_320, sd, device_index=local_rank)
        finally:
            os.environ.pop("M0_NET_CU_MASK", None)

    be = make_backend()
    flops_eval = be.flops_per_position(with_ssl=args.ssl)

    cfg_dict = json.loads(json.dumps(SELFPLAY_CFG))
    cfg_dict["selfplay"]["num_simulations"] = args.sims
    # unbounded mode (total_games=0): finished games are replaced forever, so every rank (and every pool stream) draws its
    # game indices from its own block of 2^24 -- no (seed, index) pair is ever played twice in one run
    first_index = rank * (1 << 24)
    def sync():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    substeps = -(-args.sims // args.leaves)          # passes of the hot path per searched ply (all games in step)

    def measure(eval_cache: bool, split=None):
        """One engine, warm-up, then the timed region of exactly `--steps` searched plies per resident game."""
        if split is None:
            split = "halves" if args.half_split else args.tail_split
        kw = dict(concurrent_games=args.games, total_games=0, first_game_index=first_index, leaves_per_step=args.leaves,
                  virtual_loss_active=True, ssl_in_forward=args.ssl, record_games=False, eval_cache=eval_cache,
                  tail_split=split)
        if args.streams > 1:
            made = [be]
            e = eng.SelfplayPool(lambda: made.pop() if made else make_backend(),
                                 cfg_dict, streams=args.streams, **kw)
        else:
            e = eng.SelfplayEngine(be, eng.selfplay_cfg_from_dict(cfg_dict, **kw))

        def run_plies(n):
            """Passes of the hot path until `n` more plies per resident game have been searched and played (counted by the
            engine).  The games search in step, so the region starts and ends right after a round of moves: the measured
            plies / time does not depend on where a fixed number of passes would have cut the searches."""
            target = e.stats()["plies"] + n * args.games
            passes = 0
            while e.stats()["plies"] < target and passes < 4 * n * substeps:
                e.step(1)
                passes += 1
            return passes

        if args.warmup > 0:
            run_plies(args.warmup)
        be.profile_enable(True)
        be.profile_get(reset=True)
        s0 = e.stats()
        sync()
        t0 = time.perf_counter()
        passes_timed = run_plies(args.steps)
        sync()
        dt = time.perf_counter() - t0
        s1 = e.stats()
        conv_ms, conv_flop, conv_launches = be.profile_get(reset=True)
        tail = getattr(be, "last_tail_profile", (0.0, 0))
        be.profile_enable(False)
        if args.streams > 1:
            for x in e.engines:
                x.close()                                   # the first backend is `be`: it stays open for the next measurement
            for b_ in e.backends[1:]:
                b_.close()
            e.engines, e.backends = [], []
        else:
            e.close()
        d = {k: s1[k] - s0[k] for k in ("steps", "evals", "sims", "plies", "games_finished", "ms_net", "ms_tree", "ms_host",
                                        "ms_total", "evals_cached", "rows_tail")}
        counters = np.array([d["evals"], d["plies"], d["games_finished"], conv_ms, conv_flop, conv_launches,
                             d["ms_net"], d["ms_tree"], d["ms_host"], d["sims"], passes_timed, d["evals_cached"], d["rows_tail"]],
                            dtype=np.float64)
        dt_max, tot = m0dist.reduce_clock_and_counters(dt, counters, device=torch.device("cuda", local_rank))
        return dt_max, [float(x) for x in tot], tail

    # `value` is measured with EVERY leaf going through the network (evaluation cache off: round 2's definition of the metric).
    # The product default (drop-in worker) has engine.eval_cache on; the same K plies are then timed a second time with it on
    # and reported beside `value` in `eval_cache` (--no-eval-cache: skip that; --eval-cache-in-value: `value` itself with it on)
    primary_cache = bool(args.eval_cache_in_value)
    dt_max, tot, tail_prof = measure(primary_cache)
    evals, plies, gfin, conv_ms, conv_flop, conv_launches, ms_net, ms_tree, ms_host, sims, passes_all, cached, rows_tail = tot
    passes = max(1.0, passes_all / args.gpus)           # passes per rank in the timed region
    second = None
    if not args.no_eval_cache and not primary_cache:
        dt2, tot2, _ = measure(True)
        second = {"dt": dt2, "evals": tot2[0], "plies": tot2[1], "gfin": tot2[2], "sims": tot2[9], "passes": tot2[10], "cached": tot2[11]}
    # third region (supplementary, like the second): the same K plies with engine.tail_split = "halves" -- every pass as two half
    # batches side by side (bit-identical games, tests/test_selfplay_gpu.py).  Kept OUT of `value` and of `roofline`: with two
    # forwards in flight a launch shares the chip with the other half's kernels and its event-bracketed time describes no kernel
    third = None
    if not args.no_half_split_region and not args.half_split and not args.tail_split and args.streams == 1:
        dt3, tot3, _ = measure(primary_cache, split="halves")
        third = {"dt": dt3, "evals": tot3[0], "plies": tot3[1], "gfin": tot3[2], "rows_tail": tot3[12]}

    if rank == 0:
        ppg, basis_src = game_length_basis()
        if gfin >= MIN_FINISHED * args.gpus:
            games_per_s = gfin / dt_max
            basis = f"{int(gfin)} games finished in the timed region"
        else:
            games_per_s = (plies / dt_max) / ppg
            basis = (f"{int(plies)} plies searched and played in the timed region / {ppg:.1f} plies per game [{basis_src}]; "
                     f"{int(gfin)} games finished")
        achieved = (conv_flop / (conv_ms * 1e-3)) if conv_ms > 0 else 0.0
        # launch mix: conv2 of every block (and the interaction conv) also carries the fused block tail
        tail_ms, tail_n = tail_prof
        plain_n = conv_launches - tail_n
        flop_per_launch = conv_flop / conv_launches if conv_launches else 0.0
        mix = {"with_fused_tail": {"launches": int(tail_n), "avg_launch_us": (tail_ms * 1e3 / tail_n) if tail_n else None,
                                   "achieved": (flop_per_launch / (tail_ms * 1e-3 / tail_n) / 1e12) if tail_n else None},
               "conv_only": {"launches": int(plain_n),
                             "avg_launch_us": ((conv_ms - tail_ms) * 1e3 / plain_n) if plain_n else None,
                             "achieved": (flop_per_launch / ((conv_ms - tail_ms) * 1e-3 / plain_n) / 1e12) if plain_n else None}}
        traffic = None
        try:        # PMC bytes per launch at the profiled batch, scaled to this run's boards per launch (HBM traffic is per board)
            ct = json.load(open(os.path.join(ROOT, "profiles", "conv_traffic.json")))
            boards_per_launch = (evals - rows_tail) / max(1.0, args.gpus * passes)      # the timed launches: the main instance's
            traffic = int(ct["hbm_bytes_per_launch"] * boards_per_launch / float(ct.get("boards_per_launch", 4096)))
        except Exception:
            pass
        out = {
            "metric": "self-play games/sec at 800 sims/move, ResNet-24 53M; 1/2/4/8 MI355X",
            "value": games_per_s, "unit": "games/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_max * 1e3 / max(1, args.steps), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"{args.games} concurrent games per GPU x {args.gpus} GPU, {args.sims} sims/move, "
                                   f"R24-320 (57.56M params, {flops_eval / 1e9:.4f} GFLOP/eval) fp16 MFMA, {args.leaves} leaves/tree/step"
                                   + (", 5 SSL heads in forward" if args.ssl else "")
                                   + (", repeated leaf positions served by engine.eval_cache" if primary_cache
                                      else ", every leaf evaluated by the network (engine.eval_cache off)"),
                       "games_basis": basis, "parallelism": f"games sharded x{args.gpus} (no data-path collective)"
                       + (f", {args.streams} engines / streams per GPU (kernel timings overlap)" if args.streams > 1 else "")},
            "evals_per_s": evals / dt_max, "sims_per_s": sims / dt_max, "plies_per_s": plies / dt_max, "games_finished": int(gfin),
            # engine.tail_split: a pass = a whole number of workgroup rounds on the main instance + the remaining < 1024 boards on a
            # second instance over the same weights, at the same time (results unchanged: the forward is bitwise batch invariant).
            # `roofline` is taken from the main instance's launches (their FLOP over their event time); the tail's convs run inside
            # those windows on CUs the main launches would have left idle.
            "tail_split": {"on": bool(args.tail_split or args.half_split), "mode": "halves" if args.half_split else ("tail" if args.tail_split else "off"),
                           "share_of_evaluations": rows_tail / max(1.0, evals)},
            # second timed region of the same K plies with the product default engine.eval_cache on (module docstring)
            "eval_cache": (None if second is None and not primary_cache else (
                {"in_value": True, "evaluations_from_cache": int(cached), "share_of_leaf_evaluations": cached / max(1.0, cached + evals)}
                if primary_cache else
                {"in_value": False,
                 "value_with_eval_cache": (second["gfin"] / second["dt"]) if second["gfin"] >= MIN_FINISHED * args.gpus
                 else (second["plies"] / second["dt"]) / ppg,
                 "plies_per_s": second["plies"] / second["dt"], "sims_per_s": second["sims"] / second["dt"],
                 "evals_per_s": second["evals"] / second["dt"], "evaluations_from_cache": int(second["cached"]),
                 "share_of_leaf_evaluations": second["cached"] / max(1.0, second["cached"] + second["evals"]),
                 "ms_per_step": second["dt"] * 1e3 / max(1, args.steps),
                 "note": "same engine configuration and the same K plies timed a second time with engine.eval_cache on (the "
                         "drop-in worker's default): a leaf reached twice in a pass shares one batch row, positions evaluated "
                         "in earlier passes come from a per-game cache; every simulation is played and the games are "
                         "bit-identical (tests/test_eval_cache_gpu.py)"})),
            "half_split": (None if third is None else
                           {"in_value": False,
                            "value_with_half_split": (third["gfin"] / third["dt"]) if third["gfin"] >= MIN_FINISHED * args.gpus
                            else (third["plies"] / third["dt"]) / ppg,
                            "plies_per_s": third["plies"] / third["dt"], "evals_per_s": third["evals"] / third["dt"],
                            "ms_per_step": third["dt"] * 1e3 / max(1, args.steps),
                            "share_of_evaluations_on_the_second_instance": third["rows_tail"] / max(1.0, third["evals"]),
                            "note": "the same K plies timed once more with engine.tail_split = 'halves' (opt-in): every pass as two half "
                                    "batches on two instances over the same weights, side by side, so that one half's attention blocks "
                                    "run beside the other half's power-bound convs; same evaluation-cache setting as `value`; games "
                                    "bit-identical.  Not in `value`, not in `roofline` (per-launch timings of the halves overlap)"}),
            "net_TFLOPs": evals * flops_eval / dt_max / 1e12,
            "passes_per_step": passes / max(1, args.steps), "ms_per_pass": dt_max * 1e3 / passes,
            # the timed region is the friendliest regime: all resident searches are in step, so every pass is a full batch and a
            # ply costs ceil(sims / leaves) passes.  Over a whole generation the searches drift apart (games end at different
            # plies; playout caps differ): `whole_generation` is what tools/calibrate_game_length.py measured end to end
            "passes_per_ply": {"in_step_timed_region": passes / max(1, args.steps),
                               "whole_generation": generation_passes_per_ply()[0],
                               "whole_generation_games_per_s_end_to_end": generation_passes_per_ply()[1],
                               "whole_generation_note": "profiles/game_length.json: 256 games played to completion with engine.eval_cache on"},
            "time_split_ms_per_pass": {"net": ms_net / args.gpus / passes, "tree": ms_tree / args.gpus / passes,
                                       "host": ms_host / args.gpus / passes},
            "roofline": {"bound": "mfma", "kernel": "conv_zs_kernel<*> (3x3 320->320 implicit GEMM, zero padding skipped, MFMA 16x16x32 f16)",
                         "achieved": achieved / 1e12, "peak": PEAK_FP16_DENSE / 1e12, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP16_DENSE, "traffic": traffic,
                         "launches": int(conv_launches), "avg_launch_us": (conv_ms * 1e3 / conv_launches) if conv_launches else None,
                         "launch_mix": mix,
                         "whole_net_frac": (evals * flops_eval / dt_max) / (PEAK_FP16_DENSE * args.gpus)},
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    be.close()
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
