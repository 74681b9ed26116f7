"""Drop-in for the reference's worker entry

    selfplay_worker(proc_id, cfg_dict, ckpt_path, games, q=None, shared_memory_resource=None)

(azchess/selfplay/internal.py:94-95; started as Process(target=selfplay_worker, args=(i, sp_cfg, ckpt, games, q, sm_res))
by orchestrator.py:494 and selfplay/__main__.py:68).  Same config dict, same queue messages
(internal.py:542-556 heartbeat, 665-679 game), same NPZ shards and SQLite rows (internal.py:628-653).

Instead of one game at a time with a Python MCTS, the games of this worker run concurrently on one MI355X inside
libm0engine.so; `shared_memory_resource` (the reference's inference-server handle) is accepted and ignored: leaf
batching happens in the engine.  New keys live under an `engine:` section only:
    engine: {device_index: int, concurrent_games: int, leaves_per_step: int, virtual_loss_active: bool,
             first_game_index: int, compat: {fresh_tree_per_move, tt_merge, raw_legal_priors, root_reinfer}}

The orchestrator hands the SAME cfg_dict to every worker (orchestrator.py:490-496), so what tells workers apart is
proc_id alone: worker i runs on GPU  i % (visible MI355X)  and plays the global game indices [i*games, (i+1)*games) --
the random streams are keyed by (seed, game index), so a game is the same game whichever worker or GPU plays it
(worker_placement below; `engine.device_index` / `engine.first_game_index` override).
"""
from __future__ import annotations

import logging
import os
import random
import time
from typing import Any, Dict, Optional

import numpy as np

from . import encoding
from .backend import M0Backend
from .data_writer import ReplayShardWriter, SelfplayShardWriter
from .engine import SelfplayEngine, SelfplayPool, selfplay_cfg_from_dict
from .weights import random_state_dict

START_W = "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1"
START_B = "rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR b KQkq - 0 1"


def detect_value_from_white(backend: M0Backend) -> bool:
    """selfplay/internal.py:203-238: same position with either side to move; side-to-move nets flip sign."""
    planes, _, _ = encoding.encode_fens([START_W, START_B], want_moves=False)
    _, v = backend.infer_np(planes)
    v1, v2 = float(v[0]), float(v[1])
    return not (abs(v2 + v1) < abs(v2 - v1))


def worker_placement(proc_id: int, games: int, eng_cfg: dict, n_devices: int):
    """(device index, first global game index) of worker `proc_id`: workers spread round-robin over the visible GPUs and
    own disjoint blocks of game indices."""
    dev = eng_cfg.get("device_index")
    dev = int(dev) if dev is not None else (int(proc_id) % max(1, int(n_devices)))
    first = eng_cfg.get("first_game_index")
    first = int(first) if first is not None else int(proc_id) * max(0, int(games))
    return dev, first


def check_unsupported_sections(cfg_dict: dict) -> None:
    """Worker-side features of the reference that this engine does not implement must not be dropped silently."""
    op = cfg_dict.get("openings", {}) or {}
    if op.get("polyglot") and int(op.get("max_plies", 0) or 0) > 0:
        raise NotImplementedError("openings.polyglot (selfplay/internal.py:71-91, 328-333: python-chess polyglot reader) is not "
                                  "implemented by the MI355X engine; use selfplay.opening_random_plies or engine.opening_fens")
    tb = cfg_dict.get("tablebases", {}) or {}
    if tb.get("enabled", False):
        raise NotImplementedError("tablebases.enabled (selfplay/internal.py:250-260, 560-581: Syzygy probing through python-chess) "
                                  "is not implemented by the MI355X engine; set tablebases.enabled: false")


def selfplay_worker(proc_id: int, cfg_dict: dict, ckpt_path: Optional[str], games: int, q=None,
                    shared_memory_resource: Optional[Dict[str, Any]] = None) -> None:
    logger = logging.getLogger(f"selfplay_worker_{proc_id}")
    if int(games) <= 0:                      # `for g in range(games)` (internal.py:326): nothing to play, return at once
        return
    check_unsupported_sections(cfg_dict)
    base_seed = int(cfg_dict.get("seed", 1234))
    random.seed(base_seed + proc_id)
    np.random.seed(base_seed + proc_id)
    eng_cfg = dict(cfg_dict.get("engine", {}) or {})
    from . import _lib
    device_index, first_game_index = worker_placement(proc_id, games, eng_cfg, _lib.device_count())
    model_cfg = dict(cfg_dict["model"])
    def make_backend():
        if ckpt_path and os.path.exists(ckpt_path):
            return M0Backend.from_checkpoint(model_cfg, ckpt_path, device_index)
        return M0Backend.from_state_dict(model_cfg, random_state_dict(model_cfg, seed=base_seed + proc_id), device_index)

    backend = make_backend()
    if ckpt_path and os.path.exists(ckpt_path):
        logger.info("Loaded checkpoint from %s", ckpt_path)
    else:
        logger.info("No checkpoint provided, using untrained model")
    force_vfw = bool((cfg_dict.get("mcts", {}) or {}).get("value_from_white", False))
    value_from_white = force_vfw or detect_value_from_white(backend)
    cfg2 = dict(cfg_dict)
    cfg2["mcts"] = dict(cfg_dict.get("mcts", {}) or {}, value_from_white=value_from_white)
    # SSL targets are generated whenever the model has SSL enabled (selfplay/internal.py:312-318)
    ssl_tasks = list(model_cfg.get("ssl_tasks", [])) if model_cfg.get("self_supervised", False) else []
    concurrent = int(eng_cfg.get("concurrent_games", min(max(1, games), 256)))
    eng_kw = dict(seed=base_seed, leaves_per_step=eng_cfg.get("leaves_per_step"),          # None: mcts.inference_batch_size, as the reference
                  virtual_loss_active=bool(eng_cfg.get("virtual_loss_active", True)), record_games=True,
                  ssl_targets=bool(ssl_tasks), arena_nodes=int(eng_cfg.get("arena_nodes", 0) or 0),
                  # per-game evaluation cache (repeated positions are not evaluated twice; same games): on unless switched off
                  eval_cache=bool(eng_cfg.get("eval_cache", True)))
    # engine.streams > 1: that many independent engines (own network instance and HIP stream each) share the games and
    # step concurrently -- same games, same records per game index, +3..4 % throughput at 2 (engine.SelfplayPool)
    streams = int(eng_cfg.get("streams", 1))
    if streams > 1:
        first = [backend]
        engine = SelfplayPool(lambda: first.pop() if first else make_backend(), cfg2, streams=streams,
                              concurrent_games=min(concurrent, max(1, games)), total_games=games,
                              first_game_index=first_game_index, **eng_kw)
    else:
        scfg = selfplay_cfg_from_dict(cfg2, concurrent_games=min(concurrent, max(1, games)), total_games=games,
                                      first_game_index=first_game_index, **eng_kw)
        engine = SelfplayEngine(backend, scfg)
    # opening book (internal.py:246-248, 39-69): engine.opening_fens, or the PGN of selfplay.book_path read by pgn_book.py
    book = list(eng_cfg.get("opening_fens") or [])
    sp_book = (cfg_dict.get("selfplay", {}) or {}).get("book_path")
    if not book and sp_book:
        from .pgn_book import load_opening_book
        book = load_opening_book(sp_book)
        logger.debug("worker %d: %d positions from opening book %s", proc_id, len(book), sp_book)
    if book:
        for e in (engine.engines if hasattr(engine, "engines") else [engine]):
            e.set_openings(book)
    # engine.replay_shards: emit replay-buffer shards directly (ReplayShardWriter: what the orchestrator's
    # compact_selfplay_to_replay would make of the per-game files) instead of one NPZ per game
    direct_replay = bool(eng_cfg.get("replay_shards", False))
    if direct_replay:
        writer = ReplayShardWriter(base_dir=cfg_dict.get("data_dir", "data"),
                                   max_shards=int(eng_cfg.get("max_shards", 128)), shard_size=int(eng_cfg.get("shard_size", 16384)))
    else:
        writer = SelfplayShardWriter(base_dir=cfg_dict.get("data_dir", "data"))
    last_hb = time.perf_counter()
    done = 0
    overflows = 0
    ssl_dropped = 0
    try:
        while engine.running():
            engine.step(int(eng_cfg.get("steps_per_poll", 8)))
            while True:
                rec = engine.poll()
                if rec is None:
                    break
                T = rec["moves"]
                z = float(rec["result"])
                game_data = {                                       # internal.py:628-646
                    "s": rec["s"].astype(np.float32), "pi": rec["pi"].astype(np.float32), "z": rec["z"].astype(np.float32),
                    "meta_moves": np.array([T], dtype=np.int32), "meta_result": np.array([z], dtype=np.float32),
                    "meta_resigned": np.array([1 if rec["resigned"] else 0], dtype=np.int8),
                    "meta_draw": np.array([1 if z == 0.0 else 0], dtype=np.int8),
                    "meta_avg_policy_entropy": np.array([rec["avg_policy_entropy"]], dtype=np.float32),
                    "meta_avg_sims": np.array([rec["avg_sims"]], dtype=np.float32),
                    "legal_mask": rec["legal_mask"].astype(np.uint8),
                }
                for task in ssl_tasks:                              # internal.py:647-651
                    if "ssl" in rec and task in rec["ssl"]:
                        game_data[f"ssl_{task}"] = rec["ssl"][task].astype(np.float32)
                filepath = None
                if T > 0 and direct_replay:
                    writer.add_game(game_data)
                elif T > 0:
                    filepath = writer.add_selfplay_data(game_data, worker_id=proc_id, game_id=rec["game_index"])
                done += 1
                if q is not None:                                   # internal.py:665-679
                    q.put({"type": "game", "proc": proc_id, "file": filepath, "moves": T, "result": z,
                           "secs": rec["secs"], "resigned": rec["resigned"], "resigner": rec["resigner"],
                           "draw": bool(z == 0.0), "avg_policy_entropy": rec["avg_policy_entropy"],
                           "avg_ms_per_move": rec["secs"] * 1000.0 / max(1, T), "avg_sims": rec["avg_sims"]})
            now = time.perf_counter()
            st = engine.stats()
            if int(st["arena_overflows"]) > overflows:             # truncated searches (mcts.py:435-463 raises there)
                overflows = int(st["arena_overflows"])
                logger.warning("worker %d: %d searches hit the node-arena limit or ended without visits; raise engine.arena_nodes",
                               proc_id, overflows)
            if int(st.get("ssl_dropped", 0)) > ssl_dropped:
                ssl_dropped = int(st["ssl_dropped"])
                logger.warning("worker %d: %d games were written without ssl_* targets (staging buffers could not grow)",
                               proc_id, ssl_dropped)
            if q is not None and now - last_hb >= 2.0:             # internal.py:542-556
                q.put({"type": "heartbeat", "proc": proc_id, "game": done, "moves": int(st["plies"]),
                       "avg_sims": float(st["sims"]) / max(1.0, float(st["plies"])), "resigned": False,
                       "avg_policy_entropy": 0.0})
                last_hb = now
    finally:
        if direct_replay:
            writer.close()                                          # tail shard + pruning
        engine.close()
        backend.close()
