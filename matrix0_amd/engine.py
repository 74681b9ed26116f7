"""ctypes view of the search / self-play part of the C-ABI (include/m0_engine.h)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import numpy as np

from . import _lib

c_double, c_int, c_u64 = C.c_double, C.c_int, C.c_uint64


class SelfplayCfg(C.Structure):
    _fields_ = [
        ("num_simulations", c_int), ("cpuct", c_double), ("cpuct_start", c_double), ("cpuct_end", c_double),
        ("cpuct_plies", c_int), ("use_c_base", c_int), ("cpuct_c_base", c_double), ("cpuct_c_init", c_double),
        ("dirichlet_alpha", c_double), ("dirichlet_frac", c_double), ("dirichlet_plies", c_int),
        ("selection_jitter", c_double), ("fpu_reduction", c_double), ("draw_penalty", c_double), ("virtual_loss", c_double),
        ("legal_softmax", c_int), ("enable_entropy_noise", c_int), ("no_instant_backtrack", c_int), ("value_from_white", c_int),
        ("inference_batch_size", c_int), ("playout_random_frac", c_double),
        ("max_game_len", c_int), ("min_resign_plies", c_int), ("opening_random_plies", c_int),
        ("resign_threshold", c_double), ("resign_window", c_int), ("resign_consecutive_bad", c_int),
        ("resign_min_entropy", c_double), ("resign_value_margin", c_double),
        ("temperature_start", c_double), ("temperature_end", c_double), ("temperature_moves", c_int),
        ("low_visit_threshold", c_int),
        ("draw_enabled", c_int), ("draw_min_plies", c_int), ("draw_window", c_int), ("draw_min_unique", c_int),
        ("draw_halfmove_cap", c_int), ("draw_material_threshold", c_int), ("draw_stalemate", c_int),
        ("concurrent_games", c_int), ("total_games", c_int), ("first_game_index", c_int), ("arena_nodes", c_int),
        ("seed", c_u64), ("virtual_loss_active", c_int), ("ssl_in_forward", c_int), ("ssl_targets", c_int), ("record_games", c_int),
        ("arena_mode", c_int), ("arena_temp", c_double), ("arena_temp_plies", c_int),
        ("fresh_tree_per_move", c_int), ("tt_merge", c_int), ("raw_legal_priors", c_int), ("max_children", c_int),
        ("min_child_prior", c_double), ("root_reinfer", c_int), ("eval_cache", c_int), ("eval_cache_entries", c_int),
        ("tail_split", c_int),
    ]


class SelfplayStats(C.Structure):
    _fields_ = [("steps", c_u64), ("evals", c_u64), ("sims", c_u64), ("plies", c_u64), ("games_finished", c_u64),
                ("games_started", c_u64), ("ms_total", c_double), ("ms_net", c_double), ("ms_tree", c_double),
                ("ms_host", c_double), ("arena_overflows", c_u64), ("ssl_dropped", c_u64), ("evals_cached", c_u64), ("active_games", c_int),
                ("rows_tail", c_u64)]


class GameRecord(C.Structure):
    _fields_ = [("game_index", c_int), ("moves", c_int), ("resigned", c_int), ("resigner", c_int), ("draw", c_int),
                ("total_plies", c_int), ("result", C.c_float), ("avg_policy_entropy", C.c_float), ("avg_sims", C.c_float),
                ("secs", c_double), ("s", C.POINTER(C.c_float)), ("pi", C.POINTER(C.c_float)), ("z", C.POINTER(C.c_float)),
                ("legal_mask", C.POINTER(C.c_uint8)), ("search_values", C.POINTER(C.c_float)),
                ("played", C.POINTER(C.c_uint16)), ("ssl", C.POINTER(C.c_float)), ("owner", C.c_void_p)]


_bound = False


def _bind():
    global _bound
    L = _lib.lib()
    if _bound:
        return L
    L.m0_selfplay_create.restype = C.c_void_p
    L.m0_selfplay_create.argtypes = [C.c_void_p, C.POINTER(SelfplayCfg)]
    L.m0_arena_create.restype = C.c_void_p
    L.m0_arena_create.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(SelfplayCfg)]
    L.m0_arena_create_ext.restype = C.c_void_p
    L.m0_arena_create_ext.argtypes = [C.POINTER(SelfplayCfg)]
    L.m0_arena_ext_select.argtypes = [C.c_void_p, C.POINTER(c_int), C.POINTER(c_int), C.c_void_p, C.c_void_p, c_int]
    L.m0_arena_ext_expand.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, c_int, C.c_void_p, C.c_void_p, c_int]
    L.m0_san_legal_fen.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.POINTER(c_int)]
    L.m0_san_game.argtypes = [C.c_void_p, c_int, C.c_char_p, c_int]
    L.m0_fen_after.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), c_int, C.c_char_p, c_int]
    L.m0_selfplay_destroy.argtypes = [C.c_void_p]
    L.m0_selfplay_destroy.restype = None
    L.m0_selfplay_step.argtypes = [C.c_void_p, c_int]
    L.m0_selfplay_stats_get.argtypes = [C.c_void_p, C.POINTER(SelfplayStats)]
    L.m0_selfplay_poll.argtypes = [C.c_void_p, C.POINTER(GameRecord)]
    L.m0_game_record_free.argtypes = [C.POINTER(GameRecord)]
    L.m0_game_record_free.restype = None
    L.m0_selfplay_running.argtypes = [C.c_void_p]
    L.m0_selfplay_set_openings.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), c_int]
    L.m0_selfplay_ext_select.argtypes = [C.c_void_p, C.POINTER(c_int), C.c_void_p, c_int]
    L.m0_selfplay_ext_expand.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, c_int]
    L.m0_selfplay_last_batch_nhwc.argtypes = [C.c_void_p, C.c_void_p, c_int, C.POINTER(c_int)]
    L.m0_encode_fens_nhwc.argtypes = [c_int, C.POINTER(C.c_char_p), c_int, C.c_void_p]
    L.m0_search_begin.argtypes = [C.c_void_p, c_int, C.c_char_p, c_int, c_int, c_int]
    L.m0_search_select.argtypes = [C.c_void_p, C.POINTER(c_int), C.c_void_p, c_int]
    L.m0_search_expand.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, c_int]
    L.m0_search_result.argtypes = [C.c_void_p, c_int, C.POINTER(c_int)] + [C.c_void_p] * 5 + \
        [C.POINTER(c_double), C.POINTER(c_int), C.POINTER(c_int)]
    L.m0_search_advance.argtypes = [C.c_void_p, c_int, c_int, c_int, c_int]
    L.m0_encode_fens.argtypes = [c_int, C.POINTER(C.c_char_p), c_int] + [C.c_void_p] * 5
    L.m0_move_to_index_fen.argtypes = [c_int, C.c_char_p, C.c_char_p, C.POINTER(C.c_int32)]
    L.m0_decode_move_fen.argtypes = [c_int, C.c_char_p, c_int, C.c_char_p]
    L.m0_ssl_targets_fens.argtypes = [c_int, C.POINTER(C.c_char_p), c_int, C.c_void_p]
    L.m0_sample_move_index.argtypes = [C.c_void_p, c_int, c_double, c_double]
    L.m0_playout_cap.argtypes = [c_int, c_double, c_double]
    L.m0_temperature_for.argtypes = [c_int, c_double, c_double, c_int]
    L.m0_temperature_for.restype = c_double
    L.m0_rules_probe.argtypes = [C.POINTER(SelfplayCfg), C.c_char_p, C.POINTER(C.c_char_p), c_int, C.POINTER(c_int),
                                 C.POINTER(C.c_float)]
    L.m0_arena_choose_move.argtypes = [C.c_void_p, c_int, c_double, c_int, c_int, c_double]
    _bound = True
    return L


def selfplay_cfg_from_dict(cfg: dict, *, concurrent_games: int, total_games: int = 0, first_game_index: int = 0,
                           seed: Optional[int] = None, leaves_per_step: Optional[int] = None,
                           virtual_loss_active: bool = True, ssl_in_forward: bool = False,
                           record_games: bool = True, arena_nodes: int = 0, ssl_targets: bool = False,
                           compat: Optional[dict] = None, eval_cache: Optional[bool] = None,
                           tail_split=None) -> SelfplayCfg:
    """Merge config.yaml's `mcts`, `selfplay` and draw sections exactly as selfplay_worker does
    (azchess/selfplay/internal.py:192-199, 269-304) into the engine's C struct.  MCTSConfig
    defaults are the dataclass defaults of azchess/mcts.py:61-107."""
    m = dict(cfg.get("mcts", {}) or {})
    sp = dict(cfg.get("selfplay", {}) or {})
    draw = dict(cfg.get("draw", {}) or {})
    draw.update(sp.get("draw", {}) or {})
    c = SelfplayCfg()
    c.max_children = int(m.get("max_children", 0) or 0)                 # MCTS._prune_children (mcts.py:806-826)
    c.min_child_prior = float(m.get("min_child_prior", 0.0) or 0.0)
    if c.max_children < 0 or c.max_children > 256 or c.min_child_prior < 0.0:
        raise ValueError("mcts.max_children must be in [0, 256] and mcts.min_child_prior >= 0")
    c.num_simulations = int(sp.get("num_simulations", m.get("num_simulations", 800)))
    c.cpuct = float(sp.get("cpuct", m.get("cpuct", 2.5)))
    cs, ce, cp = m.get("cpuct_start"), m.get("cpuct_end"), int(m.get("cpuct_plies", 0) or 0)
    if cs is not None and ce is not None and cp > 0:
        c.cpuct_start, c.cpuct_end, c.cpuct_plies = float(cs), float(ce), cp
    else:
        c.cpuct_start = c.cpuct_end = c.cpuct
        c.cpuct_plies = 0
    cb, ci = m.get("cpuct_c_base"), m.get("cpuct_c_init")
    c.use_c_base = int(cb is not None and ci is not None)
    c.cpuct_c_base = float(cb or 1.0)
    c.cpuct_c_init = float(ci or 0.0)
    c.dirichlet_alpha = float(sp.get("dirichlet_alpha", m.get("dirichlet_alpha", 0.3)))
    c.dirichlet_frac = float(sp.get("dirichlet_frac", m.get("dirichlet_frac", 0.25)))
    dp = m.get("dirichlet_plies", 16)
    c.dirichlet_plies = -1 if dp is None else int(dp)
    c.selection_jitter = float(sp.get("selection_jitter", m.get("selection_jitter", 0.01)))
    c.fpu_reduction = float(m.get("fpu_reduction", 0.15))
    c.draw_penalty = float(m.get("draw_penalty", -0.1))
    c.virtual_loss = float(m.get("virtual_loss", 1.0))
    c.legal_softmax = int(bool(m.get("legal_softmax", False)))
    c.enable_entropy_noise = int(bool(m.get("enable_entropy_noise", True)))
    c.no_instant_backtrack = int(bool(m.get("no_instant_backtrack", True)))
    c.value_from_white = int(bool(m.get("value_from_white", False)))
    ibs = int(m.get("inference_batch_size", m.get("simulation_batch_size", 96)) or 96)
    c.inference_batch_size = int(leaves_per_step if leaves_per_step else ibs)
    c.playout_random_frac = float(m.get("playout_random_frac", 0.0))
    c.max_game_len = int(sp.get("max_game_len", 200))
    c.min_resign_plies = int(sp.get("min_resign_plies", 24))
    c.opening_random_plies = int(sp.get("opening_random_plies", (cfg.get("openings", {}) or {}).get("random_plies", 0)))
    c.resign_threshold = float(sp.get("resign_threshold", -0.98))
    c.resign_window = int(sp.get("resign_window", 4))
    c.resign_consecutive_bad = int(sp.get("resign_consecutive_bad", 5))
    c.resign_min_entropy = float(sp.get("resign_min_entropy", 0.3))
    c.resign_value_margin = float(sp.get("resign_value_margin", 0.05))
    c.temperature_start = float(sp.get("temperature_start", 1.0))
    c.temperature_end = float(sp.get("temperature_end", 0.1))
    c.temperature_moves = int(sp.get("temperature_moves", 20))
    c.low_visit_threshold = int(sp.get("low_visit_threshold", 0) or 0)
    c.draw_enabled = int(bool(draw.get("enabled", False)))
    c.draw_min_plies = int(draw.get("min_plies", 30))
    c.draw_window = int(draw.get("window", 12))
    c.draw_min_unique = int(draw.get("min_unique", 3))
    c.draw_halfmove_cap = int(draw.get("halfmove_cap", 50))
    c.draw_material_threshold = int(draw.get("material_draw_threshold", 10))
    c.draw_stalemate = int(bool(draw.get("stalemate_draw", True)))
    c.concurrent_games = int(concurrent_games)
    c.total_games = int(total_games)
    c.first_game_index = int(first_game_index)
    c.arena_nodes = int(arena_nodes)
    c.seed = int(cfg.get("seed", 1234) if seed is None else seed)
    c.virtual_loss_active = int(bool(virtual_loss_active))
    c.ssl_in_forward = int(bool(ssl_in_forward))
    c.ssl_targets = int(bool(ssl_targets))
    c.record_games = int(bool(record_games))
    # reference behaviours the engine deviates from by default: `engine.compat` in config.yaml, or the `compat` argument
    cp = dict(((cfg.get("engine", {}) or {}).get("compat", {}) or {}))
    cp.update(compat or {})
    unknown = set(cp) - {"fresh_tree_per_move", "tt_merge", "raw_legal_priors", "root_reinfer"}
    if unknown:
        raise ValueError(f"unknown engine.compat keys: {sorted(unknown)}")
    c.fresh_tree_per_move = int(bool(cp.get("fresh_tree_per_move", False)))
    c.tt_merge = int(bool(cp.get("tt_merge", False)))
    c.raw_legal_priors = int(bool(cp.get("raw_legal_priors", False)))
    c.root_reinfer = int(bool(cp.get("root_reinfer", False)))
    # evaluation cache: `engine.eval_cache` in config.yaml or the keyword; off unless asked for (parity tests count evaluations)
    ecfg = cfg.get("engine", {}) or {}
    c.eval_cache = int(bool(ecfg.get("eval_cache", False) if eval_cache is None else eval_cache))
    c.eval_cache_entries = int(ecfg.get("eval_cache_entries", 0) or 0)
    # `engine.tail_split` (default off): true / 1 = the partial last round of a big pass runs on a second instance
    # over the same weights beside the main forward (+0.4..0.5 % games/s measured); "halves" / 2 = the pass as two halves side by
    # side (+1.3 %; the halves' kernel timings overlap)
    ts = ecfg.get("tail_split", False) if tail_split is None else tail_split
    c.tail_split = 2 if ts in ("halves", "half", 2) and ts is not True else int(bool(ts))
    return c


def move_to_uci(m: int) -> str:
    f, t, p = m & 63, (m >> 6) & 63, (m >> 12) & 7
    s = "abcdefgh"[f & 7] + str((f >> 3) + 1) + "abcdefgh"[t & 7] + str((t >> 3) + 1)
    return s + (" nbrq"[p] if p else "")


class SelfplayEngine:
    """Many concurrent games on one GPU.  `backend` is an M0Backend (its weights and stream are used
    in place); pass None for the split-step search API with an external evaluator."""

    def __init__(self, backend, cfg: SelfplayCfg):
        self._L = _bind()
        self.backend = backend
        self.cfg = cfg
        self._h = self._L.m0_selfplay_create(backend.handle if backend is not None else None, C.byref(cfg))
        if not self._h:
            raise RuntimeError(f"m0_selfplay_create failed: {_lib.last_error()}")

    def step(self, steps: int = 1) -> None:
        _lib.check(self._L.m0_selfplay_step(self._h, int(steps)), "m0_selfplay_step")

    def stats(self) -> Dict[str, float]:
        s = SelfplayStats()
        _lib.check(self._L.m0_selfplay_stats_get(self._h, C.byref(s)), "m0_selfplay_stats_get")
        return {k: getattr(s, k) for k, _ in SelfplayStats._fields_}

    def running(self) -> bool:
        return bool(self._L.m0_selfplay_running(self._h))

    def poll(self) -> Optional[dict]:
        """One finished game as the reference's NPZ dict (selfplay/internal.py:628-646) + queue metadata."""
        r = GameRecord()
        rc = self._L.m0_selfplay_poll(self._h, C.byref(r))
        if rc < 0:
            _lib.check(rc, "m0_selfplay_poll")
        if rc == 0:
            return None
        try:
            T = r.moves
            out = {
                "game_index": r.game_index, "moves": T, "resigned": bool(r.resigned),
                "resigner": {0: None, 1: "W", 2: "B"}[r.resigner], "draw": bool(r.draw), "result": float(r.result),
                "avg_policy_entropy": float(r.avg_policy_entropy), "avg_sims": float(r.avg_sims), "secs": float(r.secs),
                "played": ([move_to_uci(int(x)) for x in np.ctypeslib.as_array(r.played, shape=(r.total_plies,))]
                           if r.total_plies > 0 else []),
                "played_raw": (np.ctypeslib.as_array(r.played, shape=(r.total_plies,)).copy()
                               if r.total_plies > 0 else np.zeros(0, np.uint16)),
            }
            if r.s and T > 0:                                  # arena records carry only the moves and the result
                out["s"] = np.ctypeslib.as_array(r.s, shape=(T, 19, 8, 8)).copy()
                out["pi"] = np.ctypeslib.as_array(r.pi, shape=(T, 4672)).copy()
                out["legal_mask"] = np.ctypeslib.as_array(r.legal_mask, shape=(T, 4672)).copy()
            if T > 0:
                out["z"] = np.ctypeslib.as_array(r.z, shape=(T,)).copy()
                out["search_values"] = np.ctypeslib.as_array(r.search_values, shape=(T,)).copy()
            if r.ssl:
                ssl = np.ctypeslib.as_array(r.ssl, shape=(T, 17, 8, 8)).copy()
                # NPZ field shapes of selfplay/internal.py:475-482: piece [T,13,8,8], the others [T,8,8]
                out["ssl"] = {"piece": ssl[:, :13], "threat": ssl[:, 13], "pin": ssl[:, 14], "fork": ssl[:, 15],
                              "control": ssl[:, 16]}
        finally:
            self._L.m0_game_record_free(C.byref(r))
        return out

    def set_openings(self, fens: List[str]) -> None:
        """Opening book positions (selfplay/internal.py:34-69); call before the first step."""
        arr = (C.c_char_p * max(1, len(fens)))(*[f.encode() for f in fens])
        _lib.check(self._L.m0_selfplay_set_openings(self._h, arr, len(fens)), "m0_selfplay_set_openings")

    def ext_select(self) -> np.ndarray:
        """First half of a self-play step for an external evaluator: the leaf planes f32 [rows,19,8,8]."""
        rows = c_int(0)
        cap = self.cfg.concurrent_games * (self.cfg.inference_batch_size + 1)
        planes = np.zeros((cap, 19, 8, 8), dtype=np.float32)
        _lib.check(self._L.m0_selfplay_ext_select(self._h, C.byref(rows), planes.ctypes.data_as(C.c_void_p), cap), "m0_selfplay_ext_select")
        return planes[: rows.value]

    def ext_expand(self, logits: np.ndarray, values: np.ndarray) -> None:
        lg = np.ascontiguousarray(logits, dtype=np.float32)
        vv = np.ascontiguousarray(values, dtype=np.float32)
        _lib.check(self._L.m0_selfplay_ext_expand(self._h, lg.ctypes.data_as(C.c_void_p), vv.ctypes.data_as(C.c_void_p),
                                                  int(lg.shape[0])), "m0_selfplay_ext_expand")

    def last_batch_nhwc(self) -> np.ndarray:
        """The network batch the last select wrote on the device (what `step()` feeds the network): f16 [rows,64,32],
        row r = position r of the planes that select returned, channels 19..31 zero."""
        cap = self.cfg.concurrent_games * (self.cfg.inference_batch_size + 1) + 4
        out = np.zeros((cap, 64, 32), dtype=np.float16)
        rows = c_int(0)
        _lib.check(self._L.m0_selfplay_last_batch_nhwc(self._h, out.ctypes.data_as(C.c_void_p), cap, C.byref(rows)),
                   "m0_selfplay_last_batch_nhwc")
        return out[: rows.value]

    # ---- split-step search ----
    def search_begin(self, g: int, fen: str, sims: int, dirichlet: bool, game_uid: int) -> None:
        _lib.check(self._L.m0_search_begin(self._h, g, fen.encode(), sims, int(dirichlet), game_uid), "m0_search_begin")

    def search_select(self) -> np.ndarray:
        rows = c_int(0)
        cap = self.cfg.concurrent_games * (self.cfg.inference_batch_size + 1)
        planes = np.zeros((cap, 19, 8, 8), dtype=np.float32)
        _lib.check(self._L.m0_search_select(self._h, C.byref(rows), planes.ctypes.data_as(C.c_void_p), cap), "m0_search_select")
        return planes[: rows.value]

    def search_expand(self, logits: np.ndarray, values: np.ndarray) -> None:
        lg = np.ascontiguousarray(logits, dtype=np.float32)
        vv = np.ascontiguousarray(values, dtype=np.float32)
        _lib.check(self._L.m0_search_expand(self._h, lg.ctypes.data_as(C.c_void_p), vv.ctypes.data_as(C.c_void_p),
                                            int(lg.shape[0])), "m0_search_expand")

    def search_result(self, g: int) -> dict:
        n = c_int(0); rn = c_int(0); fin = c_int(0); rq = c_double(0)
        cn = np.zeros(256, np.int32); mv = np.zeros(256, np.uint16); idx = np.zeros(256, np.int32)
        pr = np.zeros(256, np.float64); q = np.zeros(256, np.float64)
        _lib.check(self._L.m0_search_result(self._h, g, C.byref(n), cn.ctypes.data_as(C.c_void_p), mv.ctypes.data_as(C.c_void_p),
                                            idx.ctypes.data_as(C.c_void_p), pr.ctypes.data_as(C.c_void_p),
                                            q.ctypes.data_as(C.c_void_p), C.byref(rq), C.byref(rn), C.byref(fin)),
                   "m0_search_result")
        k = n.value
        return {"finished": bool(fin.value), "n": cn[:k].copy(), "moves": [move_to_uci(int(x)) for x in mv[:k]],
                "idx": idx[:k].copy(), "prior": pr[:k].copy(), "q": q[:k].copy(), "root_q": rq.value, "root_n": rn.value}

    def search_advance(self, g: int, slot: int, sims: int, dirichlet: bool) -> None:
        _lib.check(self._L.m0_search_advance(self._h, g, slot, sims, int(dirichlet)), "m0_search_advance")

    def close(self):
        if getattr(self, "_h", None):
            self._L.m0_selfplay_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SelfplayPool:
    """`streams` independent SelfplayEngines on ONE GPU behind the SelfplayEngine interface: every engine has its own
    network instance (weights, workspace, HIP stream) and an equal share of the concurrent games, and `step()` runs them
    concurrently, one host thread each (the C calls release the GIL).  Games never interact, so nothing else changes --
    a game's record depends only on (seed, game index), whichever engine plays it -- but the tree kernels, launch
    boundaries and epilogue HBM bursts of one share now overlap the network kernels of the other: +3..4 % evaluations/s
    at 2 x 128 games against 1 x 256 (bench.py --streams 2; 4 x 64 gives nothing more).  Costs one more copy of the
    weights in HBM.  Per-launch kernel timings taken in this mode include the other stream's kernels."""

    def __init__(self, backend_factory, cfg_dict: dict, *, streams: int, concurrent_games: int, total_games: int = 0,
                 first_game_index: int = 0, **cfg_kw):
        import threading
        self._threading = threading
        streams = max(1, min(int(streams), int(concurrent_games)))
        self.backends, self.engines = [], []
        per = [concurrent_games // streams + (1 if i < concurrent_games % streams else 0) for i in range(streams)]
        tot = [0] * streams if total_games <= 0 else \
              [total_games // streams + (1 if i < total_games % streams else 0) for i in range(streams)]
        first = first_game_index
        try:
            for i in range(streams):
                be = backend_factory()
                self.backends.append(be)
                cfg = selfplay_cfg_from_dict(cfg_dict, concurrent_games=(min(per[i], tot[i]) if total_games > 0 else per[i]) or 1,
                                             total_games=tot[i], first_game_index=first, **cfg_kw)
                self.engines.append(SelfplayEngine(be, cfg))
                # unbounded mode: each engine restarts games forever, so it gets a disjoint block of indices
                first += tot[i] if total_games > 0 else (1 << 20)
        except Exception:
            self.close()
            raise
        self._rr = 0

    def step(self, steps: int = 1) -> None:
        live = [e for e in self.engines if e.running()]
        if len(live) <= 1:
            for e in live:
                e.step(steps)
            return
        errs = []

        def run(e):
            try:
                e.step(steps)
            except Exception as ex:          # re-raised in the caller's thread
                errs.append(ex)

        th = [self._threading.Thread(target=run, args=(e,)) for e in live]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            raise errs[0]

    def running(self) -> bool:
        return any(e.running() for e in self.engines)

    def poll(self) -> Optional[dict]:
        for k in range(len(self.engines)):
            e = self.engines[(self._rr + k) % len(self.engines)]
            rec = e.poll()
            if rec is not None:
                self._rr = (self._rr + k + 1) % len(self.engines)
                return rec
        return None

    def stats(self) -> Dict[str, float]:
        """Counters summed over the engines; the ms_* clocks are per-engine host clocks of concurrent work: averaged."""
        out: Dict[str, float] = {}
        sts = [e.stats() for e in self.engines]
        for k in sts[0]:
            v = sum(s[k] for s in sts)
            out[k] = v / len(sts) if k.startswith("ms_") or k == "steps" else v
        return out

    def close(self):
        for e in getattr(self, "engines", []):
            e.close()
        for b in getattr(self, "backends", []):
            try:
                b.close()
            except Exception:
                pass
        self.engines, self.backends = [], []


# ---- host decision functions (no GPU needed) ----
def sample_move_index(visits, temperature: float, u: float) -> int:
    L = _bind()
    v = np.ascontiguousarray(visits, dtype=np.int32)
    return int(L.m0_sample_move_index(v.ctypes.data_as(C.c_void_p), int(v.shape[0]), float(temperature), float(u)))


def playout_cap(sims: int, frac: float, u: float) -> int:
    return int(_bind().m0_playout_cap(int(sims), float(frac), float(u)))


def temperature_for(fullmove: int, t_start: float, t_end: float, t_moves: int) -> float:
    return float(_bind().m0_temperature_for(int(fullmove), float(t_start), float(t_end), int(t_moves)))


RULE_FLAGS = ["game_over", "game_over_claim", "adjudicate_draw", "checkmate", "stalemate", "insufficient",
              "can_claim_fifty", "repetition3", "can_claim_threefold", "fivefold", "seventyfive"]


def rules_probe(cfg: SelfplayCfg, fen: str, ucis: List[str]) -> dict:
    L = _bind()
    arr = (C.c_char_p * max(1, len(ucis)))(*[u.encode() for u in ucis])
    flags = c_int(0); res = C.c_float(0)
    _lib.check(L.m0_rules_probe(C.byref(cfg), fen.encode(), arr, len(ucis), C.byref(flags), C.byref(res)), "m0_rules_probe")
    out = {name: bool(flags.value >> i & 1) for i, name in enumerate(RULE_FLAGS)}
    out["result"] = float(res.value)
    return out


def encode_fens_nhwc(fens, device_index: int = 0) -> np.ndarray:
    """encode_board (encoding.py:11-46) as the search's select kernel writes it for the network: f16 [n,64,32]."""
    L = _bind()
    n = len(fens)
    arr = (C.c_char_p * n)(*[f.encode() for f in fens])
    out = np.empty((n, 64, 32), np.float16)
    _lib.check(L.m0_encode_fens_nhwc(int(device_index), arr, n, out.ctypes.data_as(C.c_void_p)), "m0_encode_fens_nhwc")
    return out


def planes_to_nhwc(planes: np.ndarray) -> np.ndarray:
    """f32 [n,19,8,8] -> the network's input layout f16 [n,64,32] (zero-padded channels)."""
    p = np.asarray(planes, np.float32)
    out = np.zeros((p.shape[0], 64, 32), np.float16)
    out[:, :, :19] = p.reshape(p.shape[0], 19, 64).transpose(0, 2, 1).astype(np.float16)
    return out


def ssl_targets_fens(fens, device_index: int = 0) -> dict:
    """create_enhanced_ssl_targets (ssl_algorithms.py:519-543) on the device for a list of FENs."""
    L = _bind()
    n = len(fens)
    arr = (C.c_char_p * n)(*[f.encode() for f in fens])
    out = np.empty((n, 17, 8, 8), np.float32)
    _lib.check(L.m0_ssl_targets_fens(int(device_index), arr, n, out.ctypes.data_as(C.c_void_p)), "m0_ssl_targets_fens")
    return {"piece": out[:, :13], "threat": out[:, 13], "pin": out[:, 14], "fork": out[:, 15], "control": out[:, 16]}


class ArenaEngine(SelfplayEngine):
    """Evaluation match engine (m0_arena_create): game i has `backend_a` as White when i is even; step / poll / stats as
    SelfplayEngine.  Records carry `played`, `result` (White's point of view) and `moves`."""

    def __init__(self, backend_a, backend_b, cfg: SelfplayCfg):
        self._L = _bind()
        self.backend = backend_a
        self.backend_b = backend_b
        self.cfg = cfg
        self._h = self._L.m0_arena_create(backend_a.handle, backend_b.handle, C.byref(cfg))
        if not self._h:
            raise RuntimeError(f"m0_arena_create failed: {_lib.last_error()}")


class ArenaExtEngine(SelfplayEngine):
    """Match engine without networks (m0_arena_create_ext): two external evaluators behind the infer_np seam."""

    def __init__(self, cfg: SelfplayCfg):
        self._L = _bind()
        self.backend = None
        self.cfg = cfg
        self._h = self._L.m0_arena_create_ext(C.byref(cfg))
        if not self._h:
            raise RuntimeError(f"m0_arena_create_ext failed: {_lib.last_error()}")

    def arena_ext_select(self):
        ra, rb = c_int(0), c_int(0)
        cap = self.cfg.concurrent_games * (self.cfg.inference_batch_size + 1)
        pa = np.zeros((cap, 19, 8, 8), dtype=np.float32)
        pb = np.zeros((cap, 19, 8, 8), dtype=np.float32)
        _lib.check(self._L.m0_arena_ext_select(self._h, C.byref(ra), C.byref(rb), pa.ctypes.data_as(C.c_void_p),
                                               pb.ctypes.data_as(C.c_void_p), cap), "m0_arena_ext_select")
        return pa[: ra.value], pb[: rb.value]

    def arena_ext_expand(self, lg_a, v_a, lg_b, v_b) -> None:
        la = np.ascontiguousarray(lg_a, dtype=np.float32); va = np.ascontiguousarray(v_a, dtype=np.float32)
        lb = np.ascontiguousarray(lg_b, dtype=np.float32); vb = np.ascontiguousarray(v_b, dtype=np.float32)
        _lib.check(self._L.m0_arena_ext_expand(self._h, la.ctypes.data_as(C.c_void_p), va.ctypes.data_as(C.c_void_p), int(la.shape[0]),
                                               lb.ctypes.data_as(C.c_void_p), vb.ctypes.data_as(C.c_void_p), int(lb.shape[0])),
                   "m0_arena_ext_expand")


def arena_choose_move(visits, temp: float, ply: int, temp_plies: int, u: float) -> int:
    """arena.py:73-106 (host_rules.h::arena_choose_move)."""
    L = _bind()
    v = np.ascontiguousarray(visits, dtype=np.int32)
    return int(L.m0_arena_choose_move(v.ctypes.data_as(C.c_void_p), int(len(v)), float(temp), int(ply), int(temp_plies), float(u)))


def san_legal(fen: str):
    """[(uci, san)] of the legal moves of `fen` in legal_moves order (python-chess Board.san semantics)."""
    L = _bind()
    mv = np.zeros(256, np.uint16)
    san = C.create_string_buffer(256 * 8)
    n = c_int(0)
    _lib.check(L.m0_san_legal_fen(fen.encode(), mv.ctypes.data_as(C.c_void_p), san, C.byref(n)), "m0_san_legal_fen")
    raw = san.raw
    return [(move_to_uci(int(mv[i])), raw[8 * i: 8 * i + 8].split(b"\0", 1)[0].decode()) for i in range(n.value)]


def fen_after(fen: str, ucis) -> str:
    """Board.fen() after pushing the legal moves `ucis` on `fen` (python-chess semantics); ValueError for an illegal move."""
    L = _bind()
    ucis = list(ucis)
    arr = (C.c_char_p * max(1, len(ucis)))(*[u.encode() for u in ucis])
    buf = C.create_string_buffer(128)
    rc = L.m0_fen_after(fen.encode(), arr, len(ucis), buf, len(buf))
    if rc == -1:
        raise ValueError(_lib.last_error())
    _lib.check(rc, "m0_fen_after")
    return buf.value.decode()


def san_game(moves_raw) -> str:
    """Movetext '1. e4 e5 2. Nf3 ...' of a game from the start position (moves as in a record's `played_raw`)."""
    L = _bind()
    mv = np.ascontiguousarray(moves_raw, dtype=np.uint16)
    buf = C.create_string_buffer(16 * (len(mv) + 4))
    rc = L.m0_san_game(mv.ctypes.data_as(C.c_void_p), int(len(mv)), buf, len(buf))
    if rc < 0:
        _lib.check(rc, "m0_san_game")
    return buf.value.decode().strip()
