"""ORACLE (test infrastructure, not product): CPU restatement of the reference search and
self-play decision logic with INJECTED randomness.

Follows (file:line in /root/reference):
  azchess/mcts.py:120-225    Node, Node._expand (legal-only softmax, entropy noise, renormalise)
  azchess/mcts.py:318-512    MCTS.run (root fetch/expand, Dirichlet, playout cap, result triple)
  azchess/mcts.py:514-769    _run_simulations_parallel_batched / _collect_leaf_position
  azchess/mcts.py:851-925    _select (PUCT, FPU reduction, instant-backtrack, virtual loss, jitter)
  azchess/mcts.py:927-953    _cpuct_at, _backpropagate
  azchess/mcts.py:955-992    _add_dirichlet
  azchess/mcts.py:1223-1229  _terminal_value
  azchess/mcts.py:1231-1346  TT (optional here: use_tt)
  azchess/selfplay/internal.py:386-394 temperature, 507-536 resign, 587-599/738-750 result,
                               690-735 sample_move_from_counts
  azchess/draw.py:8-84       should_adjudicate_draw

Randomness: the reference draws from Python's `random` (jitter, playout cap, opening plies) and
numpy's global RNG (entropy noise, Dirichlet, move sampling).  Neither Mersenne-Twister stream
can be shared with a GPU, so every draw here comes from counter-based streams
(`Stream`: splitmix64 of seed + counter) that the engine implements bit-identically; parity is
per function on identical injected draws (SURVEY §7 hard part 5).

Deviation switches (reference default in brackets):
  use_tt [True]          the product is tree-only (no transposition merging), tests use False
  virtual_loss_active [False: dead code in the reference, SURVEY B-1] the product applies it as written
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch

from . import chess_py as ch

MASK64 = (1 << 64) - 1
GOLDEN = 0x9E3779B97F4A7C15


def mix64(x: int) -> int:
    x &= MASK64
    x ^= x >> 30; x = (x * 0xBF58476D1CE4E5B9) & MASK64
    x ^= x >> 27; x = (x * 0x94D049BB133111EB) & MASK64
    x ^= x >> 31
    return x


class Stream:
    """Counter-based uniform stream: u_k = (mix64(seed + (k+1)*GOLDEN) >> 11) * 2^-53."""

    def __init__(self, seed: int, ctr: int = 0):
        self.seed = seed & MASK64
        self.ctr = ctr

    def at(self, k: int) -> float:
        return (mix64(self.seed + ((k + 1) * GOLDEN)) >> 11) * (1.0 / 9007199254740992.0)

    def next(self) -> float:
        u = self.at(self.ctr)
        self.ctr += 1
        return u

    def normal(self) -> float:
        """Box-Muller, two uniforms per normal (cosine branch)."""
        u1 = self.next()
        u2 = self.next()
        if u1 < 1e-300:
            u1 = 1e-300
        return math.sqrt(-2.0 * math.log(u1)) * math.cos(2.0 * math.pi * u2)

    def gamma(self, a: float) -> float:
        """Marsaglia-Tsang; a < 1 via gamma(a+1) * U^(1/a)."""
        boost = 1.0
        if a < 1.0:
            u = self.next()
            boost = u ** (1.0 / a)
            a = a + 1.0
        d = a - 1.0 / 3.0
        c = 1.0 / math.sqrt(9.0 * d)
        while True:
            x = self.normal()
            v = 1.0 + c * x
            if v <= 0.0:
                continue
            v = v * v * v
            u = self.next()
            if u < 1e-300:
                u = 1e-300
            if math.log(u) < 0.5 * x * x + d - d * v + d * math.log(v):
                return d * v * boost


def derive_seed(base: int, game: int, purpose: int) -> int:
    return mix64(mix64(base + GOLDEN * (game + 1)) ^ (purpose * 0xD6E8FEB86659FD93))


PURPOSE_JITTER, PURPOSE_NOISE, PURPOSE_DIRICHLET, PURPOSE_GAME = 1, 2, 3, 4


@dataclass
class MCTSConfig:
    """mcts.py:61-107 (fields the hot path reads)."""
    num_simulations: int = 800
    cpuct: float = 2.5
    dirichlet_alpha: float = 0.3
    dirichlet_frac: float = 0.25
    dirichlet_plies: int = 16
    selection_jitter: float = 0.01
    fpu_reduction: float = 0.15
    draw_penalty: float = -0.1
    virtual_loss: float = 1.0
    cpuct_start: Optional[float] = None
    cpuct_end: Optional[float] = None
    cpuct_plies: int = 0
    cpuct_c_base: Optional[float] = None
    cpuct_c_init: Optional[float] = None
    value_from_white: bool = False
    legal_softmax: bool = False
    no_instant_backtrack: bool = True
    inference_batch_size: int = 96
    playout_random_frac: float = 0.0
    enable_entropy_noise: bool = True
    max_children: int = 0
    min_child_prior: float = 0.0
    # which of the reference's two expansion branches non-root leaves take: False = the inference_backend branch
    # (Node._expand, legal softmax; mcts.py:654-664), True = the in-process-model branch when legal_softmax is set
    # (Node._expand_with_legal_priors: legal logits / their sum, mcts.py:697-703)
    raw_legal_priors: bool = False
    # build switches (see module docstring)
    use_tt: bool = True
    virtual_loss_active: bool = False
    numerics: str = "reference"   # "engine": see legal_priors

    @classmethod
    def from_dict(cls, d):
        known = set(cls.__dataclass_fields__.keys())
        return cls(**{k: v for k, v in d.items() if k in known})


class Node:
    __slots__ = ("parent", "prior", "n", "w", "q", "children", "move", "expanded", "move_idx")

    def __init__(self, prior=0.0, move=None, parent=None):
        self.parent = parent
        self.prior = prior
        self.n = 0
        self.w = 0.0
        self.q = 0.0
        self.children: Dict[ch.Move, "Node"] = {}
        self.move = move
        self.expanded = False
        self.move_idx = None


def legal_priors(logits: np.ndarray, idxs: List[int], legal_only: bool, allow_noise: bool,
                 noise: Optional[Stream], numerics: str = "reference") -> np.ndarray:
    """The arithmetic of Node._expand, mcts.py:144-212 -> float32 priors over the legal moves.

    numerics="reference": torch float32 softmax, numpy float32 entropy and sums (what the reference runs).
    numerics="engine":    (logit-max) in float32, exp/sum/divide in float64, rounded to float32; entropy and
                          the final legal sum in float64.  Within one float32 ulp of "reference"; it is what the
                          HIP expand kernel computes, reproducible bit-for-bit, so whole-search trajectories
                          can be compared exactly.  tests pin "engine" against "reference" at 1e-6."""
    n = len(idxs)
    logits = logits.astype(np.float32, copy=False)
    if np.any(np.isnan(logits)) or np.any(np.isinf(logits)):
        return np.full(n, 1.0 / n, dtype=np.float32)
    sel = np.ascontiguousarray(logits[idxs] if legal_only else logits)
    if numerics == "reference":
        dist = torch.softmax(torch.from_numpy(sel), dim=-1).numpy()
        policy_entropy = -np.sum(dist * np.log(dist + 1e-8))
    else:
        d32 = (sel - sel.max()).astype(np.float32)
        e = np.exp(d32.astype(np.float64))
        dist = (e / e.sum()).astype(np.float32)
        d64 = dist.astype(np.float64)
        policy_entropy = -np.sum(d64 * np.log(d64 + 1e-8))
    max_entropy = np.log(max(1, n))
    entropy_ratio = policy_entropy / max(1e-9, max_entropy)
    if allow_noise and entropy_ratio > 0.9:
        z = np.array([noise.normal() for _ in range(dist.shape[0])], dtype=np.float64) * 0.1
        dist = dist + z
        dist = np.maximum(dist, 1e-8)
        dist = dist / dist.sum()
    pri = []
    for i in range(n):
        p = float(dist[i]) if legal_only else float(dist[idxs[i]])
        if np.isnan(p) or np.isinf(p) or p < 0:
            p = 0.0
        pri.append(p)
    lp = np.asarray(pri, dtype=np.float32)
    total = lp.sum() if numerics == "reference" else np.float32(lp.astype(np.float64).sum())
    if total > 0 and not np.isnan(total) and not np.isinf(total):
        lp = lp / total
    else:
        lp = np.full(n, 1.0 / n, dtype=np.float32)
    return lp


class MCTS:
    def __init__(self, cfg: MCTSConfig, infer: Callable[[np.ndarray], Tuple[np.ndarray, np.ndarray]],
                 seed: int = 1234, game: int = 0):
        self.cfg = cfg
        self.infer_np = infer                      # the infer_np seam: [B,19,8,8] -> ([B,4672], [B])
        self.jitter = Stream(derive_seed(seed, game, PURPOSE_JITTER))
        self.noise = Stream(derive_seed(seed, game, PURPOSE_NOISE))
        self.dirichlet = Stream(derive_seed(seed, game, PURPOSE_DIRICHLET))
        self.tt: Dict[bytes, Node] = {}
        self.nn_cache: "OrderedDict[bytes, tuple]" = OrderedDict()
        self.evals = 0
        self._last_sims_run = 0
        self._last_root = None

    # ---- pieces ----
    def cpuct_at(self, ply: int) -> float:
        c = self.cfg
        if c.cpuct_c_base is not None and c.cpuct_c_init is not None:
            N = max(1.0, float(ply + 1))
            return float(c.cpuct_c_init) + math.log((N + float(c.cpuct_c_base)) / float(c.cpuct_c_base))
        if c.cpuct_start is None or c.cpuct_end is None or int(c.cpuct_plies) <= 0:
            return float(c.cpuct)
        t = min(max(ply, 0), int(c.cpuct_plies)) / float(int(c.cpuct_plies))
        return float(c.cpuct_start) + (float(c.cpuct_end) - float(c.cpuct_start)) * t

    def terminal_value(self, board: ch.Board) -> float:
        if board.is_checkmate():
            return -1.0
        if board.is_stalemate() or board.is_insufficient_material() or board.is_seventyfive_moves() or \
                board.is_fivefold_repetition():
            return float(self.cfg.draw_penalty)
        return 0.0

    def prune_children(self, node: Node) -> None:
        """MCTS._prune_children, mcts.py:806-826 (stable sort: ties keep move order; no renormalisation)."""
        if not node.children:
            return
        items = list(node.children.items())
        if self.cfg.min_child_prior > 0.0:
            items = [(m, c) for (m, c) in items if float(c.prior) >= float(self.cfg.min_child_prior)]
        if self.cfg.max_children and self.cfg.max_children > 0 and len(items) > self.cfg.max_children:
            items.sort(key=lambda mc: float(mc[1].prior), reverse=True)
            items = items[: int(self.cfg.max_children)]
        node.children = {m: c for (m, c) in items}

    def expand(self, node: Node, board: ch.Board, logits: np.ndarray, is_root: bool = False) -> None:
        if node.expanded:
            return
        moves, idxs = ch.legal_moves_with_indices(board)
        if not moves:
            return
        if self.cfg.raw_legal_priors and self.cfg.legal_softmax and not is_root:
            # Node._expand_with_legal_priors, mcts.py:227-256
            pri = np.asarray(logits, dtype=np.float32)[idxs].astype(np.float32, copy=False)
            total = float(pri.sum())
            if not np.isfinite(total) or total <= 0:
                lp = np.full(len(moves), 1.0 / len(moves), dtype=np.float32)
            else:
                lp = pri / total
        else:
            lp = legal_priors(logits, idxs, self.cfg.legal_softmax, bool(self.cfg.enable_entropy_noise), self.noise,
                              self.cfg.numerics)
        for m, idx, p in zip(moves, idxs, lp):
            c = Node(prior=float(p), move=m, parent=node)
            c.move_idx = int(idx)
            if node.parent and node.parent.q != 0.0:
                c.q = -node.parent.q
            node.children[m] = c
        node.expanded = True
        self.prune_children(node)

    def register_children(self, node: Node, board: ch.Board) -> None:
        if not self.cfg.use_tt:
            return
        for m, c in node.children.items():
            b2 = board.copy()
            b2.push(m)
            self.tt[b2._transposition_key()] = c

    def select(self, board: ch.Board, root: Node, inflight: Optional[Dict[Node, int]]):
        cfg = self.cfg
        node = root
        path = [root]
        while node.expanded:
            if not node.children:
                break
            parent_visits = max(1, node.n)
            best_score, best = -1e9, None
            depth = max(0, len(path) - 1)
            eff = self.cpuct_at(depth)
            for child in node.children.values():
                q = (float(node.q) - float(cfg.fpu_reduction)) if child.n == 0 else child.q
                u = eff * child.prior * (math.sqrt(parent_visits) / (1.0 + child.n))
                score = q + u
                if cfg.no_instant_backtrack and len(path) >= 2 and child.move is not None and path[-1].move is not None:
                    prev = path[-1].move
                    if child.move.from_square == prev.to_square and child.move.to_square == prev.from_square:
                        score -= 0.01
                if inflight is not None and cfg.virtual_loss > 0.0:
                    score -= float(inflight.get(child, 0)) * float(cfg.virtual_loss)
                jit = cfg.selection_jitter if cfg.selection_jitter > 0 else 0.001
                score += (self.jitter.next() - 0.5) * jit
                if score > best_score:
                    best_score, best = score, child
            board.push(best.move)
            nxt = self.tt.get(board._transposition_key()) if cfg.use_tt else None
            node = nxt or best
            path.append(node)
            if inflight is not None:
                inflight[best] = inflight.get(best, 0) + 1
        return node, path, board

    @staticmethod
    def backpropagate(path: List[Node], value: float) -> None:
        v = max(-1.0, min(1.0, float(value)))
        for node in reversed(path):
            node.n += 1
            node.w += v
            node.q = node.w / node.n
            v = -v

    def add_dirichlet(self, root: Node) -> None:
        cfg = self.cfg
        if not root.children or cfg.dirichlet_frac <= 0:
            return
        k = len(root.children)
        g = [self.dirichlet.gamma(cfg.dirichlet_alpha) for _ in range(k)]
        s = sum(g)
        frac = cfg.dirichlet_frac
        for i, child in enumerate(root.children.values()):
            new = child.prior * (1 - frac) + (g[i] / s) * frac
            child.prior = max(1e-8, min(1.0 - 1e-8, new))

    def _infer_one(self, board: ch.Board):
        p, v = self.infer_np(ch.encode_board(board)[None])
        self.evals += 1
        v = float(np.clip(float(v[0]), -1.0, 1.0))
        if self.cfg.value_from_white and not board.turn:
            v = -v
        return p[0], v

    # ---- batched simulations (mcts.py:514-769) ----
    def run_batched(self, board: ch.Board, root: Node, sims: int) -> None:
        L = int(self.cfg.inference_batch_size) or 96
        done = 0
        while done < sims:
            batch_n = min(L, sims - done)
            inflight = {} if self.cfg.virtual_loss_active else None
            samples = []
            for _ in range(batch_n):
                node, path, leaf = self.select(board.copy(), root, inflight)
                if leaf.is_game_over():
                    self.backpropagate(path, self.terminal_value(leaf))
                else:
                    samples.append((node, list(path), leaf))
            if samples:
                x = np.stack([ch.encode_board(b) for (_, _, b) in samples], axis=0)
                pol, val = self.infer_np(x)
                self.evals += len(samples)
                for (node, path, leaf), p, v in zip(samples, pol, val):
                    if not node.expanded:
                        self.expand(node, leaf, p)
                        self.register_children(node, leaf)
                    self.backpropagate(path, float(np.clip(v, -1.0, 1.0)))
            done += batch_n

    # ---- MCTS.run (mcts.py:318-512) ----
    def run(self, board: ch.Board, num_simulations: Optional[int] = None, ply: Optional[int] = None,
            sims_override: Optional[int] = None):
        cfg = self.cfg
        if board.is_game_over():
            return {}, np.zeros(4672, dtype=np.float32), self.terminal_value(board)
        key = board._transposition_key()
        root = self.tt.get(key) if cfg.use_tt else self._last_child_root(board)
        v = 0.0
        if root is None:
            root = Node()
            logits, v = self._infer_one(board)
            self.expand(root, board, logits, is_root=True)
            if cfg.use_tt:
                self.tt[key] = root
        else:
            # mcts.py:359-371: a root found in the table is re-evaluated unless (logits, v) of this position sit in
            # the 10 000-entry nn_cache (mcts.py:44-59), which is written here and nowhere else
            cached = self.nn_cache.get(key)
            if cached is None:
                logits, v = self._infer_one(board)
                self.nn_cache[key] = (logits, v)
                self.nn_cache.move_to_end(key)
                if len(self.nn_cache) > 10000:
                    self.nn_cache.popitem(last=False)
            else:
                self.nn_cache.move_to_end(key)
                _, v = cached
        if cfg.dirichlet_plies is None or ply is None or ply < int(cfg.dirichlet_plies):
            self.add_dirichlet(root)
        sims = num_simulations if num_simulations is not None else cfg.num_simulations
        if sims_override is not None:
            sims = sims_override            # playout-cap draw injected by the caller
        if not root.expanded:
            logits, v = self._infer_one(board)
            self.expand(root, board, logits, is_root=True)
            self.register_children(root, board)
            root.q = float(np.clip(v, -1.0, 1.0))
            if cfg.dirichlet_plies is None or ply is None or ply < int(cfg.dirichlet_plies):
                pass  # reference applied Dirichlet before this expansion (no children then): no-op there
        self.run_batched(board, root, sims)
        visit_counts = {m: c.n for m, c in root.children.items()}
        pi = np.zeros(4672, dtype=np.float32)
        total = sum(c.n for c in root.children.values())
        if total > 0:
            for m, c in root.children.items():
                pi[c.move_idx] = c.n / total
        self._last_sims_run = sims
        self._last_root = root
        root_q = float(root.q) if root.n > 0 else float(v)
        return visit_counts, pi, root_q

    def _last_child_root(self, board):
        """Tree-only mode: reuse the played child's subtree (what the TT gives the reference for the
        position actually reached); a fresh Node otherwise."""
        lr = self._last_root
        if lr is None or not hasattr(self, "_last_move") or self._last_move is None:
            return None
        c = lr.children.get(self._last_move)
        self._last_move = None
        return c

    def note_move_played(self, move: ch.Move):
        self._last_move = move


# ---- self-play decision functions (host logic of the engine mirrors these) ----
def playout_cap(sims: int, frac: float, u: float) -> int:
    """mcts.py:378-387 with random.randint(low, high) replaced by low + floor(u*(high-low+1))."""
    if frac > 0.0 and sims > 0:
        low = int(max(1, sims * (1.0 - frac)))
        high = int(max(low, sims * (1.0 + frac)))
        return low + min(high - low, int(u * (high - low + 1)))
    return sims


def temperature_for(fullmove_number: int, t_start: float, t_end: float, t_moves: int) -> float:
    """selfplay/internal.py:386-394"""
    if t_moves <= 0:
        return t_end
    t = min(max(fullmove_number, 0), t_moves) / float(max(1, t_moves))
    return t_start + (t_end - t_start) * t


def sample_move_index(visits: List[int], temperature: float, u: float) -> int:
    """selfplay/internal.py:690-735 with np.random.choice(p=...) replaced by inverse-CDF on u
    (numpy: cdf = cumsum(p) in float64, cdf /= cdf[-1], searchsorted(u, side='right'))."""
    v = np.array(visits, dtype=np.float32)
    if np.all(v == 0):
        return min(len(visits) - 1, int(u * len(visits)))
    if temperature < 1e-3:
        return int(np.argmax(v))
    with np.errstate(over="ignore", invalid="ignore"):
        d = v ** (1.0 / temperature)
        s = d.sum()
        if s <= 0 or np.isnan(s):
            return min(len(visits) - 1, int(u * len(visits)))
        d = d / s
    if np.any(np.isnan(d)):                        # internal.py:727-732 (overflowed powers: inf / inf)
        return min(len(visits) - 1, int(u * len(visits)))
    cdf = np.cumsum(d.astype(np.float64))
    cdf /= cdf[-1]
    return int(min(len(visits) - 1, np.searchsorted(cdf, u, side="right")))


def sample_move_draws(visits: List[int], temperature: float) -> bool:
    """Whether sample_move_from_counts consumes a random draw: every branch but the deterministic arg-max
    (internal.py:706-708) calls np.random.choice."""
    return all(x == 0 for x in visits) or not (temperature < 1e-3)


def policy_entropy(pi: np.ndarray) -> float:
    """selfplay/internal.py:430-431"""
    p = np.clip(pi.astype(np.float64, copy=False), 1e-12, 1.0)
    return float(-np.sum(p * np.log(p)))


@dataclass
class ResignState:
    consec_bad: int = 0
    recent_values: List[float] = field(default_factory=list)
    recent_entropies: List[float] = field(default_factory=list)


def resign_update(st: ResignState, v: float, n_states: int, sp_cfg: dict) -> bool:
    """selfplay/internal.py:507-536 (entropy window is appended by the caller, :434-436)."""
    thr = float(sp_cfg.get("resign_threshold", -0.98))
    min_plies = int(sp_cfg.get("min_resign_plies", 24))
    window_k = int(sp_cfg.get("resign_window", 4))
    min_entropy = float(sp_cfg.get("resign_min_entropy", 0.3))
    margin = float(sp_cfg.get("resign_value_margin", 0.05))
    if not (thr > -1.0 and n_states >= min_plies):
        return False
    st.recent_values.append(float(v))
    if len(st.recent_values) > window_k:
        st.recent_values.pop(0)
    if v < thr:
        st.consec_bad += 1
    else:
        st.consec_bad = 0
    seq_bad = int(sp_cfg.get("resign_consecutive_bad", 5))
    stable_bad = False
    if len(st.recent_values) >= max(2, window_k // 2):
        stable_bad = (sum(st.recent_values) / len(st.recent_values)) < (thr + margin)
    low_unc = False
    if len(st.recent_entropies) >= max(2, window_k // 2):
        low_unc = (sum(st.recent_entropies) / len(st.recent_entropies)) < min_entropy
    return st.consec_bad >= seq_bad and (stable_bad or low_unc)


def should_adjudicate_draw(board: ch.Board, moves: List[ch.Move], cfg: dict) -> bool:
    """azchess/draw.py:8-84"""
    if board.is_insufficient_material():
        return True
    if board.can_claim_fifty_moves():
        return True
    if board.is_repetition(3) or board.can_claim_threefold_repetition():
        return True
    if bool(cfg.get("stalemate_draw", True)) and board.is_stalemate():
        return True
    if not bool(cfg.get("enabled", False)):
        return False
    if len(moves) < int(cfg.get("min_plies", 30)):
        return False
    window = int(cfg.get("window", 12))
    min_unique = int(cfg.get("min_unique", 3))
    if window > 0 and min_unique > 0 and len(moves) >= window:
        if len(set(str(m) for m in moves[-window:])) < min_unique:
            return True
    cap = int(cfg.get("halfmove_cap", 50))
    if cap and board.halfmove_clock >= cap:
        return True
    thr = int(cfg.get("material_draw_threshold", 10))
    if thr > 0:
        mat = 0
        for color in (True, False):
            mat += len(board.pieces(ch.PAWN, color)) + 3 * len(board.pieces(ch.KNIGHT, color)) + \
                3 * len(board.pieces(ch.BISHOP, color)) + 5 * len(board.pieces(ch.ROOK, color)) + \
                9 * len(board.pieces(ch.QUEEN, color))
        if mat <= thr:
            return True
    return False


def game_result(board: ch.Board) -> float:
    """selfplay/internal.py:738-750"""
    if board.is_checkmate():
        return -1.0 if board.turn else 1.0
    res = board.result(claim_draw=True)
    return 1.0 if res == "1-0" else (-1.0 if res == "0-1" else 0.0)
