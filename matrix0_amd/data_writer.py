"""Self-play shard writer: the slice of DataManager the worker uses
(azchess/data_manager.py:198-243 _save_npz_shard / add_selfplay_data, :105-131 schema, :1622-1638 _record_shard).

Same on-disk contract: `<base>/selfplay/selfplay_<YYYYmmdd_HHMMSS>_<uuid8>.npz` written atomically
with np.savez_compressed, one row per shard in `<base>/data_metadata.db` table `shards`."""
from __future__ import annotations

import hashlib
import os
import sqlite3
import tempfile
import time
import uuid
from datetime import datetime
from pathlib import Path
from typing import Dict

import numpy as np

VERSION = "1.0.0"


class SelfplayShardWriter:
    def __init__(self, base_dir: str = "data"):
        self.base_dir = Path(base_dir)
        self.selfplay_dir = self.base_dir / "selfplay"
        self.selfplay_dir.mkdir(parents=True, exist_ok=True)
        self.db_path = self.base_dir / "data_metadata.db"
        conn = self._connect()
        conn.execute("""CREATE TABLE IF NOT EXISTS shards (
            path TEXT PRIMARY KEY, size_bytes INTEGER, sample_count INTEGER, created_at TEXT, checksum TEXT,
            version TEXT, source TEXT, corrupted BOOLEAN DEFAULT FALSE, last_accessed TEXT)""")
        conn.execute("CREATE TABLE IF NOT EXISTS data_stats (key TEXT PRIMARY KEY, value TEXT, updated_at TEXT)")
        conn.commit()
        conn.close()

    def _connect(self):
        conn = sqlite3.connect(self.db_path, timeout=30)
        conn.execute("PRAGMA journal_mode=WAL")
        conn.execute("PRAGMA synchronous=NORMAL")
        conn.execute("PRAGMA busy_timeout=30000")
        return conn

    @staticmethod
    def validate_policy_targets(pi: np.ndarray) -> bool:
        """data_manager.py:1948-2007 (warn-only there): rows non-negative, finite, summing to 1 +- 0.01."""
        if pi.ndim != 2 or not np.all(np.isfinite(pi)) or np.any(pi < 0):
            return False
        return bool(np.all(np.abs(pi.sum(axis=1) - 1.0) <= 0.01))

    def add_selfplay_data(self, data: Dict[str, np.ndarray], worker_id: int, game_id: int) -> str:
        ts = datetime.now().strftime("%Y%m%d_%H%M%S")
        path = self.selfplay_dir / f"{self.selfplay_dir.name}_{ts}_{uuid.uuid4().hex[:8]}.npz"
        for attempt in range(3):
            try:
                with tempfile.NamedTemporaryFile(dir=str(self.selfplay_dir), suffix=".npz.tmp", delete=False) as tf:
                    tmp = Path(tf.name)
                    np.savez_compressed(tf, **data)
                os.replace(tmp, path)
                break
            except Exception:
                if attempt == 2:
                    raise
                time.sleep(0.1 * (attempt + 1))
        h = hashlib.sha256()
        with open(path, "rb") as f:
            for chunk in iter(lambda: f.read(1 << 20), b""):
                h.update(chunk)
        n = int(np.asarray(data["s"]).shape[0]) if "s" in data else 0
        conn = self._connect()
        conn.execute("INSERT OR REPLACE INTO shards (path, size_bytes, sample_count, created_at, checksum, version, source, "
                     "last_accessed) VALUES (?, ?, ?, ?, ?, ?, ?, ?)",
                     (str(path), path.stat().st_size, n, ts, h.hexdigest(), VERSION, "selfplay", ts))
        conn.commit()
        conn.close()
        return str(path)


class ReplayShardWriter(SelfplayShardWriter):
    """Replay-buffer side of the same store: what `DataManager.compact_selfplay_to_replay`
    (azchess/data_manager.py:1378-1493) and `add_training_data` (:245-262) produce -- `<base>/replays/
    replays_<YYYYmmdd_HHMMSS>_<uuid8>.npz` shards of exactly `shard_size` samples (the tail shard is shorter) with keys
    `s f32[N,19,8,8]`, `pi f32[N,4672]`, `z f32[N]` and `legal_mask u8[N,4672]` when every game carried one, one row per
    shard in `shards` with source "selfplay", oldest replay shards pruned beyond `max_shards` (:1263-1293).

    Two ways in, same bytes out: `compact_selfplay_to_replay()` folds the per-game files of `<base>/selfplay/` in sorted
    name order (sources are moved to `<base>/backups/`, their rows dropped); `add_game()` takes finished games directly
    from the engine (no per-game file at all) -- call `close()` to flush the tail."""

    def __init__(self, base_dir: str = "data", max_shards: int = 128, shard_size: int = 16384):
        super().__init__(base_dir)
        self.replays_dir = self.base_dir / "replays"
        self.backups_dir = self.base_dir / "backups"
        self.replays_dir.mkdir(parents=True, exist_ok=True)
        self.backups_dir.mkdir(parents=True, exist_ok=True)
        self.max_shards = int(max_shards)
        self.shard_size = int(shard_size)
        self._buf = {"s": [], "pi": [], "z": [], "legal_mask": [], "_has_mask": []}
        self._count = 0
        # legal_mask is decided per flushed shard: written iff every row of that shard came from a game that carried one
        # (the reference concatenates the masks of a shard's games or drops the key for that shard)
        self.summary = {"games": 0, "moves": 0, "resigned": 0, "draws": 0, "entropy": 0.0, "avg_sims": 0.0}
        self.written = []

    # -- intake ---------------------------------------------------------------------------------------------------
    def add_game(self, data: Dict[str, np.ndarray]) -> None:
        s, pi, z = np.asarray(data["s"]), np.asarray(data["pi"]), np.asarray(data["z"])
        if not (s.shape[0] == pi.shape[0] == z.shape[0]):
            raise ValueError("s / pi / z row counts differ")
        self._buf["s"].append(s); self._buf["pi"].append(pi); self._buf["z"].append(z)
        lm = data.get("legal_mask")
        have = lm is not None and np.asarray(lm).shape[0] == s.shape[0]
        self._buf["legal_mask"].append(np.asarray(lm).reshape(s.shape[0], -1).astype(np.uint8, copy=False) if have
                                       else np.zeros((s.shape[0], pi.shape[1]), np.uint8))
        self._buf["_has_mask"].append(np.full(s.shape[0], have, dtype=bool))
        self._count += int(s.shape[0])
        g = self.summary
        g["games"] += 1
        g["moves"] += int(np.asarray(data.get("meta_moves", [s.shape[0]])).reshape(-1)[0])
        g["resigned"] += int(np.asarray(data.get("meta_resigned", [0])).reshape(-1)[0])
        g["draws"] += int(np.asarray(data.get("meta_draw", [0])).reshape(-1)[0])
        g["entropy"] += float(np.asarray(data.get("meta_avg_policy_entropy", [0.0])).reshape(-1)[0])
        g["avg_sims"] += float(np.asarray(data.get("meta_avg_sims", [0.0])).reshape(-1)[0])
        while self._count >= self.shard_size:
            self._flush(self.shard_size)

    def close(self) -> None:
        if self._count > 0:
            self._flush(self._count)
        self.cleanup_old_shards(self.max_shards)

    # -- compaction of existing per-game files --------------------------------------------------------------------
    def compact_selfplay_to_replay(self) -> int:
        files = sorted(p for p in self.selfplay_dir.glob("*.npz") if p.is_file())
        for f in files:
            try:
                with np.load(f) as d:
                    game = {k: d[k] for k in d.files}
            except Exception:
                self._mark_corrupted(str(f))
                continue
            self.add_game(game)
            (self.backups_dir / f.name).write_bytes(f.read_bytes())
            f.unlink()
            conn = self._connect()
            conn.execute("DELETE FROM shards WHERE path = ?", (str(f),))
            conn.commit(); conn.close()
        self.close()
        return len(files)

    # -- internals ------------------------------------------------------------------------------------------------
    def _flush(self, take: int) -> None:
        cat = {k: np.concatenate(v, axis=0) for k, v in self._buf.items() if v}
        payload = {"s": cat["s"][:take], "pi": cat["pi"][:take], "z": cat["z"][:take]}
        if bool(cat["_has_mask"][:take].all()):
            payload["legal_mask"] = cat["legal_mask"][:take].astype(np.uint8, copy=False)
        for k in ("s", "pi", "z", "legal_mask", "_has_mask"):
            self._buf[k] = [cat[k][take:]]
        self._count -= take
        self.written.append(self._write_shard(payload, self.replays_dir, "selfplay"))

    def _write_shard(self, data: Dict[str, np.ndarray], dir_path: Path, source: str) -> str:
        ts = datetime.now().strftime("%Y%m%d_%H%M%S")
        path = dir_path / f"{dir_path.name}_{ts}_{uuid.uuid4().hex[:8]}.npz"
        with tempfile.NamedTemporaryFile(dir=str(dir_path), suffix=".npz.tmp", delete=False) as tf:
            tmp = Path(tf.name)
            np.savez_compressed(tf, **data)
        os.replace(tmp, path)
        h = hashlib.sha256()
        with open(path, "rb") as f:
            for chunk in iter(lambda: f.read(1 << 20), b""):
                h.update(chunk)
        conn = self._connect()
        conn.execute("INSERT OR REPLACE INTO shards (path, size_bytes, sample_count, created_at, checksum, version, source, "
                     "last_accessed) VALUES (?, ?, ?, ?, ?, ?, ?, ?)",
                     (str(path), path.stat().st_size, int(data["s"].shape[0]), ts, h.hexdigest(), VERSION, source, ts))
        conn.commit(); conn.close()
        return str(path)

    def _mark_corrupted(self, path: str) -> None:
        conn = self._connect()
        conn.execute("UPDATE shards SET corrupted = 1 WHERE path = ?", (path,))
        conn.commit(); conn.close()

    def cleanup_old_shards(self, keep_recent: int) -> None:
        """Only replay shards under <base>/replays, never stockfish-tagged ones; newest `keep_recent` stay."""
        root = str(self.replays_dir.resolve())
        conn = self._connect()
        rows = conn.execute("SELECT path, created_at, source FROM shards").fetchall()
        elig = [(p, c) for p, c, src in rows
                if str(Path(p).resolve()).startswith(root) and not (src or "").startswith("stockfish:")]
        elig.sort(key=lambda r: r[1], reverse=True)
        for p, _ in elig[keep_recent:]:
            Path(p).unlink(missing_ok=True)
            conn.execute("DELETE FROM shards WHERE path = ?", (p,))
        conn.commit(); conn.close()
