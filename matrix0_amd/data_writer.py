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
