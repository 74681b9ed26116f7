"""Training-side reader of the shards this package writes: the replay path of the reference's trainer input
(azchess/training/npz_dataset.py:15-109 NPZBatchIterableDataset(mode="replay") over DataManager.get_training_batch /
_iter_shard_samples, azchess/data_manager.py:264-470).  Same contract: shards come from the `shards` table of
<base>/data_metadata.db (rows marked corrupted are skipped), a shard must hold `s` float32 (T,planes,8,8), `pi` float32
(T,4672) and `z` float32 (T,) or (T,1) without NaN/Inf or it is MARKED corrupted and skipped (data_manager.py:411-424,
1861-1946), `legal_mask` is reshaped to (T,4672) uint8, samples are shuffled within a shard and shards within an epoch, and
only full batches are yielded, as tuples (s, pi, z, legal_mask) -- or (s, pi, z) when some row of the batch has no mask.
Pure data plumbing (numpy + sqlite3): no GPU code here."""
from __future__ import annotations

import sqlite3
from pathlib import Path
from typing import Iterator, List, Optional, Tuple

import numpy as np


class ReplayReader:
    def __init__(self, base_dir: str = "data", expected_planes: int = 19, seed: Optional[int] = None):
        self.base_dir = Path(base_dir)
        self.db_path = self.base_dir / "data_metadata.db"
        self.expected_planes = int(expected_planes)
        self.rng = np.random.default_rng(seed)

    # -- shard table ----------------------------------------------------------------------------------------------
    def valid_shards(self, sources: Tuple[str, ...] = ("selfplay", "")) -> List[str]:
        if not self.db_path.exists():
            return []
        conn = sqlite3.connect(str(self.db_path), timeout=30)
        rows = conn.execute("SELECT path, source FROM shards WHERE corrupted = 0 OR corrupted IS NULL").fetchall()
        conn.close()
        return [p for p, src in rows if (src or "") in sources and Path(p).exists()]

    def mark_corrupted(self, path: str) -> None:
        conn = sqlite3.connect(str(self.db_path), timeout=30)
        conn.execute("UPDATE shards SET corrupted = 1 WHERE path = ?", (path,))
        conn.commit(); conn.close()

    # -- validation (data_manager.py _validate_shapes / _validate_dtypes_and_ranges) ------------------------------------
    def _load(self, path: str):
        with np.load(path) as d:
            if not all(k in d.files for k in ("s", "pi", "z")):
                return None
            s, pi, z = d["s"], d["pi"], d["z"]
            lm = d["legal_mask"] if "legal_mask" in d.files else None
        if z.ndim == 2 and z.shape[1] == 1:
            z = z.reshape(z.shape[0])
        ok = (s.dtype == np.float32 and pi.dtype == np.float32 and z.dtype == np.float32 and s.ndim == 4 and
              s.shape[1:] == (self.expected_planes, 8, 8) and pi.ndim == 2 and pi.shape[1] in (4672, 1858) and z.ndim == 1 and
              s.shape[0] == pi.shape[0] == z.shape[0] and s.shape[0] > 0)
        if ok:
            ok = bool(np.isfinite(s).all() and np.isfinite(pi).all() and np.isfinite(z).all())
        if not ok:
            return None
        if lm is not None:
            try:
                lm = lm.reshape(lm.shape[0], -1).astype(np.uint8, copy=False)
                if lm.shape != (s.shape[0], pi.shape[1]):
                    lm = None
            except Exception:
                lm = None
        return s, pi, z, lm

    # -- iteration ------------------------------------------------------------------------------------------------
    def iter_samples(self, epochs: Optional[int] = None) -> Iterator[Tuple[np.ndarray, np.ndarray, np.float32, Optional[np.ndarray]]]:
        paths = self.valid_shards()
        e = 0
        while paths and (epochs is None or e < epochs):
            order = list(paths)
            self.rng.shuffle(order)
            progressed = False
            for path in order:
                try:
                    got = self._load(path)
                except Exception:
                    got = None
                if got is None:
                    self.mark_corrupted(path)
                    paths.remove(path)
                    continue
                s, pi, z, lm = got
                idx = self.rng.permutation(s.shape[0])
                for i in idx:
                    progressed = True
                    yield s[i], pi[i], z[i], (lm[i] if lm is not None else None)
            if not progressed:
                return
            e += 1

    def get_training_batch(self, batch_size: int, epochs: Optional[int] = None):
        """Full batches only; tuples (s, pi, z, legal_mask) or (s, pi, z) -- data_manager.py:334-392."""
        if not self.valid_shards():
            raise RuntimeError("No valid training data available")
        buf = []
        for sample in self.iter_samples(epochs):
            buf.append(sample)
            if len(buf) == batch_size:
                s = np.ascontiguousarray(np.stack([b[0] for b in buf]), dtype=np.float32)
                pi = np.ascontiguousarray(np.stack([b[1] for b in buf]), dtype=np.float32)
                z = np.ascontiguousarray(np.array([b[2] for b in buf], dtype=np.float32))
                if all(b[3] is not None for b in buf):
                    yield s, pi, z, np.ascontiguousarray(np.stack([b[3] for b in buf]), dtype=np.uint8)
                else:
                    yield s, pi, z
                buf = []


class NPZBatchIterableDataset:
    """training/npz_dataset.py:15-82, mode 'replay' (the curriculum modes mix external Stockfish shards, which this package
    does not produce).  Iterating yields ready-to-train numpy batches; wrap it in a torch DataLoader(batch_size=None) exactly
    as build_training_dataloader does (npz_dataset.py:85-109)."""

    def __init__(self, reader: ReplayReader, batch_size: int, mode: str = "replay", epochs: Optional[int] = None):
        if mode != "replay":
            raise NotImplementedError("only mode='replay' reads self-play shards; curriculum phases need external data")
        self.reader, self.batch_size, self.epochs = reader, int(batch_size), epochs

    def __iter__(self):
        return self.reader.get_training_batch(self.batch_size, self.epochs)
