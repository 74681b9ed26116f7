"""Training-side reader (matrix0_amd/npz_dataset.py) over shards written by this package, and -- in the build container,
where /root/reference exists -- the REFERENCE's own DataManager reading the same shards (the drop-in direction that matters:
the reference trainer consumes what the engine writes)."""
import os
import sqlite3
import sys

import numpy as np
import pytest

from matrix0_amd.data_writer import ReplayShardWriter
from matrix0_amd.npz_dataset import NPZBatchIterableDataset, ReplayReader


def _game(rng, n, tag, with_mask=True):
    pi = rng.random((n, 4672)).astype(np.float32)
    pi /= pi.sum(axis=1, keepdims=True)
    g = {"s": np.full((n, 19, 8, 8), tag, np.float32), "pi": pi, "z": rng.uniform(-1, 1, n).astype(np.float32)}
    if with_mask:
        g["legal_mask"] = (pi > np.median(pi)).astype(np.uint8)
    return g


def _write(tmp_path, rng, games=6, n=40, shard_size=64):
    w = ReplayShardWriter(base_dir=str(tmp_path), shard_size=shard_size, max_shards=100)
    for k in range(games):
        w.add_game(_game(rng, n, float(k + 1)))
    w.close()
    return w


def test_reader_yields_every_sample_once_per_epoch_in_full_batches(tmp_path):
    rng = np.random.default_rng(0)
    w = _write(tmp_path, rng)                       # 240 samples in shards of 64, 64, 64, 48
    r = ReplayReader(str(tmp_path), seed=1)
    assert sorted(r.valid_shards()) == sorted(w.written)
    batches = list(NPZBatchIterableDataset(r, 32, epochs=1))
    assert len(batches) == 240 // 32
    for b in batches:
        assert len(b) == 4
        s, pi, z, lm = b
        assert s.shape == (32, 19, 8, 8) and s.dtype == np.float32 and pi.shape == (32, 4672) and z.shape == (32,)
        assert lm.shape == (32, 4672) and lm.dtype == np.uint8
        assert np.allclose(pi.sum(1), 1.0, atol=1e-4)
    tags = np.concatenate([b[0][:, 0, 0, 0] for b in batches])
    assert len(tags) == 224 and set(np.unique(tags)) <= {1.0, 2.0, 3.0, 4.0, 5.0, 6.0}


def test_bad_shards_are_marked_corrupted_and_skipped(tmp_path):
    rng = np.random.default_rng(1)
    w = _write(tmp_path, rng, games=4, n=64, shard_size=64)
    bad = w.written[1]
    with np.load(bad) as d:
        blob = {k: d[k] for k in d.files}
    blob["pi"][3, 7] = np.nan
    np.savez_compressed(bad, **blob)
    r = ReplayReader(str(tmp_path), seed=2)
    got = list(r.get_training_batch(64, epochs=1))
    assert len(got) == 3
    conn = sqlite3.connect(str(tmp_path / "data_metadata.db"))
    assert conn.execute("SELECT corrupted FROM shards WHERE path = ?", (bad,)).fetchone()[0] == 1
    conn.close()
    assert bad not in ReplayReader(str(tmp_path)).valid_shards()


@pytest.mark.skipif(not os.path.isdir("/root/reference/azchess"), reason="the reference exists in the build container only")
def test_reference_datamanager_reads_our_shards(tmp_path):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import refshim
    refshim.install()
    import logging
    logging.disable(logging.CRITICAL)
    from azchess.data_manager import DataManager
    rng = np.random.default_rng(3)
    _write(tmp_path, rng, games=5, n=64, shard_size=64)
    dm = DataManager(base_dir=str(tmp_path), expected_planes=19)
    it = dm.get_training_batch(64, "cpu")
    seen = []
    for _ in range(5):
        b = next(it)
        assert len(b) == 4 and b[0].shape == (64, 19, 8, 8) and b[1].shape == (64, 4672) and b[3].shape == (64, 4672)
        seen.append(b[0][:, 0, 0, 0])
    assert set(np.unique(np.concatenate(seen))) <= {1.0, 2.0, 3.0, 4.0, 5.0}
    stats = [s for s in dm._get_all_shards() if not s.corrupted]
    assert len(stats) == 5 and all(s.sample_count == 64 for s in stats)
