"""Shard writer vs the reference's on-disk contract (data_manager.py:198-243, 105-131)."""
import sqlite3

import numpy as np

from matrix0_amd.data_writer import SelfplayShardWriter


def test_writer_npz_and_sqlite_row(tmp_path):
    w = SelfplayShardWriter(base_dir=str(tmp_path))
    T = 5
    pi = np.zeros((T, 4672), np.float32); pi[:, 7] = 1.0
    data = {"s": np.zeros((T, 19, 8, 8), np.float32), "pi": pi, "z": np.ones((T,), np.float32),
            "meta_moves": np.array([T], np.int32), "legal_mask": np.zeros((T, 4672), np.uint8)}
    path = w.add_selfplay_data(data, worker_id=0, game_id=3)
    assert path.endswith(".npz") and "/selfplay/selfplay_" in path
    z = np.load(path)
    assert z["s"].shape == (T, 19, 8, 8) and z["s"].dtype == np.float32 and z["pi"].dtype == np.float32
    assert z["legal_mask"].dtype == np.uint8 and z["meta_moves"].dtype == np.int32
    rows = sqlite3.connect(str(tmp_path / "data_metadata.db")).execute(
        "SELECT path, sample_count, version, source, corrupted, length(checksum) FROM shards").fetchall()
    assert rows == [(path, T, "1.0.0", "selfplay", 0, 64)]
    assert w.validate_policy_targets(pi) and not w.validate_policy_targets(pi * 2)
