"""ReplayShardWriter: the replay-buffer format of DataManager.compact_selfplay_to_replay / add_training_data
(azchess/data_manager.py:245-262, 1378-1493): shard size, order, keys, SQLite rows, backups, pruning."""
import sqlite3

import numpy as np

from matrix0_amd.data_writer import ReplayShardWriter, SelfplayShardWriter


def _game(rng, n, with_mask=True, tag=0.0):
    pi = rng.random((n, 4672)).astype(np.float32)
    pi /= pi.sum(axis=1, keepdims=True)
    d = {"s": np.full((n, 19, 8, 8), tag, np.float32), "pi": pi, "z": np.full((n,), tag, np.float32),
         "meta_moves": np.array([n], np.int32), "meta_result": np.array([0.0], np.float32),
         "meta_resigned": np.array([0], np.int8), "meta_draw": np.array([1], np.int8),
         "meta_avg_policy_entropy": np.array([2.0], np.float32), "meta_avg_sims": np.array([800.0], np.float32)}
    if with_mask:
        d["legal_mask"] = (rng.random((n, 4672)) < 0.01).astype(np.uint8)
    return d


def _rows(base):
    conn = sqlite3.connect(base / "data_metadata.db")
    rows = conn.execute("SELECT path, sample_count, source, version FROM shards ORDER BY path").fetchall()
    conn.close()
    return rows


def test_compaction_of_per_game_files(tmp_path):
    rng = np.random.default_rng(0)
    sp = SelfplayShardWriter(str(tmp_path))
    lens = [7, 12, 5, 9, 11]
    paths = [sp.add_selfplay_data(_game(rng, n, tag=float(i + 1)), 0, i) for i, n in enumerate(lens)]
    w = ReplayShardWriter(str(tmp_path), max_shards=128, shard_size=16)
    assert w.compact_selfplay_to_replay() == len(lens)
    shards = [np.load(p) for p in w.written]
    assert [s["s"].shape[0] for s in shards] == [16, 16, 12]                 # 44 samples: two full shards + the tail
    for s in shards:
        assert set(s.files) == {"s", "pi", "z", "legal_mask"}
        assert s["s"].dtype == np.float32 and s["legal_mask"].dtype == np.uint8 and s["s"].shape[1:] == (19, 8, 8)
    # sample order = games in sorted file-name order, rows in game order
    order = sorted(range(len(lens)), key=lambda i: paths[i].split("/")[-1])
    expect = np.concatenate([np.full(lens[i], float(i + 1), np.float32) for i in order])
    assert np.array_equal(np.concatenate([s["z"] for s in shards]), expect)
    # sources moved to backups, their rows dropped, replay rows present
    assert not list((tmp_path / "selfplay").glob("*.npz"))
    assert len(list((tmp_path / "backups").glob("*.npz"))) == len(lens)
    rows = _rows(tmp_path)
    assert len(rows) == 3 and all(r[2] == "selfplay" and r[3] == "1.0.0" for r in rows)
    assert sorted(r[1] for r in rows) == [12, 16, 16]
    assert all("/replays/replays_" in r[0] for r in rows)
    assert w.summary["games"] == 5 and w.summary["moves"] == sum(lens) and w.summary["draws"] == 5


def test_direct_emission_mask_dropped_if_any_game_lacks_it_and_pruning(tmp_path):
    rng = np.random.default_rng(1)
    w = ReplayShardWriter(str(tmp_path), max_shards=2, shard_size=8)
    w.add_game(_game(rng, 6, with_mask=True))
    w.add_game(_game(rng, 6, with_mask=False))         # the buffer now cannot supply masks for every row
    w.add_game(_game(rng, 6, with_mask=True))
    assert len(w.written) == 2 and [np.load(p)["s"].shape[0] for p in w.written] == [8, 8]
    assert "legal_mask" not in np.load(w.written[0]).files
    w.close()
    assert len(w.written) == 3
    # only the newest max_shards replay shards survive (created_at ties are broken arbitrarily: count only)
    left = list((tmp_path / "replays").glob("*.npz"))
    assert len(left) == 2 and len(_rows(tmp_path)) == 2


def test_legal_mask_is_decided_per_shard(tmp_path):
    """A game without legal_mask must not cost later shards theirs: the key is dropped only from the shard(s) that
    contain rows of that game (round-1 advisor finding)."""
    from matrix0_amd.data_writer import ReplayShardWriter
    w = ReplayShardWriter(base_dir=str(tmp_path), shard_size=64, max_shards=100)

    def game(n, with_mask, tag):
        g = {"s": np.full((n, 19, 8, 8), tag, np.float32), "pi": np.full((n, 4672), 1.0 / 4672, np.float32),
             "z": np.zeros(n, np.float32)}
        if with_mask:
            g["legal_mask"] = np.ones((n, 4672), np.uint8)
        return g

    w.add_game(game(10, False, 1.0))          # mask-less rows land in shard 0
    for k in range(4):
        w.add_game(game(50, True, 2.0 + k))   # 200 masked rows: shards 0..2 full, 18 rows left
    w.close()
    shards = [np.load(p) for p in w.written]
    assert [int(s["s"].shape[0]) for s in shards] == [64, 64, 64, 18]
    assert "legal_mask" not in shards[0].files
    for s in shards[1:]:
        assert "legal_mask" in s.files and s["legal_mask"].shape == (s["s"].shape[0], 4672) and s["legal_mask"].all()
