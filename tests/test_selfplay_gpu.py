"""End-to-end self-play on the GPU with a small network: games are replayed through the oracle's rules,
records checked against the reference's NPZ contract (selfplay/internal.py:628-651)."""
import numpy as np
import pytest

from oracle import chess_py as ch
from oracle import mcts_ref as ref
from oracle import net_ref

pytestmark = pytest.mark.gpu

NET = dict(planes=19, channels=32, blocks=2, attention_heads=2, policy_size=4672, norm="group", activation="silu",
           preact=True, policy_factor_rank=16, self_supervised=False)
CFG = {"seed": 99,
       "mcts": {"cpuct": 2.5, "cpuct_start": 3.0, "cpuct_end": 2.0, "cpuct_plies": 40, "dirichlet_plies": 30,
                "selection_jitter": 0.05, "fpu_reduction": 0.1, "draw_penalty": -0.05, "legal_softmax": True,
                "inference_batch_size": 8, "playout_random_frac": 0.05},
       "selfplay": {"num_simulations": 48, "max_game_len": 40, "min_resign_plies": 50, "resign_threshold": -0.85,
                    "opening_random_plies": 6, "temperature_start": 1.2, "temperature_end": 0.3, "temperature_moves": 40,
                    "draw": {"min_plies": 30, "window": 8, "min_unique": 4, "halfmove_cap": 100}}}


def test_selfplay_games_are_legal_and_records_follow_the_contract():
    from matrix0_amd.backend import M0Backend
    from matrix0_amd import engine as eng
    be = M0Backend.from_state_dict(NET, net_ref.random_state_dict(NET, seed=1))
    cfg = eng.selfplay_cfg_from_dict(CFG, concurrent_games=6, total_games=9)
    e = eng.SelfplayEngine(be, cfg)
    games = []
    for _ in range(4000):
        e.step(8)
        while True:
            r = e.poll()
            if r is None:
                break
            games.append(r)
        if not e.running():
            break
    st = e.stats()
    assert len(games) == 9 and st["games_finished"] == 9 and st["active_games"] == 0
    assert sorted(g["game_index"] for g in games) == list(range(9))
    assert st["evals"] > 0 and st["plies"] == sum(g["moves"] for g in games)
    for g in games:
        T = g["moves"]
        assert 1 <= T <= 40
        assert g["s"].shape == (T, 19, 8, 8) and g["s"].dtype == np.float32
        assert g["pi"].shape == (T, 4672) and g["pi"].dtype == np.float32
        assert g["z"].shape == (T,) and g["legal_mask"].shape == (T, 4672) and g["legal_mask"].dtype == np.uint8
        np.testing.assert_allclose(g["pi"].sum(axis=1), 1.0, atol=1e-4)
        assert np.all(g["pi"] >= 0) and np.all(g["pi"][g["legal_mask"] == 0] == 0)
        # replay through the oracle: legality, planes, masks, result
        b = ch.Board()
        played = [ch.Move.from_uci(u) for u in g["played"]]
        n_open = len(played) - T + (1 if False else 0)
        # plies recorded = searched plies; the last searched ply's move is played unless the game resigned there
        n_open = len(played) - (T - (1 if g["resigned"] else 0))
        assert 0 <= n_open <= 6
        moves = []
        for i, m in enumerate(played):
            assert m in b.legal_moves, (i, m)
            if i >= n_open:
                t = i - n_open
                assert np.array_equal(g["s"][t], ch.encode_board(b))
                assert np.array_equal(g["legal_mask"][t].astype(bool), ch.get_legal_actions(b))
                assert g["pi"][t][ch.move_to_index(b, m)] > 0          # the played move was visited
            b.push(m)
            moves.append(m)
        turn0 = 1 if (n_open % 2 == 0) else -1
        signs = np.array([turn0 * (1 if t % 2 == 0 else -1) for t in range(T)], np.float32)
        if not g["resigned"]:
            ended = b.is_game_over() or T >= 40 or ref.should_adjudicate_draw(b, moves, CFG["selfplay"]["draw"])
            assert ended
            if b.is_game_over(claim_draw=True):
                assert g["result"] == ref.game_result(b)
            else:
                assert abs(g["result"] - g["search_values"][-1]) < 1e-6      # length cap: last root_q (internal.py:594-597)
        np.testing.assert_allclose(g["z"], g["result"] * signs, atol=1e-6)
        assert g["draw"] == (g["result"] == 0.0)


def test_selfplay_is_deterministic_for_a_seed():
    from matrix0_amd.backend import M0Backend
    from matrix0_amd import engine as eng
    be = M0Backend.from_state_dict(NET, net_ref.random_state_dict(NET, seed=1))
    outs = []
    for rep in range(2):
        cfg = eng.selfplay_cfg_from_dict(CFG, concurrent_games=3, total_games=3)
        e = eng.SelfplayEngine(be, cfg)
        games = {}
        while e.running():
            e.step(16)
            while (r := e.poll()) is not None:
                games[r["game_index"]] = r
        outs.append(games)
        e.close()
    # the three games race for batch rows, so this also needs a network whose results do not depend on the row
    # (tests/test_net_gpu.py::test_narrow_networks_do_not_depend_on_the_position_in_the_batch): every record bit for bit
    assert sorted(outs[0]) == sorted(outs[1]) == [0, 1, 2]
    for g in outs[0]:
        assert outs[0][g]["played"] == outs[1][g]["played"], g
        for k in ("pi", "z", "search_values", "s", "legal_mask"):
            assert np.array_equal(outs[0][g][k], outs[1][g][k]), (g, k)


def _play_all(e):
    games = {}
    for _ in range(6000):
        e.step(8)
        while (r := e.poll()) is not None:
            games[r["game_index"]] = r
        if not e.running():
            break
    return games


def test_two_engines_on_two_streams_play_the_same_games():
    """engine.SelfplayPool: the games of a GPU shared by two independent engines (own network instance and HIP stream),
    stepped concurrently from two host threads.  A game depends only on (seed, game index) and the network is
    batch-invariant, so every record must be bit-identical to the single-engine run."""
    from matrix0_amd.backend import M0Backend
    from matrix0_amd import engine as eng
    # a 320-channel network: the configuration the engines ship with (both paths are bitwise batch-invariant: test_net_gpu)
    net = dict(NET, channels=320, blocks=3, attention_heads=20, policy_factor_rank=128)
    sd = net_ref.random_state_dict(net, seed=2)
    one = eng.SelfplayEngine(M0Backend.from_state_dict(net, sd), eng.selfplay_cfg_from_dict(CFG, concurrent_games=6, total_games=10))
    ref_games = _play_all(one)
    pool = eng.SelfplayPool(lambda: M0Backend.from_state_dict(net, sd), CFG, streams=2, concurrent_games=6, total_games=10)
    games = _play_all(pool)
    st = pool.stats()
    pool.close()
    assert sorted(games) == sorted(ref_games) == list(range(10)) and st["games_finished"] == 10
    for i in range(10):
        a, b = ref_games[i], games[i]
        assert a["played"] == b["played"] and a["result"] == b["result"] and a["moves"] == b["moves"], i
        assert np.array_equal(a["pi"], b["pi"]) and np.array_equal(a["z"], b["z"]) and np.array_equal(a["s"], b["s"]), i


def test_drop_in_worker_writes_shards_and_queue_messages(tmp_path):
    """selfplay_worker(proc_id, cfg_dict, ckpt_path, games, q): same call as the reference (internal.py:94-95);
    checks the queue message schema (internal.py:665-679) and the NPZ/SQLite contract (internal.py:628-653)."""
    import queue
    import sqlite3
    from matrix0_amd.selfplay import selfplay_worker
    cfg = dict(CFG, model=NET, data_dir=str(tmp_path), engine={"concurrent_games": 3, "leaves_per_step": 8})
    q = queue.Queue()
    selfplay_worker(0, cfg, None, 4, q, None)
    msgs = []
    while not q.empty():
        msgs.append(q.get())
    games = [m for m in msgs if m["type"] == "game"]
    assert len(games) == 4
    want = {"type", "proc", "file", "moves", "result", "secs", "resigned", "resigner", "draw", "avg_policy_entropy",
            "avg_ms_per_move", "avg_sims"}
    for m in games:
        assert set(m) == want and m["proc"] == 0
        z = np.load(m["file"])
        T = m["moves"]
        assert z["s"].shape == (T, 19, 8, 8) and z["s"].dtype == np.float32
        assert z["pi"].shape == (T, 4672) and z["pi"].dtype == np.float32 and z["z"].shape == (T,)
        assert z["legal_mask"].shape == (T, 4672) and z["legal_mask"].dtype == np.uint8
        assert z["meta_moves"].dtype == np.int32 and int(z["meta_moves"][0]) == T
        assert z["meta_result"].dtype == np.float32 and z["meta_resigned"].dtype == np.int8 and z["meta_draw"].dtype == np.int8
        assert z["meta_avg_policy_entropy"].dtype == np.float32 and z["meta_avg_sims"].dtype == np.float32
        np.testing.assert_allclose(z["pi"].sum(axis=1), 1.0, atol=1e-3)
    n = sqlite3.connect(str(tmp_path / "data_metadata.db")).execute("SELECT count(*), sum(sample_count) FROM shards").fetchone()
    assert n[0] == 4 and n[1] == sum(m["moves"] for m in games)


def test_worker_direct_replay_shards(tmp_path):
    """engine.replay_shards: the worker writes replay-buffer shards (data_manager.py:245-262, 1378-1493 format)
    instead of one NPZ per game; the samples are the games' rows in completion order."""
    import queue
    import sqlite3
    from matrix0_amd.selfplay import selfplay_worker
    cfg = dict(CFG, model=NET, data_dir=str(tmp_path),
               engine={"concurrent_games": 3, "leaves_per_step": 8, "replay_shards": True, "shard_size": 32})
    q = queue.Queue()
    selfplay_worker(0, cfg, None, 4, q, None)
    games = []
    while not q.empty():
        m = q.get()
        if m["type"] == "game":
            games.append(m)
    assert len(games) == 4 and all(m["file"] is None for m in games)
    total = sum(m["moves"] for m in games)
    assert not list((tmp_path / "selfplay").glob("*.npz"))
    shards = sorted((tmp_path / "replays").glob("replays_*.npz"))
    sizes = sorted(np.load(p)["s"].shape[0] for p in shards)
    assert sum(sizes) == total and all(n == 32 for n in sizes[1:]) and 0 < sizes[0] <= 32
    for p in shards:
        z = np.load(p)
        assert set(z.files) == {"s", "pi", "z", "legal_mask"} and z["legal_mask"].dtype == np.uint8
        np.testing.assert_allclose(z["pi"].sum(axis=1), 1.0, atol=1e-3)
    rows = sqlite3.connect(str(tmp_path / "data_metadata.db")).execute("SELECT sum(sample_count), count(*) FROM shards").fetchone()
    assert rows[0] == total and rows[1] == len(shards)


@pytest.mark.parametrize("leaves", [96, 16])
def test_full_size_configuration_properties(leaves):
    """BASELINE configs[1] at full size -- 256 concurrent games, 800 simulations per move, R24-320, 96 leaves per tree and
    pass (the reference's mcts.inference_batch_size, bench.py's default: 24 576 positions per forward) and 16 -- cut to 2
    searched plies per game: the invariants that do not depend on size.  Every searched ply spent
    exactly its simulation budget (800, or the reduced playout cap), pi is the normalised visit distribution over
    legal moves only, every played move is legal and was visited, planes/masks equal the oracle's on replay."""
    import bench
    from matrix0_amd.backend import M0Backend
    from matrix0_amd import engine as eng
    cfgd = {k: (dict(v) if isinstance(v, dict) else v) for k, v in bench.SELFPLAY_CFG.items()}
    cfgd["selfplay"] = dict(cfgd["selfplay"], max_game_len=2)
    be = M0Backend.from_state_dict(bench.R24_320, net_ref.random_state_dict(bench.R24_320, seed=0))
    cfg = eng.selfplay_cfg_from_dict(cfgd, concurrent_games=256, total_games=256, leaves_per_step=leaves,
                                     virtual_loss_active=True, record_games=True)
    e = eng.SelfplayEngine(be, cfg)
    games = []
    for _ in range(400):
        e.step(10)
        while (r := e.poll()) is not None:
            games.append(r)
        if not e.running():
            break
    st = e.stats()
    assert len(games) == 256 and st["games_finished"] == 256
    assert sorted(g["game_index"] for g in games) == list(range(256))
    sims = int(cfgd["selfplay"]["num_simulations"])
    assert sims == 800
    total_plies = sum(g["moves"] for g in games)
    assert st["plies"] == total_plies
    # every search spends its whole budget: 800 simulations, or the reduced playout cap on ~5% of the plies
    assert st["sims"] <= sims * total_plies and st["sims"] >= 0.9 * sims * total_plies
    for g in games:
        T = g["moves"]
        assert 1 <= T <= 2
        np.testing.assert_allclose(g["pi"].sum(axis=1), 1.0, atol=1e-4)
        assert np.all(g["pi"] >= 0) and np.all(g["pi"][g["legal_mask"] == 0] == 0)
        assert np.all(np.abs(g["z"]) <= 1.0)
        b = ch.Board()
        played = [ch.Move.from_uci(u) for u in g["played"]]
        n_open = len(played) - (T - (1 if g["resigned"] else 0))
        assert 0 <= n_open <= 12
        for i, m in enumerate(played):
            assert m in b.legal_moves, (i, m)
            if i >= n_open:
                t = i - n_open
                assert np.array_equal(g["s"][t], ch.encode_board(b))
                assert np.array_equal(g["legal_mask"][t].astype(bool), ch.get_legal_actions(b))
                assert g["pi"][t][ch.move_to_index(b, m)] > 0
            b.push(m)


def test_a_game_does_not_depend_on_the_games_it_shares_the_gpu_with():
    """The random streams are keyed by (seed, game index) and the network forward is bitwise batch-invariant, so game k is the
    same game whether it is searched next to 255 others (24 576 positions per forward) or next to 3: moves, visit
    distributions and outcomes of games 0..3 agree bit for bit between the two runs (what makes results independent of how
    the games are sharded over workers and GPUs)."""
    import bench
    from matrix0_amd.backend import M0Backend
    from matrix0_amd import engine as eng
    cfgd = {k: (dict(v) if isinstance(v, dict) else v) for k, v in bench.SELFPLAY_CFG.items()}
    cfgd["selfplay"] = dict(cfgd["selfplay"], max_game_len=2)
    be = M0Backend.from_state_dict(bench.R24_320, net_ref.random_state_dict(bench.R24_320, seed=0))
    runs = []
    for G in (256, 4):
        cfg = eng.selfplay_cfg_from_dict(cfgd, concurrent_games=G, total_games=G, leaves_per_step=96,
                                         virtual_loss_active=True, record_games=True)
        e = eng.SelfplayEngine(be, cfg)
        games = {}
        for _ in range(200):
            e.step(6)
            while (r := e.poll()) is not None:
                games[r["game_index"]] = r
            if not e.running():
                break
        assert len(games) == G
        runs.append(games)
        e.close()
    for k in range(4):
        a, b = runs[0][k], runs[1][k]
        assert a["played"] == b["played"], k
        assert np.array_equal(a["pi"], b["pi"]) and np.array_equal(a["z"], b["z"]) and np.array_equal(a["s"], b["s"]), k


def test_full_size_selfplay_is_reproducible():
    """Same seed, same build -> the same games, bit for bit, at the benchmark network size (64 games, 800 simulations,
    one searched ply): any race in the MFMA kernels' LDS / DMA choreography shows up here as a different visit count."""
    import bench
    from matrix0_amd.backend import M0Backend
    from matrix0_amd import engine as eng
    cfgd = {k: (dict(v) if isinstance(v, dict) else v) for k, v in bench.SELFPLAY_CFG.items()}
    cfgd["selfplay"] = dict(cfgd["selfplay"], max_game_len=1)
    be = M0Backend.from_state_dict(bench.R24_320, net_ref.random_state_dict(bench.R24_320, seed=0))
    runs = []
    for _ in range(2):
        cfg = eng.selfplay_cfg_from_dict(cfgd, concurrent_games=64, total_games=64, leaves_per_step=16,
                                         virtual_loss_active=True, record_games=True)
        e = eng.SelfplayEngine(be, cfg)
        games = {}
        for _ in range(200):
            e.step(10)
            while (r := e.poll()) is not None:
                games[r["game_index"]] = r
            if not e.running():
                break
        e.close()
        assert len(games) == 64
        runs.append(games)
    for i in range(64):
        a, b = runs[0][i], runs[1][i]
        assert a["played"] == b["played"], i
        assert np.array_equal(a["pi"], b["pi"]) and np.array_equal(a["search_values"], b["search_values"]), i


def test_baseline_config0_one_game_64_sims_full_size_net():
    """BASELINE configs[0] as a parity case: 1 self-play game, 64 sims/move, random-init R24-320 (the reference's
    CPU-runnable case), here on the GPU; the game is replayed through the oracle's rules and encoder."""
    from matrix0_amd.backend import M0Backend
    from matrix0_amd import engine as eng
    from matrix0_amd.weights import random_state_dict
    r24 = dict(planes=19, channels=320, blocks=24, attention_heads=20, policy_size=4672, norm="group", activation="silu",
               preact=True, policy_factor_rank=128, self_supervised=True, ssl_tasks=["piece", "threat", "pin", "fork", "control"])
    be = M0Backend.from_state_dict(r24, random_state_dict(r24, seed=0))
    cfg_d = {"seed": 1234, "mcts": dict(CFG["mcts"], inference_batch_size=16),
             "selfplay": dict(CFG["selfplay"], num_simulations=64, max_game_len=24, opening_random_plies=12)}
    e = eng.SelfplayEngine(be, eng.selfplay_cfg_from_dict(cfg_d, concurrent_games=1, total_games=1, ssl_in_forward=True,
                                                          ssl_targets=True))
    g = None
    while e.running():
        e.step(16)
        r = e.poll()
        if r is not None:
            g = r
    assert g is not None and 1 <= g["moves"] <= 24
    T = g["moves"]
    b = ch.Board()
    played = [ch.Move.from_uci(u) for u in g["played"]]
    n_open = len(played) - (T - (1 if g["resigned"] else 0))
    for i, mv in enumerate(played):
        assert mv in b.legal_moves
        if i >= n_open:
            t = i - n_open
            assert np.array_equal(g["s"][t], ch.encode_board(b))
            assert np.array_equal(g["legal_mask"][t].astype(bool), ch.get_legal_actions(b))
            from oracle import ssl_ref
            tg = ssl_ref.targets(ch.encode_board(b))
            for k in ("piece", "threat", "pin", "fork", "control"):
                assert np.array_equal(g["ssl"][k][t], tg[k].astype(np.float32)), k
        b.push(mv)
    st = e.stats()
    assert 64 * T * 0.9 <= st["sims"] <= 64 * T * 1.1 + 64      # playout cap +-5 %
    np.testing.assert_allclose(g["pi"].sum(axis=1), 1.0, atol=1e-4)


def test_tail_split_leaves_the_games_unchanged():
    """engine.tail_split: a pass of >= 2048 rows on the 320-wide network = a whole number of workgroup rounds on the main instance + the
    rest on a second instance over the same weights, concurrently.  The forward is bitwise batch invariant, so the games must be
    the same, bit for bit, with the split on and off -- and the split must actually have been taken.  Same for tail_split = "halves"
    (two half batches side by side)."""
    from matrix0_amd.backend import M0Backend
    from matrix0_amd import engine as eng
    net = dict(planes=19, channels=320, blocks=3, attention_heads=20, policy_size=4672, norm="group", activation="silu",
               preact=True, policy_factor_rank=128, self_supervised=True, ssl_tasks=["piece", "control"])
    be = M0Backend.from_state_dict(net, net_ref.random_state_dict(net, seed=31))
    cfgd = {k: (dict(v) if isinstance(v, dict) else v) for k, v in CFG.items()}
    cfgd["mcts"] = dict(cfgd["mcts"], inference_batch_size=96)
    cfgd["selfplay"] = dict(cfgd["selfplay"], num_simulations=200, max_game_len=3, opening_random_plies=1)

    def play(split, ssl):
        e = eng.SelfplayEngine(be, eng.selfplay_cfg_from_dict(cfgd, concurrent_games=48, total_games=48, tail_split=split,
                                                              ssl_in_forward=ssl))
        games = _play_all(e)
        st = e.stats()
        e.close()
        return games, st

    for ssl in (False, True):
        off, st_off = play(False, ssl)
        assert st_off["rows_tail"] == 0
        # True: main part + tail (< 1024 rows); "halves": two halves side by side (passes of >= 4096 rows: 48 games x 96 leaves)
        for mode in (True, "halves"):
            on, st_on = play(mode, ssl)
            assert sorted(off) == sorted(on) == list(range(48))
            for i in range(48):
                assert off[i]["played"] == on[i]["played"], (mode, i)
                for k in ("pi", "z", "s", "legal_mask", "search_values"):
                    assert np.array_equal(off[i][k], on[i][k]), (mode, i, k)
            assert st_on["rows_tail"] > 0 and st_on["evals"] == st_off["evals"], mode
            if mode == "halves":
                assert st_on["rows_tail"] > 0.25 * st_on["evals"], (st_on["rows_tail"], st_on["evals"])
    be.close()
