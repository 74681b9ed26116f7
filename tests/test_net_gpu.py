"""GPU parity: HIP network forward (through the C-ABI / infer_np seam) vs the fp32 oracle
and vs the reference golden vectors.

Tolerances (fp16 MFMA operands, fp32 accumulate, fp16 activations in HBM vs the fp32 reference), set from what the
hardware delivers (tools/net_margins.py on an MI355X, round 2: max|dlogit| 7.2e-4 on logits spanning +-0.43, |dv| 5.4e-4,
relative L2 1.5e-3, SSL maps 4e-3 on values spanning +-2, top-1 over the legal moves identical on 256/256 positions):
    max|dlogit| <= 5e-3    |dv| <= 2e-3    ||dp|| / ||p|| <= 5e-3    SSL <= 1e-2    KL(softmax) <= 1e-5
    top-1 over legal moves agrees on >= 99 % of real positions
Every comparison appends its observed maxima to gpurun_out/net_parity.jsonl so the record shows the margin."""
import json
import os
import numpy as np
import pytest
import torch

from oracle import net_ref
from tests.golden_util import SEEDED_NET_CASES, load_net_golden, load_seeded_net_golden

pytestmark = pytest.mark.gpu

LOGIT_TOL = 5e-3
VALUE_TOL = 2e-3
REL_L2_TOL = 5e-3
SSL_TOL = 1e-2
KL_TOL = 1e-5
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check(tag, p, v, p_ref, v_ref, ssl=None, ssl_ref=None):
    """All network tolerances in one place; observed values go to stdout and gpurun_out/net_parity.jsonl."""
    p_ref = np.asarray(p_ref, np.float32); v_ref = np.asarray(v_ref, np.float32)
    obs = {"case": tag, "dlogit": float(np.abs(p - p_ref).max()), "logit_absmax": float(np.abs(p_ref).max()),
           "dv": float(np.abs(v - v_ref).max()), "rel_l2": float(np.linalg.norm(p - p_ref) / max(1e-30, np.linalg.norm(p_ref))),
           "kl": _kl(p_ref, p)}
    if ssl_ref:
        obs["ssl"] = {t: float(np.abs(ssl[t] - np.asarray(r)).max()) for t, r in ssl_ref.items()}
    print("net parity", json.dumps(obs))
    try:
        os.makedirs(os.path.join(_ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(_ROOT, "gpurun_out", "net_parity.jsonl"), "a") as f:
            f.write(json.dumps(obs) + "\n")
    except OSError:
        pass
    assert obs["dlogit"] <= LOGIT_TOL, obs
    assert obs["dv"] <= VALUE_TOL, obs
    assert obs["rel_l2"] <= REL_L2_TOL, obs
    assert obs["kl"] <= KL_TOL, obs
    for t, e in obs.get("ssl", {}).items():
        assert e <= SSL_TOL, (t, obs)
    return obs


def _kl(p_ref, p):
    a = torch.log_softmax(torch.from_numpy(p_ref), -1)
    b = torch.log_softmax(torch.from_numpy(p), -1)
    return float((a.exp() * (a - b)).sum(-1).max())


@pytest.mark.parametrize("name", ["gn_silu_preact", "gn_dense_leaky", "stride2"])
def test_hip_net_matches_reference_golden(name):
    from matrix0_amd.backend import M0Backend
    cfg, sd, x, p_ref, v_ref, ssl_ref = load_net_golden(name)
    be = M0Backend.from_state_dict(cfg, sd)
    if ssl_ref:
        p, v, ssl = be.infer_np_ssl(x)
    else:
        p, v = be.infer_np(x)
        ssl = {}
    assert p.shape == p_ref.shape and v.shape == v_ref.shape
    _check(f"golden:{name}", p, v, p_ref, v_ref, ssl, ssl_ref)
    assert be.param_count() == sum(int(np.prod(t.shape)) for t in sd.values())


@pytest.mark.parametrize("name", SEEDED_NET_CASES)
def test_hip_320_wide_kernels_match_the_reference_module(name):
    """The kernels that are > 90 % of the GPU time (conv_zs_kernel with the GroupNorm epilogue and the fused block tail, the fused
    attention block with and without the next block's pre-activated output, the big-tile 1x1 convs) against numbers the
    REFERENCE module itself computed at C = 320, H = 20 (resnet.py:27-84, 137-190, 656-760) -- not through the oracle."""
    from matrix0_amd.backend import M0Backend
    cfg, sd, x, p_ref, v_ref, ssl_ref = load_seeded_net_golden(name)
    be = M0Backend.from_state_dict(cfg, sd)
    p, v, ssl = be.infer_np_ssl(x)
    _check(f"golden:{name}", p, v, p_ref, v_ref, ssl, ssl_ref)
    # a batch that is not a multiple of the 4-board tile / the 2-board attention workgroup
    p3, v3 = be.infer_np(x[:3])
    assert np.array_equal(p3, p[:3]) and np.array_equal(v3, v[:3])
    be.close()


def test_unsupported_config_fails_loudly():
    from matrix0_amd.backend import M0Backend
    cfg, sd, *_ = load_net_golden("bn_relu_postact")
    with pytest.raises(RuntimeError, match="unsupported network config"):
        M0Backend(cfg)


def _r24_cfg():
    return dict(planes=19, channels=320, blocks=24, attention_heads=20, policy_size=4672, norm="group",
                activation="silu", preact=True, policy_factor_rank=128, self_supervised=True,
                ssl_tasks=["piece", "threat", "pin", "fork", "control"])


def test_r24_320_vs_oracle():
    """Full-size benchmark network, random-init weights, vs the fp32 CPU oracle."""
    from matrix0_amd.backend import M0Backend
    cfg = _r24_cfg()
    sd = net_ref.random_state_dict(cfg, seed=0)
    be = M0Backend.from_state_dict(cfg, sd)
    assert be.param_count() == 57_562_210
    assert abs(be.flops_per_position(False) / 6.3085e9 - 1) < 1e-3
    assert abs(be.flops_per_position(True) / 6.3417e9 - 1) < 1e-3
    g = torch.Generator().manual_seed(5)
    B = 6   # not a multiple of the 4-board tile: exercises padding
    x = torch.zeros(B, 19, 8, 8)
    x[:, :12] = (torch.rand(B, 12, 8, 8, generator=g) < 0.08).float()
    x[:, 12:17] = (torch.rand(B, 5, 1, 1, generator=g) < 0.5).float()
    x[:, 17:] = torch.rand(B, 2, 1, 1, generator=g)
    p_ref, v_ref, ssl_ref = net_ref.forward(sd, cfg, x, return_ssl=True)
    p, v, ssl = be.infer_np_ssl(x.numpy())
    _check("r24_320:random_planes", p, v, p_ref.numpy(), v_ref.numpy(), ssl, {t: r.numpy() for t, r in ssl_ref.items()})
    # single position + repeated call determinism
    p1, v1 = be.infer_np(x[0].numpy())
    assert np.array_equal(p1[0], p[0]) and v1[0] == v[0]


def test_fused_block_tail_matches_split_kernels(monkeypatch):
    """conv2 with the residual-block tail in its epilogue (conv_tail.h) vs conv2 + se_gate + ew_board as three
    kernels: the same arithmetic with one fp16 rounding moved, on a ragged batch (70 boards = 17.5 tiles), and both
    against the fp32 oracle.  Also covers a network without squeeze-excite (gate == 1) and relu."""
    from matrix0_amd.backend import M0Backend
    for extra in ({}, {"se": False, "activation": "relu"}):
        cfg = dict(_r24_cfg(), blocks=4, **extra)
        sd = net_ref.random_state_dict(cfg, seed=3)
        be = M0Backend.from_state_dict(cfg, sd)
        g = torch.Generator().manual_seed(11)
        B = 70
        x = torch.zeros(B, 19, 8, 8)
        x[:, :12] = (torch.rand(B, 12, 8, 8, generator=g) < 0.08).float()
        x[:, 12:17] = (torch.rand(B, 5, 1, 1, generator=g) < 0.5).float()
        x[:, 17:] = torch.rand(B, 2, 1, 1, generator=g)
        monkeypatch.setenv("M0_FUSE_TAIL", "0")
        be_split = M0Backend.from_state_dict(cfg, sd)         # the kernel switches are read once, when a network is created
        monkeypatch.delenv("M0_FUSE_TAIL")
        p_split, v_split = be_split.infer_np(x.numpy())
        be_split.close()
        p_fused, v_fused = be.infer_np(x.numpy())
        assert np.abs(p_fused - p_split).max() <= 2e-3
        assert np.abs(v_fused - v_split).max() <= 2e-3
        p_ref, v_ref = net_ref.forward(sd, cfg, x, return_ssl=False)[:2]
        for tag, p, v in (("split", p_split, v_split), ("fused", p_fused, v_fused)):
            _check(f"tail_{tag}:{sorted(extra.items())}", p, v, p_ref.numpy(), v_ref.numpy())


def test_conv_kernel_variants_agree(monkeypatch):
    """The two 3x3 conv kernels of the library compute the same convolution: conv_zs_kernel (default: the MFMA tiles that only
    multiply zero padding are not issued, squeeze-excite FCs on the matrix cores) and conv_pp16_kernel (M0_CONV_ZS=0 when the
    network is created; also the fallback for squeeze-excite layers wider than 96 hidden units).  Variants: silu + SE
    (80 hidden units), relu + SE, a small SE ratio (hidden = 8: one FC1 tile, one FC2 k-step), and a ragged batch."""
    from matrix0_amd.backend import M0Backend
    for extra in ({}, {"activation": "relu"}, {"se_ratio": 0.025}):
        cfg = dict(_r24_cfg(), blocks=3, **extra)
        sd = net_ref.random_state_dict(cfg, seed=21)
        be = M0Backend.from_state_dict(cfg, sd)
        g = torch.Generator().manual_seed(31)
        B = 41
        x = torch.zeros(B, 19, 8, 8)
        x[:, :12] = (torch.rand(B, 12, 8, 8, generator=g) < 0.08).float()
        x[:, 12:17] = (torch.rand(B, 5, 1, 1, generator=g) < 0.5).float()
        x[:, 17:] = torch.rand(B, 2, 1, 1, generator=g)
        # the kernel switches are read once, when a network is created: one backend per variant
        monkeypatch.setenv("M0_CONV_ZS", "0")
        be16 = M0Backend.from_state_dict(cfg, sd)
        monkeypatch.delenv("M0_CONV_ZS")
        p16, v16 = be16.infer_np(x.numpy())
        be16.close()
        pzs, vzs = be.infer_np(x.numpy())
        assert np.abs(pzs - p16).max() <= 2e-3, extra
        assert np.abs(vzs - v16).max() <= 2e-3, extra
        p_ref, v_ref = net_ref.forward(sd, cfg, x, return_ssl=False)[:2]
        for tag, p, v in (("pp16", p16, v16), ("zs", pzs, vzs)):
            _check(f"conv_{tag}:{sorted(extra.items())}", p, v, p_ref.numpy(), v_ref.numpy())
        # both kernels' tails handle the four boards of a tile in one thread: a board's result must not depend on which of the
        # four it is (the squeeze-excite sums are explicit fused multiply-adds for that reason, round 4) -- rotate the batch
        perm = np.roll(np.arange(B), 1)
        monkeypatch.setenv("M0_CONV_ZS", "0")
        be16 = M0Backend.from_state_dict(cfg, sd)
        monkeypatch.delenv("M0_CONV_ZS")
        pr, vr = be16.infer_np(x.numpy()[perm])
        be16.close()
        assert np.array_equal(pr, p16[perm]) and np.array_equal(vr, v16[perm]), extra
        pr, vr = be.infer_np(x.numpy()[perm])
        assert np.array_equal(pr, pzs[perm]) and np.array_equal(vr, vzs[perm]), extra


def test_fused_attention_block_matches_split_kernels(monkeypatch):
    """The whole attention block in one kernel (attn_block.hip: qkv GEMM, scores/softmax/PV, proj GEMM, residual,
    LayerNorm, the next block's GroupNorm + activation) vs qkv + attn_core + proj + ew_board as four kernels, on a ragged
    batch, and both against the fp32 oracle.  Variants: the default (relative bias, mix 0.2), masked only / unmasked
    only branches without relative bias, relu, an attention block as the LAST tower layer (no second output), and the
    zero-padded 288-channel trunk (18 real heads of 20)."""
    from matrix0_amd.backend import M0Backend
    variants = (
        dict(blocks=6),
        dict(blocks=6, attention_unmasked_mix=1.0, attention_relbias=False, activation="relu"),
        dict(blocks=3, attention_unmasked_mix=0.0),
        dict(blocks=6, channels=288, attention_heads=18),
    )
    for extra in variants:
        cfg = dict(_r24_cfg(), **extra)
        sd = net_ref.random_state_dict(cfg, seed=5)
        be = M0Backend.from_state_dict(cfg, sd)
        g = torch.Generator().manual_seed(12)
        B = 37
        x = torch.zeros(B, 19, 8, 8)
        x[:, :12] = (torch.rand(B, 12, 8, 8, generator=g) < 0.08).float()
        x[:, 12:17] = (torch.rand(B, 5, 1, 1, generator=g) < 0.5).float()
        x[:, 17:] = torch.rand(B, 2, 1, 1, generator=g)
        monkeypatch.setenv("M0_FUSE_ATTN", "0")
        be_split = M0Backend.from_state_dict(cfg, sd)         # switches are read once, at creation
        monkeypatch.delenv("M0_FUSE_ATTN")
        p_split, v_split = be_split.infer_np(x.numpy())
        be_split.close()
        p_fused, v_fused = be.infer_np(x.numpy())
        assert np.abs(p_fused - p_split).max() <= 2e-3, extra
        assert np.abs(v_fused - v_split).max() <= 2e-3, extra
        p_ref, v_ref = net_ref.forward(sd, cfg, x, return_ssl=False)[:2]
        for tag, p, v in (("split", p_split, v_split), ("fused", p_fused, v_fused)):
            _check(f"attn_{tag}:{sorted(extra.items())}", p, v, p_ref.numpy(), v_ref.numpy())
        be.close()


def test_shipped_config_288x22_zero_padded_trunk():
    """The reference's shipped config.yaml network (288 channels x 22 blocks, 18 heads, rank-160 policy, attention
    stride 2, leaky value head): the engine zero-pads its trunk to 320 channels so it runs on the MFMA big-tile kernels;
    padded channels must stay exactly zero through GroupNorm / squeeze-excite / attention / LayerNorm."""
    from matrix0_amd.backend import M0Backend
    cfg = dict(planes=19, channels=288, blocks=22, attention=True, attention_heads=18, policy_size=4672, norm="group",
               activation="silu", value_activation="leaky_relu", preact=True, policy_factor_rank=160,
               infer_attention_stride=2, self_supervised=True, ssl_tasks=["piece", "threat", "pin", "fork", "control"])
    sd = net_ref.random_state_dict(cfg, seed=0)
    be = M0Backend.from_state_dict(cfg, sd)
    assert be.param_count() == 44_193_314
    g = torch.Generator().manual_seed(5)
    B = 5
    x = torch.zeros(B, 19, 8, 8)
    x[:, :12] = (torch.rand(B, 12, 8, 8, generator=g) < 0.08).float()
    x[:, 12:17] = (torch.rand(B, 5, 1, 1, generator=g) < 0.5).float()
    x[:, 17:] = torch.rand(B, 2, 1, 1, generator=g)
    p_ref, v_ref, ssl_ref = net_ref.forward(sd, cfg, x, return_ssl=True)
    p, v, ssl = be.infer_np_ssl(x.numpy())
    _check("shipped_288x22", p, v, p_ref.numpy(), v_ref.numpy(), ssl, {t: r.numpy() for t, r in ssl_ref.items()})


def test_r24_320_top1_over_legal_moves_on_real_positions():
    """identical top-1 prior over the legal moves on >= 99 % of positions -- 256 positions of the
    reference's tactical set, R24-320 with the benchmark's synthetic weights, vs the fp32 oracle."""
    import gzip
    from matrix0_amd.backend import M0Backend
    from matrix0_amd.weights import random_state_dict
    from oracle import chess_py as ch
    cfg = _r24_cfg()
    sd = random_state_dict(cfg, seed=0, varied=True)
    be = M0Backend.from_state_dict(cfg, sd)
    rows = json.load(gzip.open(os.path.join(_ROOT, "tests", "golden", "tactical_legal_counts.json.gz"), "rt"))
    boards = [ch.Board(r[0]) for r in rows[::40][:256]]
    x = np.stack([ch.encode_board(b) for b in boards])
    masks = np.stack([ch.get_legal_actions(b) for b in boards])
    p, v, ssl = be.infer_np_ssl(x)
    with torch.no_grad():
        p_ref, v_ref, ssl_ref = net_ref.forward(sd, cfg, torch.from_numpy(x), return_ssl=True)
    _check("r24_320:real_positions", p, v, p_ref.numpy(), v_ref.numpy(), ssl, {t: r.numpy() for t, r in ssl_ref.items()})
    a = np.where(masks, p, -np.inf).argmax(1)
    b = np.where(masks, p_ref.numpy(), -np.inf).argmax(1)
    agree = float((a == b).mean())
    print("top-1 over legal moves agreement:", agree)
    assert agree >= 0.99


def test_from_checkpoint_reads_the_reference_checkpoint_layout(tmp_path):
    """M0Backend.from_checkpoint on .pt files in the reference's dict layout (create_v2_checkpoint.py:64-77): the weights are
    taken from model_ema, else model, else model_state_dict, else the raw dict (selfplay/internal.py:172-174,
    orchestrator.py:373-388); the duplicate alias keys a reference state_dict carries (ssl_head.* / ssl_piece_head.* =
    ssl_heads.piece.*, resnet.py:359-444) and unknown keys are ignored; a missing tensor is an error here (the reference
    re-initialises it silently, resnet.py:1418-1440)."""
    from matrix0_amd.backend import M0Backend
    from matrix0_amd.weights import random_state_dict
    cfg = dict(planes=19, channels=64, blocks=3, attention_heads=4, policy_size=4672, norm="group", activation="silu",
               preact=True, policy_factor_rank=32, self_supervised=True, ssl_tasks=["piece", "control"])
    good = random_state_dict(cfg, seed=21, varied=True)
    other = random_state_dict(cfg, seed=22, varied=True)
    with_alias = dict(good)
    for k, t in good.items():
        if k.startswith("ssl_heads.piece."):
            with_alias["ssl_head." + k[len("ssl_heads.piece."):]] = t.clone()
            with_alias["ssl_piece_head." + k[len("ssl_heads.piece."):]] = t.clone()
    with_alias["optimizer_leftover.weight"] = torch.zeros(3)
    x = np.zeros((2, 19, 8, 8), np.float32); x[:, 12] = 1.0; x[0, 0, 6, :] = 1.0; x[1, 5, 7, 4] = 1.0
    want_p, want_v = M0Backend.from_state_dict(cfg, good).infer_np(x)
    layouts = {
        "ema_first": {"model_ema": with_alias, "model": other, "model_state_dict": other, "optimizer": {"state": {}}, "global_step": 7},
        "model_then": {"model": with_alias, "model_state_dict": other, "version": "v2"},
        "model_state_dict": {"model_state_dict": with_alias, "model_config": cfg},
        "raw": with_alias,
    }
    for name, blob in layouts.items():
        path = str(tmp_path / f"{name}.pt")
        torch.save(blob, path)
        be = M0Backend.from_checkpoint(cfg, path)
        p, v = be.infer_np(x)
        assert np.array_equal(p, want_p) and np.array_equal(v, want_v), name
        be.close()
    broken = {k: t for k, t in good.items() if k != "tower.1.conv2.weight"}
    torch.save({"model": broken}, str(tmp_path / "broken.pt"))
    with pytest.raises((RuntimeError, ValueError)):
        M0Backend.from_checkpoint(cfg, str(tmp_path / "broken.pt"))


def test_full_size_batch_invariance_and_determinism():
    """Size-independent properties at the benchmark batch size: a board's outputs do not depend on which other boards
    share its launch (4-board tiles, per-board statistics), nor on the run -- 4096 boards in one call, the same boards
    in ragged chunks, and a repeat must agree bit for bit."""
    from matrix0_amd.backend import M0Backend
    cfg = _r24_cfg()
    be = M0Backend.from_state_dict(cfg, net_ref.random_state_dict(cfg, seed=0))
    rng = np.random.default_rng(7)
    B = 4096
    x = np.zeros((B, 19, 8, 8), np.float32)
    x[:, :12] = (rng.random((B, 12, 8, 8)) < 0.08).astype(np.float32)
    x[:, 12:17] = (rng.random((B, 5, 1, 1)) < 0.5).astype(np.float32)
    x[:, 17:] = rng.random((B, 2, 1, 1)).astype(np.float32)
    p_all, v_all = be.infer_np(x)
    assert np.isfinite(p_all).all() and np.isfinite(v_all).all() and np.abs(v_all).max() <= 1.0
    p_rep, v_rep = be.infer_np(x)
    assert np.array_equal(p_all, p_rep) and np.array_equal(v_all, v_rep)
    start = 0
    for n in (1, 3, 254, 1000, 2838):                  # ragged chunk sizes summing to 4096
        p, v = be.infer_np(x[start:start + n])
        assert np.array_equal(p, p_all[start:start + n]) and np.array_equal(v, v_all[start:start + n]), (start, n)
        start += n
    assert start == B


@pytest.mark.parametrize("kw", [{}, {"channels": 64, "attention_heads": 4}, {"blocks": 3}, {"se": False}],
                         ids=["32ch", "64ch", "3blocks_attention", "no_se"])
def test_narrow_networks_do_not_depend_on_the_position_in_the_batch(kw):
    """The narrow-trunk path (small-tile convs, `se_gate_kernel`, `attn_core_kernel`, `ew_board_kernel`: every network that is not
    320 wide, i.e. every small test network): a board's logits and value must be the same bits wherever it sits in a batch and
    whatever else is in it.  Round 4: `se_gate_kernel` handled 8 boards per thread and hipcc fused the multiply-adds of some of
    the eight only -- the gate depended on the position modulo 8 in the last bit, root values of concurrent games differed from run
    to run (the games race for batch rows), and once in a few dozen runs a move flipped."""
    from matrix0_amd.backend import M0Backend
    cfg = dict(planes=19, channels=32, blocks=2, attention_heads=2, policy_size=4672, norm="group", activation="silu",
               preact=True, policy_factor_rank=16, self_supervised=False)
    cfg.update(kw)
    be = M0Backend.from_state_dict(cfg, net_ref.random_state_dict(cfg, seed=1))
    rng = np.random.default_rng(3)
    B = 290                                                  # two 256-row FC tiles, a partial 4-board tile, a partial group of 8
    x = np.zeros((B, 19, 8, 8), np.float32)
    x[:, :12] = (rng.random((B, 12, 8, 8)) < 0.08).astype(np.float32)
    x[:, 12:17] = (rng.random((B, 5, 1, 1)) < 0.5).astype(np.float32)
    x[:, 17:] = rng.random((B, 2, 1, 1)).astype(np.float32)
    p0, v0 = be.infer_np(x)
    for i in (0, 5, 6, 7, 255, 256, 289):                    # alone in a batch of one
        p, v = be.infer_np(x[i:i + 1])
        assert np.array_equal(p[0], p0[i]) and v[0] == v0[i], i
    for perm in (np.arange(B)[::-1], np.roll(np.arange(B), 1), np.roll(np.arange(B), 6), rng.permutation(B)):
        p, v = be.infer_np(x[perm])
        assert np.array_equal(p, p0[perm]) and np.array_equal(v, v0[perm])
    p, v = be.infer_np(x[:143])
    assert np.array_equal(p, p0[:143]) and np.array_equal(v, v0[:143])
    be.close()


def test_bench_size_forward_is_batch_invariant():
    """The forward of a self-play pass at the bench configuration takes 256 games x 96 leaves (+ 256 re-evaluated roots): 24 832
    boards in one launch sequence, 24 rounds of workgroups per CU in the conv kernels.  Boards 0..63, the last 61 boards and five 70-board
    slices of that batch must come out bit for bit as they do in a small call (logits of 4672 floats per board: only these slices
    are copied back and compared)."""
    from matrix0_amd.backend import M0Backend
    cfg = _r24_cfg()
    be = M0Backend.from_state_dict(cfg, net_ref.random_state_dict(cfg, seed=0))
    rng = np.random.default_rng(17)
    B = 24832
    x = np.zeros((B, 19, 8, 8), np.float32)
    x[:, :12] = (rng.random((B, 12, 8, 8)) < 0.08).astype(np.float32)
    x[:, 12:17] = (rng.random((B, 5, 1, 1)) < 0.5).astype(np.float32)
    x[:, 17:] = rng.random((B, 2, 1, 1)).astype(np.float32)
    p_all, v_all = be.infer_np(x)
    assert np.isfinite(p_all).all() and np.isfinite(v_all).all() and np.abs(v_all).max() <= 1.0
    p0, v0 = be.infer_np(x[:64])
    assert np.array_equal(p0, p_all[:64]) and np.array_equal(v0, v_all[:64])
    p1, v1 = be.infer_np(x[B - 61:])
    assert np.array_equal(p1, p_all[B - 61:]) and np.array_equal(v1, v_all[B - 61:])
    for start in (4093, 8190, 12289, 16411, 20477):          # slices across tile / launch-round boundaries
        p, v = be.infer_np(x[start:start + 70])
        assert np.array_equal(p, p_all[start:start + 70]) and np.array_equal(v, v_all[start:start + 70]), start
    # race screen: the whole batch three more times, every board bit for bit (a wave that reads an operand before its wait has
    # covered it shows up as a handful of boards that differ from run to run: attn_block.hip's bias loads, round 3)
    for _ in range(3):
        pr, vr = be.infer_np(x)
        assert np.array_equal(vr, v_all)
        assert np.array_equal(pr, p_all)
        del pr
    # the 5 SSL heads in the same forward (BASELINE configs[3] at this batch size): same property
    ps, vs, ssl_all = be.infer_np_ssl(x)
    assert np.array_equal(ps, p_all) and np.array_equal(vs, v_all)
    for start in (0, 12289, B - 70):
        _, _, ssl = be.infer_np_ssl(x[start:start + 70])
        for task, t in ssl.items():
            assert np.array_equal(t, ssl_all[task][start:start + 70]), (task, start)


def test_workspace_regrowth_keeps_results():
    """A backend that is called with growing batches re-allocates (and clears) its workspace: the clears run on the NULL stream,
    the forward on a non-blocking one, so the re-allocation must be complete before the first kernel writes the new buffers
    (a missing synchronisation there once gave garbage, differing from run to run, for the first call after every regrowth)."""
    from matrix0_amd.backend import M0Backend
    cfg = dict(_r24_cfg(), blocks=6)
    be = M0Backend.from_state_dict(cfg, net_ref.random_state_dict(cfg, seed=2))
    rng = np.random.default_rng(23)
    B = 12288
    x = np.zeros((B, 19, 8, 8), np.float32)
    x[:, :12] = (rng.random((B, 12, 8, 8)) < 0.08).astype(np.float32)
    x[:, 12:17] = (rng.random((B, 5, 1, 1)) < 0.5).astype(np.float32)
    x[:, 17:] = rng.random((B, 2, 1, 1)).astype(np.float32)
    p0, v0 = be.infer_np(x[:64])
    for n in (700, 5000, 9000, B):                       # every call regrows the workspace
        p, v = be.infer_np(x[:n])
        assert np.array_equal(p[:64], p0) and np.array_equal(v[:64], v0), n
        p2, v2 = be.infer_np(x[:n])
        assert np.array_equal(p, p2) and np.array_equal(v, v2), n


def test_infer_np_rejects_bad_shape_and_nan():
    from matrix0_amd.backend import M0Backend
    cfg, sd, x, *_ = load_net_golden("gn_silu_preact")
    be = M0Backend.from_state_dict(cfg, sd)
    with pytest.raises(ValueError):
        be.infer_np(np.zeros((2, 18, 8, 8), np.float32))
    bad = x.copy()
    bad[0, 0, 0, 0] = np.nan
    with pytest.raises(ValueError):
        be.infer_np(bad)


def test_cabi_weight_broadcast_single_rank_and_guards():
    """m0_dist_* / m0_net_broadcast_weights (the RCCL broadcast without torch, csrc/capi_dist.hip) with a one-rank communicator:
    the collective runs, the network computes what it computed before, and a network that is not finalized is refused.
    (The driver's multi-GPU node exercises more ranks; ranks = GPUs, and this box has one.)"""
    from matrix0_amd.backend import M0Backend
    from matrix0_amd.dist import CAbiDist
    from matrix0_amd import _lib
    cfg, sd, x, *_ = load_net_golden("gn_silu_preact")
    be = M0Backend.from_state_dict(cfg, sd)
    p0, v0 = be.infer_np(x)
    d = CAbiDist(0, 1, CAbiDist.unique_id(), 0)
    d.broadcast_weights(be, root=0)
    p1, v1 = be.infer_np(x)
    assert np.array_equal(p0, p1) and np.array_equal(v0, v1)
    raw = M0Backend(cfg)                      # created, weights not loaded / finalized
    with pytest.raises(RuntimeError, match="finalized"):
        d.broadcast_weights(raw, root=0)
    with pytest.raises(ValueError):
        d.broadcast_weights(be, root=3)       # no such rank
    raw.close(); d.close(); be.close()
