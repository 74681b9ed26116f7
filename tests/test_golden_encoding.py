"""azchess/encoding.py pinned by outputs of the REAL reference file (tests/golden/ref_encoding.npz, written by
tools/gen_golden_mcts.py: encode_board / move_to_index / MoveEncoder.get_legal_actions / decode_move of the reference run on
its own 10 000 tactical FENs + edge cases).  CPU: the oracle and the host build of the product's bitboard core; GPU: the
device kernels through the C-ABI, compared with the golden file directly.  Bit-exact."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import chess_py as ch
from tests.golden_ref import load_npz, planes_from_bits, uci

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gold():
    z = load_npz("ref_encoding.npz")
    g = {k: z[k] for k in z.files}
    g["fens"] = [str(f) for f in g["fens"]]
    g["off"] = np.concatenate([[0], np.cumsum(g["nlegal"].astype(np.int64))])
    return g


def test_oracle_encoding_matches_reference_outputs(gold):
    fens = gold["fens"]
    assert len(fens) >= 10000
    for i in range(0, len(fens), 3):
        b = ch.Board(fens[i])
        assert np.array_equal(ch.encode_board(b), planes_from_bits(gold["plane_bits"][i], gold["counters"][i])), fens[i]
        moves, idxs = ch.legal_moves_with_indices(b)
        lo, hi = gold["off"][i], gold["off"][i + 1]
        assert [m.uci() for m in moves] == [uci(int(c)) for c in gold["moves"][lo:hi]], fens[i]
        assert idxs == gold["idx"][lo:hi].tolist(), fens[i]
        mask = ch.get_legal_actions(b)
        assert int(mask.sum()) == hi - lo and all(mask[j] for j in idxs)
    for i, j, code in gold["decode_probe"]:
        assert ch.decode_move(ch.Board(fens[int(i)]), int(j)).uci() == uci(int(code)), (fens[int(i)], int(j))


def test_product_bitboard_core_matches_reference_outputs(gold):
    """csrc/chess_core.h compiled for the host (tests/host_shim): the code the device kernels are built from."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "host_shim")])
    shim = C.CDLL(os.path.join(HERE, "_build", "libchess_shim.so"))
    mv = (C.c_int32 * 256)()
    idx = (C.c_int32 * 256)()
    buf = np.zeros((19, 8, 8), np.float32)
    fens = gold["fens"]
    for i in range(len(fens)):
        n = shim.hc_legal(fens[i].encode(), mv, idx)
        lo, hi = gold["off"][i], gold["off"][i + 1]
        assert n == hi - lo, fens[i]
        got = [(mv[k] & 255) | (((mv[k] >> 8) & 255) << 6) | ((mv[k] >> 16) << 12) for k in range(n)]
        assert got == gold["moves"][lo:hi].tolist(), fens[i]
        assert [idx[k] for k in range(n)] == gold["idx"][lo:hi].tolist(), fens[i]
        if i % 4 == 0:
            assert shim.hc_encode(fens[i].encode(), buf.ctypes.data_as(C.c_void_p)) == 0
            assert np.array_equal(buf, planes_from_bits(gold["plane_bits"][i], gold["counters"][i])), fens[i]


def test_permutations_match_reference(gold):
    from matrix0_amd import encoding as enc
    assert np.array_equal(enc.build_horizontal_flip_permutation(), gold["hflip"])
    assert np.array_equal(enc.build_rotate180_permutation(), gold["rot180"])


@pytest.mark.gpu
def test_device_encoding_matches_reference_outputs(gold):
    """Every position of the golden file through m0_encode_fens: planes, legal mask, move ORDER, policy indices."""
    from matrix0_amd import encoding as enc
    fens = gold["fens"]
    planes, mask, moves = enc.encode_fens(fens)
    for i in range(len(fens)):
        lo, hi = gold["off"][i], gold["off"][i + 1]
        assert moves[i][0] == [uci(int(c)) for c in gold["moves"][lo:hi]], fens[i]
        assert moves[i][1] == gold["idx"][lo:hi].tolist(), fens[i]
        assert np.array_equal(planes[i], planes_from_bits(gold["plane_bits"][i], gold["counters"][i])), fens[i]
        want = np.zeros(4672, bool)
        want[gold["idx"][lo:hi]] = True
        assert np.array_equal(mask[i], want), fens[i]
    me = enc.MoveEncoder()
    for i, j, code in gold["decode_probe"]:
        assert me.decode_move(fens[int(i)], int(j)) == uci(int(code)), (fens[int(i)], int(j))
    # move_to_index one move at a time (encoding.py:114-150) on a sample, and the ValueError of an illegal move
    for i in range(0, len(fens), 97):
        lo, hi = gold["off"][i], gold["off"][i + 1]
        for c, want_idx in zip(gold["moves"][lo:hi], gold["idx"][lo:hi]):
            assert enc.move_to_index(fens[i], uci(int(c))) == int(want_idx)
    with pytest.raises(ValueError):
        enc.move_to_index(ch.START_FEN, "e2e5")


@pytest.mark.gpu
def test_network_input_of_the_search_matches_reference_outputs(gold):
    """The network input of the TIMED path: csrc/tree.hip::encode_nhwc, the device function select_kernel calls for every
    leaf (fp16, NHWC [64 squares][32 channels]), run on every position of the golden file through m0_encode_fens_nhwc and
    compared with fp16(encode_board of the reference) bit for bit -- channels 19..31 must be zero."""
    from matrix0_amd import engine as eng
    fens = gold["fens"]
    got = eng.encode_fens_nhwc(fens)
    assert got.shape == (len(fens), 64, 32) and got.dtype == np.float16
    want = eng.planes_to_nhwc(np.stack([planes_from_bits(gold["plane_bits"][i], gold["counters"][i]) for i in range(len(fens))]))
    bad = np.nonzero((got.view(np.uint16) != want.view(np.uint16)).any(axis=(1, 2)))[0]
    assert bad.size == 0, fens[int(bad[0])]
    assert not got[:, :, 19:].any()
