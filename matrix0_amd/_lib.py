"""ctypes loader for libm0engine.so (the C-ABI in include/m0_engine.h).

The product path fails loudly if the HIP library is missing: there is no CPU
fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libm0engine.so")

M0_OK = 0
M0_ERR_INVALID = -1
M0_ERR_UNSUPPORTED = -2
M0_ERR_HIP = -3
M0_ERR_STATE = -4
M0_ERR_NONFINITE = -5
POLICY_SIZE = 4672

ACT = {"relu": 1, "silu": 2, "leaky_relu": 3}
SSL_BITS = {"piece": 1, "threat": 2, "pin": 4, "fork": 8, "control": 16}
SSL_CH = {"piece": 13, "threat": 1, "pin": 1, "fork": 1, "control": 3}
SSL_ORDER = ["piece", "threat", "pin", "fork", "control"]


class NetCfg(C.Structure):
    _fields_ = [
        ("planes", C.c_int), ("channels", C.c_int), ("blocks", C.c_int), ("attention", C.c_int),
        ("attention_heads", C.c_int), ("attention_every_k", C.c_int), ("attention_relbias", C.c_int),
        ("attention_unmasked_mix", C.c_float), ("se", C.c_int), ("se_ratio", C.c_float),
        ("chess_features", C.c_int), ("piece_square_tables", C.c_int), ("policy_factor_rank", C.c_int),
        ("norm_group", C.c_int), ("activation", C.c_int), ("value_activation", C.c_int), ("preact", C.c_int),
        ("self_supervised", C.c_int), ("ssl_tasks", C.c_int), ("infer_attention_stride", C.c_int),
    ]


_lib = None


class EngineLibraryMissing(RuntimeError):
    pass


def lib():
    """Load libm0engine.so once; raise if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). matrix0_amd has no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 and libm0engine.so is linked against the
    # system one with the same SONAME.  If torch comes first the loader hands libm0engine the copy already resident;
    # the other way round the process ends up with two runtimes and the second to touch the device finds none
    # (observed: build() then smoke() in one interpreter).  torch is a dependency anyway (checkpoints, distributed).
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    L.m0_last_error.restype = C.c_char_p
    L.m0_version.restype = C.c_char_p
    L.m0_net_create.restype = C.c_void_p
    L.m0_net_create.argtypes = [C.POINTER(NetCfg), C.c_int]
    L.m0_net_destroy.argtypes = [C.c_void_p]
    L.m0_net_destroy.restype = None
    L.m0_net_load_weight.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_int]
    L.m0_net_finalize.argtypes = [C.c_void_p]
    L.m0_net_infer.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.m0_net_ssl_channels.argtypes = [C.c_void_p]
    L.m0_net_param_count.argtypes = [C.c_void_p]
    L.m0_net_param_count.restype = C.c_int64
    L.m0_net_flops_per_position.argtypes = [C.c_void_p, C.c_int]
    L.m0_net_flops_per_position.restype = C.c_double
    L.m0_net_bench_forward.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
    L.m0_net_profile_enable.argtypes = [C.c_void_p, C.c_int]
    L.m0_net_profile_get.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]
    L.m0_net_profile_get_tail.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    _lib = L
    return L


def device_count() -> int:
    return int(lib().m0_device_count())


def last_error() -> str:
    return (lib().m0_last_error() or b"").decode("utf-8", "replace")


def check(rc: int, what: str = "m0 call"):
    """Non-zero return -> exception types the reference's callers catch
    (mcts.py:623: TimeoutError/RuntimeError; NaN/Inf -> ValueError, tests/test_error_handling.py)."""
    if rc == M0_OK:
        return
    msg = f"{what} failed ({rc}): {last_error()}"
    if rc == M0_ERR_NONFINITE:
        raise ValueError(msg)
    if rc == M0_ERR_INVALID:
        raise ValueError(msg)
    raise RuntimeError(msg)


def net_cfg_from_dict(d: dict) -> NetCfg:
    """NetConfig dict (config.yaml `model:` section, resnet.py:247-282 defaults) -> C struct."""
    g = d.get
    tasks = g("ssl_tasks", ["piece"]) if g("self_supervised", True) else []
    bits = 0
    for t in tasks:
        bits |= SSL_BITS.get(t, 0)
    c = NetCfg()
    c.planes = int(g("planes", 19)); c.channels = int(g("channels", 160)); c.blocks = int(g("blocks", 14))
    c.attention = int(bool(g("attention", True))); c.attention_heads = int(g("attention_heads", 8))
    c.attention_every_k = int(g("attention_every_k", 3)); c.attention_relbias = int(bool(g("attention_relbias", True)))
    c.attention_unmasked_mix = float(g("attention_unmasked_mix", 0.2))
    c.se = int(bool(g("se", True))); c.se_ratio = float(g("se_ratio", 0.25))
    c.chess_features = int(bool(g("chess_features", True))); c.piece_square_tables = int(bool(g("piece_square_tables", True)))
    c.policy_factor_rank = int(g("policy_factor_rank", 0))
    c.norm_group = 1 if g("norm", "batch") == "group" else 0
    c.activation = ACT.get(g("activation", "relu"), 1)
    c.value_activation = ACT.get(g("value_activation", "silu"), 1)
    c.preact = int(bool(g("preact", False)))
    c.self_supervised = int(bool(g("self_supervised", True))); c.ssl_tasks = bits
    c.infer_attention_stride = max(1, int(g("infer_attention_stride", 1)))
    if int(g("policy_size", 4672)) != 4672:
        raise ValueError("Unsupported policy_size; only the legacy 4672 mapping exists (resnet.py:302-306)")
    return c
