"""matrix0_amd — MI355X-native self-play hot path for Matrix0 (encode -> MCTS -> ResNet-24 forward).

Host-side mirrors of the reference seams; all compute is in csrc/ (HIP, gfx950) behind the
C-ABI of include/m0_engine.h.
"""
from ._lib import EngineLibraryMissing, LIB_PATH  # noqa: F401

__all__ = ["EngineLibraryMissing", "LIB_PATH"]
