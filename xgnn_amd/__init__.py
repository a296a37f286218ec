"""xgnn_amd -- MI355X-native GGMS hot path (neighbour sampling + feature extract).

The product is ``lib/libggms_hip.so`` (hand-written HIP for gfx950 behind a C
ABI, ``include/ggms.h`` and ``include/samgraph.h``).  This package is the thin
Python side: a ctypes loader (`_lib`), tensor-level operator wrappers (`ops`)
and the mirror of the reference's ``samgraph`` / ``samgraph.torch`` façade.

There is no CPU fallback: if the HIP library is missing or no GPU is visible,
operators raise.
"""
from ._lib import lib, GgmsError, LIB_PATH  # noqa: F401

__all__ = ["lib", "GgmsError", "LIB_PATH"]
