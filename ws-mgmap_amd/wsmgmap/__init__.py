"""wsmgmap — MI355X-native implementation of WS-MGMap's per-step policy hot path.

Host side: Python on PyTorch-ROCm (device memory, streams, autograd tape, torch.distributed).
Compute of the three hot operators: hand-written gfx950 HIP kernels in libwsmgmap.so, bound
through the C ABI of include/wsmgmap.h (see _abi.py).  There is no CPU / eager fallback.
"""
from . import _abi  # noqa: F401

__all__ = ["_abi"]
