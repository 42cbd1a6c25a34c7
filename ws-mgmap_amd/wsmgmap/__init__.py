"""wsmgmap — MI355X-native implementation of WS-MGMap's per-step policy hot path.

Host side: Python on PyTorch-ROCm (device memory, streams, autograd tape, torch.distributed).
Compute of the three hot operators: hand-written gfx950 HIP kernels in libwsmgmap.so, bound
through the C ABI of include/wsmgmap.h (see _abi.py).  There is no CPU / eager fallback.
"""
from . import _abi  # noqa: F401

__all__ = ["_abi"]


def _prefer_rocblas():
    """The policy's dense GEMMs outside the conv engine are small (heads, RNN projections, [256,512] x [512,256] weight
    gradients).  PyTorch-ROCm's default hipBLASLt heuristic picks a 256x256 macro-tile for several of them — ONE
    workgroup, 118 us for a 67-MFLOP product that rocBLAS runs in 7 us (measured on MI355X, tools/bench notes in
    DESIGN.md) — so rocBLAS is selected process-wide when the package is imported.  WSMG_KEEP_BLAS=1 leaves PyTorch's
    choice alone."""
    import os
    if os.environ.get("WSMG_KEEP_BLAS", "0") == "1":
        return
    try:
        import warnings
        import torch
        if torch.version.hip:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                torch.backends.cuda.preferred_blas_library("cublas")   # "cublas" = rocBLAS on ROCm builds
    except Exception:   # pragma: no cover  (older torch: keep the default)
        pass


_prefer_rocblas()
