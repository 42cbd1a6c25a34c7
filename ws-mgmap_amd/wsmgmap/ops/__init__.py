"""Host-side operators of the hot path: thin torch.autograd.Function wrappers around the C ABI of libwsmgmap.so, one module per
operator family (round 4: the 1 981-line ops.py, split) —

    core       argument checks, pointers / streams, live kernel timing, section marks, per-pass state, stream joins
    conv       the map stack's convolutions (operator 2)            norm    BatchNorm / GroupNorm / channel sums
    nhwc       ReLU, pools, upsampling, layout changes, concatenation   attention   operator 3 (+ e4m3 forms)
    heads      fused losses and heads                                rnn     persistent GRU / bi-LSTM
    bev        operator 1

PyTorch is used only for device memory, streams and the autograd tape; every FLOP of the three named operators runs in the
hand-written gfx950 kernels.  All tensors are contiguous and resident on the GPU; anything else raises (there is no CPU or eager
fallback).  `wsmgmap.ops.<name>` keeps working for every name the flat module had."""
from . import core, nhwc, norm, conv, attention, heads, rnn, bev      # noqa: F401
from .. import _abi                                                       # noqa: F401

for _m in (core, nhwc, norm, conv, attention, heads, rnn, bev):
    for _k, _v in vars(_m).items():
        if not _k.startswith("__") and _k not in ("ctypes", "torch", "sw"):
            globals()[_k] = _v
del _m, _k, _v
