"""wsmgmap.ops.nhwc — the small NHWC operators around the convolutions: ReLU, pools, bilinear upsampling (+ concatenation), layout
changes at the engine's boundary, channel concatenation, gradient fan-in, token mean.
"""
import ctypes

import torch

from .. import _abi
from ..debug import sw
from .core import _p, _raw_stream, _stream, _req, _f32, _sfx, _workspace, _rows_of, _conv_out, _launch, _zeros_f32, _prelaid, _join_side_at_end, TokenGradSink


# ----------------------------------------------------------------------------- small NHWC ops
class _Relu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _req(x)
        y = torch.empty_like(x)
        _abi.call("wsmg_relu_fwd" + _sfx(x), _p(x), _p(y), x.numel(), _stream())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        _abi.call("wsmg_relu_bwd" + _sfx(y), _p(dy), _p(y), _p(dx), y.numel(), _stream())
        return dx


class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _req(x)
        B, H, W, C = x.shape
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty(B, OH, OW, C, device=x.device, dtype=x.dtype)
        ctx.dims = (B, H, W, C, OH, OW)
        if x.dtype == torch.bfloat16 and x.requires_grad:
            # training: keep every output element's winning tap (one byte) — the backward then reads four taps per input element
            # instead of recomputing four windows' arg-max from x
            idx = torch.empty(B, OH, OW, C // 4, device=x.device, dtype=torch.int32)
            _abi.call("wsmg_maxpool3x3s2_fwd_idx_bf16", _p(x), _p(y), _p(idx), B, H, W, C, OH, OW, _stream())
            ctx.save_for_backward(idx)
            ctx.by_idx = True
            return y
        _abi.call("wsmg_maxpool3x3s2_fwd" + _sfx(x), _p(x), _p(y), B, H, W, C, OH, OW, _stream())
        ctx.save_for_backward(x)
        ctx.by_idx = False
        return y

    @staticmethod
    def backward(ctx, dy):
        B, H, W, C, OH, OW = ctx.dims
        dy = dy.contiguous()
        dx = torch.empty(B, H, W, C, device=dy.device, dtype=dy.dtype)
        if ctx.by_idx:
            (idx,) = ctx.saved_tensors
            _abi.call("wsmg_maxpool3x3s2_bwd_idx_bf16", _p(dy), _p(idx), _p(dx), B, H, W, C, OH, OW, _stream())
            return dx
        (x,) = ctx.saved_tensors
        _abi.call("wsmg_maxpool3x3s2_bwd" + _sfx(x), _p(dy), _p(x), _p(dx), B, H, W, C, OH, OW, _stream())
        return dx


class _Up2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _req(x)
        B, H, W, C = x.shape
        y = torch.empty(B, 2 * H, 2 * W, C, device=x.device, dtype=x.dtype)
        _abi.call("wsmg_upsample2x_fwd" + _sfx(x), _p(x), _p(y), B, H, W, C, _stream())
        ctx.shape = (B, H, W, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, H, W, C = ctx.shape
        dy, ld = _rows_of(dy, C)
        dx = torch.empty(B, H, W, C, device=dy.device, dtype=dy.dtype)
        if ld != C:      # a channel slice of a concatenation's gradient, read in place
            _abi.call("wsmg_upsample2x_bwd_ld" + _sfx(dy), _p(dy), ld, _p(dx), B, H, W, C, _stream())
        else:
            _abi.call("wsmg_upsample2x_bwd" + _sfx(dy), _p(dy), _p(dx), B, H, W, C, _stream())
        return dx


class _AvgPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _req(x)
        B, H, W, C = x.shape
        y = torch.empty(B, H // 2, W // 2, C, device=x.device, dtype=x.dtype)
        _abi.call("wsmg_avgpool2_fwd" + _sfx(x), _p(x), _p(y), B, H, W, C, _stream())
        ctx.shape = (B, H, W, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, H, W, C = ctx.shape
        dy = dy.contiguous()
        dx = torch.empty(B, H, W, C, device=dy.device, dtype=dy.dtype)
        _abi.call("wsmg_avgpool2_bwd" + _sfx(dy), _p(dy), _p(dx), B, H, W, C, _stream())
        return dx


class _ToNHWC(torch.autograd.Function):
    """float32 [B,C,H,W] contiguous -> [B,H,W,c_dst] (zero-padded channels) in float32 or bf16."""

    @staticmethod
    def forward(ctx, x, c_dst, dtype):
        _req(x)
        _f32(x)
        B, C, H, W = x.shape
        y = torch.empty(B, H, W, c_dst, device=x.device, dtype=dtype)
        _abi.call("wsmg_nchw_to_nhwc" + _sfx(y), _p(x), _p(y), B, C, H, W, c_dst, _stream())
        ctx.shape = (B, C, H, W, c_dst)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W, c_dst = ctx.shape
        dy = dy.contiguous()
        dx = torch.empty(B, C, H, W, device=dy.device, dtype=torch.float32)
        _abi.call("wsmg_nhwc_to_nchw" + _sfx(dy), _p(dy), _p(dx), B, c_dst, H, W, C, _stream())
        return dx, None, None


class _ToNCHW(torch.autograd.Function):
    """[B,H,W,C] float32 or bf16 -> float32 [B,c_dst,H,W] contiguous (drops padded channels)."""

    @staticmethod
    def forward(ctx, x, c_dst):
        _req(x)
        B, H, W, C = x.shape
        y = torch.empty(B, c_dst, H, W, device=x.device, dtype=torch.float32)
        _abi.call("wsmg_nhwc_to_nchw" + _sfx(x), _p(x), _p(y), B, C, H, W, c_dst, _stream())
        ctx.shape = (B, H, W, C, c_dst, x.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, H, W, C, c_dst, dtype = ctx.shape
        dy = dy.contiguous().float()
        dx = torch.empty(B, H, W, C, device=dy.device, dtype=dtype)
        _abi.call("wsmg_nchw_to_nhwc" + _sfx(dx), _p(dy), _p(dx), B, c_dst, H, W, C, _stream())
        return dx, None


relu = _Relu.apply
maxpool3x3s2 = _MaxPool.apply
upsample2x = _Up2.apply
avgpool2 = _AvgPool2.apply


def to_nhwc(x, c_dst=None, dtype=torch.float32):
    return _ToNHWC.apply(x, x.shape[1] if c_dst is None else c_dst, dtype)


def to_nchw(x, c_dst=None):
    return _ToNCHW.apply(x, x.shape[-1] if c_dst is None else c_dst)


class _CatChannels(torch.autograd.Function):
    """torch.cat([a, b], dim=-1) of two NHWC tensors in one 16-byte-vectorised launch (wsmg_cat_channels); the gradients
    are the two channel slices of dy (views, as torch.cat returns them)."""

    @staticmethod
    def forward(ctx, a, b):
        _req(a, b)
        if a.dtype != b.dtype or a.shape[:-1] != b.shape[:-1]:
            raise _abi.WsmgError("cat_channels: tensors must agree in dtype and in every dimension but the last")
        ca, cb = a.shape[-1], b.shape[-1]
        y = torch.empty(a.shape[:-1] + (ca + cb,), device=a.device, dtype=a.dtype)
        es = a.element_size()
        _abi.call("wsmg_cat_channels", _p(a), _p(b), _p(y), a.numel() // ca, ca * es, cb * es, _stream())
        ctx.ca = ca
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy[..., :ctx.ca], dy[..., ctx.ca:]


class _Fanout3(torch.autograd.Function):
    """Three aliases of one activation for its three consumers; backward adds the three gradients in ONE pass (wsmg_add3_bf16,
    float32 sums, one rounding) where autograd's own accumulation is two add launches over the tensor."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x), x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, g0, g1, g2):
        gs = [g for g in (g0, g1, g2) if g is not None]
        if (len(gs) == 3 and all(g.is_cuda and g.dtype == torch.bfloat16 and g.shape == gs[0].shape for g in gs)
                and gs[0].numel() % 8 == 0):
            gs = [g.contiguous() for g in gs]
            if all(g.data_ptr() % 16 == 0 for g in gs):
                out = torch.empty_like(gs[0])
                _abi.call("wsmg_add3_bf16", _p(gs[0]), _p(gs[1]), _p(gs[2]), _p(out), out.numel(), _stream())
                return out
        if not gs:
            return None
        out = gs[0]
        for g in gs[1:]:
            out = out + g
        return out


def fanout3(x):
    """(x, x, x) for an activation with three consumers, whose gradients then meet in one launch (see _Fanout3)."""
    if not (x.requires_grad and torch.is_grad_enabled()):
        return x, x, x
    return _Fanout3.apply(x)


class _Up2Cat(torch.autograd.Function):
    """cat([upsample2x(a), b], channels) in one launch, with autograd: the gradients are the upsampling's backward of dy's first
    channel slice (read in place: wsmg_upsample2x_bwd_ld) and dy's second slice as a view, as `_Up2` + `_CatChannels` return them."""

    @staticmethod
    def forward(ctx, a, b):
        B, H, W, Ca = a.shape
        Cb = b.shape[-1]
        y = torch.empty(B, 2 * H, 2 * W, Ca + Cb, device=a.device, dtype=torch.bfloat16)
        _abi.call("wsmg_upsample2x_cat_bf16", _p(a), _p(b), _p(y), B, H, W, Ca, Cb, _stream())
        ctx.shape = (B, H, W, Ca)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, H, W, Ca = ctx.shape
        da = None
        if ctx.needs_input_grad[0]:
            part, ld = _rows_of(dy[..., :Ca], Ca)
            da = torch.empty(B, H, W, Ca, device=dy.device, dtype=dy.dtype)
            if ld != Ca:
                _abi.call("wsmg_upsample2x_bwd_ld_bf16", _p(part), ld, _p(da), B, H, W, Ca, _stream())
            else:
                _abi.call("wsmg_upsample2x_bwd_bf16", _p(part), _p(da), B, H, W, Ca, _stream())
        return da, (dy[..., Ca:] if ctx.needs_input_grad[1] else None)


def upsample2x_cat(a, b):
    """cat([upsample2x(a), b], channels) of bf16 NHWC activations in one launch (wsmg_upsample2x_cat_bf16: the upsampled tensor
    is never materialised), forward and — round 3 — under autograd.  Other types / channel counts: the two separate operators."""
    if a.dtype != torch.bfloat16 or b.dtype != torch.bfloat16 or a.shape[-1] % 8 or b.shape[-1] % 8:
        return cat_channels(upsample2x(a), b)
    _req(a, b)
    B, H, W, Ca = a.shape
    if b.shape[:3] != (B, 2 * H, 2 * W):
        raise _abi.WsmgError(f"upsample2x_cat: {tuple(b.shape)} is not twice the size of {tuple(a.shape)}")
    return _Up2Cat.apply(a.contiguous(), b.contiguous())


def cat_channels(a, b):
    """Channel concatenation of two NHWC activations (falls back to torch.cat when a channel run is not a multiple of 16
    bytes)."""
    es = a.element_size()
    if (a.shape[-1] * es) % 16 or (b.shape[-1] * es) % 16:
        return torch.cat([a, b], dim=-1)
    return _CatChannels.apply(a.contiguous(), b.contiguous())


class _TokenMean(torch.autograd.Function):
    """mean over the token axis of x [B, I, C] in float32.  The gradient of a mean is one [B, C] row repeated over the
    I tokens: it is returned as a stride-0 expanded view in x's dtype (autograd adds it to the attention's gradient of
    the same tokens in one pass) instead of MeanBackward's materialised float32 [B, I, C] tensor, its conversion
    and the add (div 112 us + copy 92 us + add 73 us at B=512, I=576, C=256)."""

    @staticmethod
    def forward(ctx, x, sink=None):
        ctx.shape, ctx.dtype, ctx.sink = x.shape, x.dtype, sink
        if sink is not None:
            ctx.save_for_backward(x)
        return x.mean(dim=1, dtype=torch.float32)

    @staticmethod
    def backward(ctx, g):
        sink = ctx.sink
        dx = sink.take() if sink is not None else None
        if dx is not None:       # the attention's gradient is parked: merge the broadcast row and the ReLU mask in one pass
            (x,) = ctx.saved_tensors
            B, I, C = ctx.shape
            g = g.contiguous().float()
            _req(dx, x, g)
            _abi.call("wsmg_token_grad_merge" + _sfx(dx), _p(dx), _p(x), _p(g), B, I, C, int(sink.relu), _stream())
            sink.masked = bool(sink.relu)
            return dx, None
        return (g * (1.0 / ctx.shape[1])).to(ctx.dtype).unsqueeze(1).expand(ctx.shape), None


def mean_last(x):
    """x [..., n] float32 (n <= 160) -> mean over the last axis, one launch (no gradient: the caller's input is a cached feature)."""
    _req(x)
    _f32(x)
    n = x.shape[-1]
    out = torch.empty(x.shape[:-1], device=x.device, dtype=torch.float32)
    _abi.call("wsmg_mean_rows", _p(x), x.numel() // n, n, _p(out), _stream())
    return out


def token_mean(x, sink=None):
    return _TokenMean.apply(x, sink)
