"""wsmgmap.ops.conv — the map stack's convolutions on the gfx950 conv engine (operator 2): weight layouts, forward / backward-data /
deterministic weight gradient, ConvTranspose2d, inference-only forms with folded BatchNorm and split-K, multi-tensor copies.
"""
import ctypes

import torch

from .. import _abi
from ..debug import sw
from .core import _p, _raw_stream, _stream, _req, _f32, _sfx, _workspace, _rows_of, _conv_out, _launch, _zeros_f32, _prelaid, _join_side_at_end, TokenGradSink
from .nhwc import cat_channels
from .norm import channel_sum

# ----------------------------------------------------------------------------- convolution


def prelayout_conv_weights(weights, dtype):
    """Lay out the operands of all the given OIHW float32 convolution parameters in ONE launch (wsmg_weight_relayout_multi)
    at the start of a forward pass; `_weight_layouts` then finds them instead of launching per layer.  Input channels are
    padded to the engine's multiple of 32.  The cache lives until the next forward pass (reset_pass_state)."""
    descs, keep = [], []
    for w in weights:
        if w.dim() != 4 or not w.is_cuda or w.dtype != torch.float32 or not w.is_contiguous():
            continue
        O, I, KH, KW = w.shape
        cin_pad = (I + 31) // 32 * 32
        key = (w.data_ptr(), w._version, cin_pad, dtype)
        if key in _prelaid:
            continue
        ohwi = torch.empty(O, KH, KW, cin_pad, device=w.device, dtype=dtype)
        ihwo = torch.empty(cin_pad, KH, KW, O, device=w.device, dtype=dtype)
        _prelaid[key] = (ohwi, ihwo)
        descs.append(_abi.RelayoutDesc(w.data_ptr(), ohwi.data_ptr(), ihwo.data_ptr(), O, I, KH, KW, cin_pad, 0))
        keep.append(w)
    if descs:
        arr = (_abi.RelayoutDesc * len(descs))(*descs)
        _abi.call("wsmg_weight_relayout_multi", ctypes.cast(arr, ctypes.c_void_p), len(descs), int(dtype == torch.bfloat16), _stream())


def _weight_layouts(w_oihw, cin_pad, dtype, need_ihwo):
    """(OHWI, IHWO or None) of an OIHW float32 parameter in `dtype`, input channels zero-padded to cin_pad: ONE launch
    (wsmg_weight_relayout) instead of permute copy + cast + second permute (+ pad)."""
    hit = _prelaid.get((w_oihw.data_ptr(), w_oihw._version, cin_pad, dtype))
    if hit is not None:
        return hit[0], (hit[1] if need_ihwo else None)
    O, I, KH, KW = w_oihw.shape
    w_ohwi = torch.empty(O, KH, KW, cin_pad, device=w_oihw.device, dtype=dtype)
    w_ihwo = torch.empty(cin_pad, KH, KW, O, device=w_oihw.device, dtype=dtype) if need_ihwo else None
    _abi.call("wsmg_weight_relayout" + ("_bf16" if dtype == torch.bfloat16 else ""), _p(w_oihw), O, I, KH, KW, cin_pad,
              _p(w_ohwi), _p(w_ihwo), _stream())
    return w_ohwi, w_ihwo


def _weight_grad_oihw(dw_ohwi, I):
    # (deferring these conversions to one multi-tensor launch at the end of the backward pass was tried: 12.84 vs 12.69 ms per
    # update — they already overlap the next layers' launches, and the merged launch sits on the critical path before Adam)
    O, KH, KW, Ipad = dw_ohwi.shape
    out = torch.empty(O, I, KH, KW, device=dw_ohwi.device, dtype=torch.float32)
    _abi.call("wsmg_weight_grad_to_oihw", _p(dw_ohwi), O, I, KH, KW, Ipad, _p(out), _stream())
    return out


# Deterministic weight gradients (C ABI: wsmg_conv2d_bwd_weight[_bf16]_plan / _slabs + wsmg_weight_grad_reduce_oihw): the
# weight-gradient kernels' workgroups STORE their partial tiles into slabs of a workspace and one more launch adds the slabs in
# a fixed order while it re-lays dW out as OIHW — no float atomics, no zero-fill of dW, bit-identical gradients from run to
# run (the reference sets cudnn.deterministic, run.py:107-108).  debug.sw.wgrad_atomics restores the atomic form (A/B).
_wgrad_ws = {}        # (device, stream) -> float32 workspace


def _wgrad_workspace(device, floats):
    """One slab workspace per stream, shared by the layers (launches of one stream use it one after the other)."""
    key = (device.index, _raw_stream())
    ws = _wgrad_ws.get(key)
    if ws is None or ws.numel() < floats:
        if ws is not None:
            ws.record_stream(torch.cuda.current_stream())      # launches that still read the old one are queued on this stream
        ws = torch.empty(max(int(floats), 1 << 24), device=device, dtype=torch.float32)
        _wgrad_ws[key] = ws
    return ws


def _weight_grad(sfx, x, dy, dims, fl, Cin_w):
    """OIHW float32 weight gradient [Cout, Cin_w, KH, KW] of the convolution `dims` from x [B,H,W,Cin] and dy [B,OH,OW,Cout]: the
    weight-gradient kernel's workgroups store their partial tiles into slabs, one more launch adds the slabs in a fixed order and
    lays dW out as OIHW (bit-reproducible).  debug.sw.wgrad_atomics: the float-atomics form (A/B)."""
    B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW = dims
    if sw.wgrad_atomics:
        dw_ohwi = _zeros_f32((Cout, KH, KW, Cin), x.device)
        _launch("wsmg_conv2d_bwd_weight" + sfx, fl, _p(x), _p(dy), _p(dw_ohwi), *dims, _stream())
        return _weight_grad_oihw(dw_ohwi, Cin_w)
    nsplit, floats = ctypes.c_int(0), ctypes.c_longlong(0)
    _abi.call("wsmg_conv2d_bwd_weight" + sfx + "_plan", *dims, ctypes.cast(ctypes.byref(nsplit), ctypes.c_void_p),
              ctypes.cast(ctypes.byref(floats), ctypes.c_void_p))
    ws = _wgrad_workspace(x.device, floats.value)
    _launch("wsmg_conv2d_bwd_weight" + sfx + "_slabs", fl, _p(x), _p(dy), _p(ws), nsplit.value, floats.value, *dims, _stream())
    out = torch.empty(Cout, Cin_w, KH, KW, device=x.device, dtype=torch.float32)
    # (timed with the family it belongs to: ADVICE r03 — its 0.18 ms per update were left out of the weight-gradient family's total)
    _launch("wsmg_weight_grad_reduce_oihw", 0.0, _p(ws), nsplit.value, Cout, Cin_w, KH, KW, Cin, _p(out), _stream(),
            prof_as="wsmg_conv2d_bwd_weight" + sfx + "_slabs")
    return out


class ReluSink:
    """Links a convolution whose ReLU was fused into its forward epilogue (no BatchNorm: map_encoded_linear, map_classified_linear,
    mg_map_policy.py:89-96) to the operator that consumes its output: when that operator's gradient kernel masks the gradient with
    the ReLU itself (wsmg_conv2d_bwd_data_bf16_ex's relu_y) it sets `masked`, and this convolution's backward skips its own masking pass.
    (ops.TokenGradSink plays the same role for map_cated_linear.)"""

    def __init__(self):
        self.relu = False
        self.masked = False


def _mask_of(mask_z, mask_sinks):
    """The tensor a gradient kernel masks its output with: `mask_z` (the consumed tensor = the producers' ReLU outputs) when every
    producer's ReLU is fused and the switch is on; marks the producers so that their own mask passes do not run."""
    if mask_z is None or not mask_sinks or not sw.relu_producer_mask or not all(m.relu for m in mask_sinks):
        return None
    for m in mask_sinks:
        m.masked = True
    return mask_z


class _Conv2d(torch.autograd.Function):
    """y = conv2d(x, w) + b on NHWC x; w is the reference's OIHW float32 parameter (its .grad comes back OIHW
    float32).  x float32 -> f32 MFMA engine; x bf16 -> bf16 operands, float32 accumulation and dW.  If x has more
    channels than w (the engine pads activations to multiples of 32) the weight is zero-padded to match."""

    @staticmethod
    def forward(ctx, x, w_oihw, bias, stride, pad, bias_grad_zero=False, relu=False, stats=None, relu_sink=None, into=None, lo_sink=None):
        _req(x, w_oihw, bias)
        _f32(w_oihw, bias)
        sfx = _sfx(x)
        B, H, W, Cin = x.shape
        Cout, Cin_w, KH, KW = w_oihw.shape
        assert Cin >= Cin_w, (x.shape, w_oihw.shape)
        w, w_ihwo = _weight_layouts(w_oihw.contiguous(), Cin, x.dtype, ctx.needs_input_grad[0])
        OH, OW = _conv_out(H, KH, stride, pad), _conv_out(W, KW, stride, pad)
        fl = 2.0 * B * OH * OW * Cout * Cin * KH * KW
        dims = (B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW)
        if into is not None and sfx and stats is None:
            # the output goes straight into its channel slice of the concatenation that follows (base [B,OH,OW,Ctot], first channel)
            base, c_off = into
            y = base[..., c_off:c_off + Cout]
            _launch("wsmg_conv2d_fwd_bf16_ex", fl, _p(x), _p(w), _p(bias), ctypes.c_void_p(base.data_ptr() + c_off * 2), 2 if relu else 0,
                    None, 0, base.shape[-1], *dims, _stream())
        else:
            y = torch.empty(B, OH, OW, Cout, device=x.device, dtype=x.dtype)
            if sfx and stats is not None:   # + the output's BatchNorm sums in the epilogue (see bn_stats_slabs)
                _launch("wsmg_conv2d_fwd_bf16_stats", fl, _p(x), _p(w), _p(bias), _p(y), 2 if relu else 0, _p(stats), stats.shape[0],
                        *dims, _stream())
            elif sfx:   # ReLU, when asked for, runs in the conv epilogue (flag bit 1)
                _launch("wsmg_conv2d_fwd_bf16", fl, _p(x), _p(w), _p(bias), _p(y), 2 if relu else 0, *dims, _stream())
            else:
                _launch("wsmg_conv2d_fwd", fl, _p(x), _p(w), _p(bias), _p(y), *dims, _stream())
                if relu:
                    _abi.call("wsmg_relu_fwd", _p(y), _p(y), y.numel(), _stream())
        ctx.save_for_backward(x, w_ihwo, y if relu else None)
        ctx.cfg = dims + (bias is not None, sfx, Cin_w)
        ctx.bias_grad_zero = bool(bias_grad_zero)
        ctx.relu_sink = relu_sink if relu else None
        if ctx.relu_sink is not None:
            relu_sink.relu = True
        ctx.lo_sink = lo_sink if sfx else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w_ihwo, y_relu = ctx.saved_tensors
        *dims, has_bias, sfx, Cin_w = ctx.cfg
        B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW = dims
        if ctx.relu_sink is not None and ctx.relu_sink.masked:
            y_relu = None        # the gradient's producer applied the ReLU mask already (TokenGradSink / ReluSink)
        if y_relu is not None:   # fused ReLU: mask the incoming gradient with the saved output first
            masked = torch.empty(y_relu.shape, device=y_relu.device, dtype=y_relu.dtype)
            dy, ld = _rows_of(dy, Cout) if sfx else (dy.contiguous(), Cout)
            if not y_relu.is_contiguous():
                y_relu = y_relu.contiguous()
            if ld != Cout:       # a channel slice of a concatenation's gradient, read in place
                _abi.call("wsmg_relu_bwd_rows_bf16", _p(dy), ld, _p(y_relu), _p(masked), dy.numel() // Cout, Cout, _stream())
            else:
                _abi.call("wsmg_relu_bwd" + sfx, _p(dy), _p(y_relu), _p(masked), dy.numel(), _stream())
            dy = masked
        else:
            dy = dy.contiguous()
        dx = dw = db = None
        fl = 2.0 * B * OH * OW * Cout * Cin * KH * KW
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            if sfx:
                _launch("wsmg_conv2d_bwd_data_bf16", fl, _p(dy), _p(w_ihwo), _p(dx), 0, *dims, _stream())
            else:
                _launch("wsmg_conv2d_bwd_data", fl, _p(dy), _p(w_ihwo), _p(dx), *dims, _stream())
        if ctx.needs_input_grad[1]:
            dw = _weight_grad(sfx, x, dy, dims, fl, Cin_w)
            lo = ctx.lo_sink.take() if ctx.lo_sink is not None else None
            if lo is not None and lo.shape == dy.shape:
                # COMPUTE_DTYPE = "bf16+f32grad": + the weight gradient of the gradient's low half (norm.GradLoSink)
                dw = dw + _weight_grad(sfx, x, lo, dims, fl, Cin_w)
        if has_bias and ctx.needs_input_grad[2]:
            # a bias in front of a train-mode BatchNorm cancels in (x - mean): its gradient is sum(dy) = 0 exactly;
            # the caller says so and the channel reduction over dy is skipped
            db = _zeros_f32((Cout,), dy.device) if ctx.bias_grad_zero else channel_sum(dy.view(-1, Cout))
        return dx, dw, db, None, None, None, None, None, None, None, None


class _ConvT2d(torch.autograd.Function):
    """ConvTranspose2d(k4,s2,p1) = backward-data of the adjoint convolution.  `w_iohw` is the nn.ConvTranspose2d
    parameter [Cin_t, Cout_t, KH, KW] = the adjoint conv's OIHW weight (O = Cin_t channels on the small grid)."""

    @staticmethod
    def forward(ctx, x, w_iohw, stride, pad, stats=None):
        _req(x, w_iohw)
        _f32(w_iohw)
        sfx = _sfx(x)
        B, Hs, Ws, Ct_in = x.shape           # small grid (adjoint conv's output)
        O, I, KH, KW = w_iohw.shape          # adjoint conv: I channels (big grid) -> O channels (small grid)
        assert O == Ct_in
        Hb, Wb = (Hs - 1) * stride - 2 * pad + KH, (Ws - 1) * stride - 2 * pad + KW
        w, w_ihwo = _weight_layouts(w_iohw.contiguous(), I, x.dtype, True)
        y = torch.empty(B, Hb, Wb, I, device=x.device, dtype=x.dtype)
        fl = 2.0 * B * Hs * Ws * O * I * KH * KW
        dims = (B, Hb, Wb, I, O, KH, KW, stride, pad, Hs, Ws)
        if sfx and stats is not None:
            _launch("wsmg_conv2d_bwd_data_bf16_stats", fl, _p(x), _p(w_ihwo), _p(y), 0, _p(stats), stats.shape[0], *dims, _stream())
        elif sfx:
            _launch("wsmg_conv2d_bwd_data_bf16", fl, _p(x), _p(w_ihwo), _p(y), 0, *dims, _stream())
        else:
            _launch("wsmg_conv2d_bwd_data", fl, _p(x), _p(w_ihwo), _p(y), *dims, _stream())
        ctx.save_for_backward(x, w)
        ctx.cfg = dims + (sfx,)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        *dims, sfx = ctx.cfg
        B, Hb, Wb, I, O, KH, KW, stride, pad, Hs, Ws = dims
        dy = dy.contiguous()
        dx = dw = None
        fl = 2.0 * B * Hs * Ws * O * I * KH * KW
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            if sfx:
                _launch("wsmg_conv2d_fwd_bf16", fl, _p(dy), _p(w), None, _p(dx), 0, *dims, _stream())
            else:
                _launch("wsmg_conv2d_fwd", fl, _p(dy), _p(w), None, _p(dx), *dims, _stream())
        if ctx.needs_input_grad[1]:
            dw = _weight_grad(sfx, dy, x, tuple(dims), fl, I)
        return dx, dw, None, None, None


def conv2d(x, weight_oihw, bias, stride=1, pad=0, bias_grad_zero=False, relu=False, stats=None, relu_sink=None, into=None, lo_sink=None):
    """x NHWC; weight = the reference's OIHW parameter (laid out for the engine in one launch inside the autograd
    node, so the parameter's .grad comes back OIHW float32).  bias_grad_zero: the output feeds a train-mode
    BatchNorm, so d(loss)/d(bias) is identically zero and is returned as zeros.  relu: y = relu(conv + bias), fused into
    the conv epilogue in bf16 mode.  relu_sink: see ReluSink / TokenGradSink.  into = (base, c): write the output into channels
    [c, c + Cout) of the bf16 tensor `base` [B,OH,OW,Ctot] (the concatenation that follows) and return that slice.  lo_sink: see
    norm.GradLoSink (COMPUTE_DTYPE = "bf16+f32grad")."""
    return _Conv2d.apply(x, weight_oihw, bias, stride, pad, bias_grad_zero, relu, stats, relu_sink, into, lo_sink)


class _Conv2dCat2(torch.autograd.Function):
    """relu(conv2d(cat([a, b], channels), w) + bias), bf16, where a and b ARE the two channel slices of `base` — written in place
    by their producers (conv2d(into=(base, .))) — so that no concatenation pass runs in either direction: forward reads `base`;
    backward's input-gradient kernel masks the gradient with the producers' fused ReLUs (relu_y = base) and stores
    it as its two contiguous parts (dx2 / split_c), which go to the producers as they are.  mg_map_policy.py:89-100,197,207."""

    @staticmethod
    def forward(ctx, a, b, base, w_oihw, bias, relu, relu_sink, mask_sinks):
        _req(base, w_oihw, bias)
        _f32(w_oihw, bias)
        B, H, W, Cin = base.shape
        Cout, Cin_w, KH, KW = w_oihw.shape
        assert Cin == Cin_w and KH == 3 and KW == 3 and base.dtype == torch.bfloat16
        w, w_ihwo = _weight_layouts(w_oihw.contiguous(), Cin, base.dtype, True)
        y = torch.empty(B, H, W, Cout, device=base.device, dtype=base.dtype)
        fl = 2.0 * B * H * W * Cout * Cin * 9
        dims = (B, H, W, Cin, Cout, 3, 3, 1, 1, H, W)
        _launch("wsmg_conv2d_fwd_bf16", fl, _p(base), _p(w), _p(bias), _p(y), 2 if relu else 0, *dims, _stream())
        ctx.save_for_backward(base, w_ihwo, y if relu else None)
        ctx.cfg = dims + (bias is not None, a.shape[-1])
        ctx.relu_sink = relu_sink if relu else None
        if ctx.relu_sink is not None:
            relu_sink.relu = True
        ctx.mask_sinks = mask_sinks
        return y

    @staticmethod
    def backward(ctx, dy):
        base, w_ihwo, y_relu = ctx.saved_tensors
        *dims, has_bias, ca = ctx.cfg
        B, H, W, Cin, Cout = dims[:5]
        if ctx.relu_sink is not None and ctx.relu_sink.masked:
            y_relu = None
        if y_relu is not None:
            masked = torch.empty_like(y_relu)
            dy = dy.contiguous()
            _abi.call("wsmg_relu_bwd_bf16", _p(dy), _p(y_relu), _p(masked), dy.numel(), _stream())
            dy = masked
        else:
            dy = dy.contiguous()
        fl = 2.0 * B * H * W * Cout * Cin * 9
        da = torch.empty(B, H, W, ca, device=dy.device, dtype=dy.dtype)
        db = torch.empty(B, H, W, Cin - ca, device=dy.device, dtype=dy.dtype)
        _launch("wsmg_conv2d_bwd_data_bf16_ex", fl, _p(dy), _p(w_ihwo), _p(da), _p(_mask_of(base, ctx.mask_sinks)), _p(db), ca, *dims, _stream())
        dw = _weight_grad("_bf16", base, dy, tuple(dims), fl, Cin) if ctx.needs_input_grad[3] else None
        dbias = channel_sum(dy.view(-1, Cout)) if (has_bias and ctx.needs_input_grad[4]) else None
        return da, db, None, dw, dbias, None, None, None


def conv2d_over_written_parts(a, b, base, weight_oihw, bias, relu=True, relu_sink=None, mask_sinks=None):
    """See _Conv2dCat2; falls back to cat_channels + conv2d when a / b are not the in-place slices of `base`."""
    ca, cb = a.shape[-1], b.shape[-1]
    ok = (base.dtype == torch.bfloat16 and base.is_contiguous() and base.shape[-1] == ca + cb and ca % 8 == 0 and cb % 8 == 0
          and a.data_ptr() == base.data_ptr() and b.data_ptr() == base.data_ptr() + ca * base.element_size()
          and a.stride() == base.stride() and b.stride() == base.stride() and tuple(weight_oihw.shape[1:]) == (ca + cb, 3, 3))
    if not ok:
        return conv2d(cat_channels(a.contiguous(), b.contiguous()), weight_oihw, bias, 1, 1, relu=relu, relu_sink=relu_sink)
    return _Conv2dCat2.apply(a, b, base, weight_oihw, bias, relu, relu_sink, mask_sinks)


def conv2d_cat(xs, weight_oihw, bias, stride=1, pad=0, bias_grad_zero=False, relu=False, relu_sink=None):
    """conv2d over the channel concatenation of the NHWC tensors `xs`: one vectorised concatenation (wsmg_cat_channels), then the
    convolution.  (Running the convolution part by part over the weight's input-channel slices — no concatenated tensor — was
    built and measured in round 2: 15.30-15.39 vs 15.14 ms per update; two shorter reductions, two epilogues and twice the
    weight-gradient launches cost more than the copies they remove.  The code is gone, the measurement is in DESIGN.md section 7.)"""
    xs = list(xs)
    x = xs[0] if len(xs) == 1 else (cat_channels(xs[0], xs[1]) if len(xs) == 2 else torch.cat(xs, dim=-1))
    return conv2d(x, weight_oihw, bias, stride, pad, bias_grad_zero, relu, relu_sink=relu_sink)


_splitk_plans = {}    # layer geometry -> (ksplit, partial floats) from wsmg_conv2d_splitk_plan


def _splitk_plan(B, OH, OW, Cin, Cout, KH, KW):
    key = (B, OH, OW, Cin, Cout, KH, KW)
    plan = _splitk_plans.get(key)
    if plan is None:
        ks, floats = ctypes.c_int(0), ctypes.c_longlong(0)
        _abi.call("wsmg_conv2d_splitk_plan", B, OH, OW, Cin, Cout, KH, KW, ctypes.cast(ctypes.byref(ks), ctypes.c_void_p),
                  ctypes.cast(ctypes.byref(floats), ctypes.c_void_p))
        plan = _splitk_plans[key] = (ks.value, floats.value)
    return plan


def conv2d_infer_bf16(x, w_ohwi_bf16, bias, stride, pad, relu, out_f32=False, add_to=None):
    """Inference-only bf16 convolution with a pre-laid-out OHWI bf16 weight, float32 bias and optional fused ReLU
    (one launch; used by the frozen encoders with eval-mode BatchNorm folded into weight and bias).
    add_to: a tensor of the output's shape and type that the result is ADDED to, in place, before the ReLU (the identity branch
    of a residual block) — it is returned.  Layers too small to fill the chip run split-K (wsmg_conv2d_fwd_bf16_splitk);
    debug.sw.conv_splitk = False turns that off."""
    _req(x, w_ohwi_bf16, bias, add_to)
    if x.dtype != torch.bfloat16 or w_ohwi_bf16.dtype != torch.bfloat16:
        raise _abi.WsmgError("conv2d_infer_bf16 needs bf16 activations and weights")
    B, H, W, Cin = x.shape
    Cout, KH, KW, Cin2 = w_ohwi_bf16.shape
    assert Cin == Cin2, (x.shape, w_ohwi_bf16.shape)
    OH, OW = _conv_out(H, KH, stride, pad), _conv_out(W, KW, stride, pad)
    odt = torch.float32 if out_f32 else torch.bfloat16
    if add_to is not None:
        if add_to.shape != (B, OH, OW, Cout) or add_to.dtype != odt or add_to.device != x.device:
            raise _abi.WsmgError("conv2d_infer_bf16: add_to must have the output's shape, type and device")
        y = add_to
    else:
        y = torch.empty(B, OH, OW, Cout, device=x.device, dtype=odt)
    flags = (2 if relu else 0) | (1 if out_f32 else 0) | (4 if add_to is not None else 0)
    fl = 2.0 * B * OH * OW * Cout * Cin * KH * KW
    dims = (B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW)
    ks, floats = _splitk_plan(B, OH, OW, Cin, Cout, KH, KW) if sw.conv_splitk else (1, 0)
    if ks > 1:
        part = torch.empty(floats, device=x.device, dtype=torch.float32)
        _launch("wsmg_conv2d_fwd_bf16_splitk", fl, _p(x), _p(w_ohwi_bf16), _p(bias), _p(y), flags, ks, _p(part), *dims, _stream())
    else:
        _launch("wsmg_conv2d_fwd_bf16", fl, _p(x), _p(w_ohwi_bf16), _p(bias), _p(y), flags, *dims, _stream())
    return y


@torch.no_grad()
def copy_multi(dsts, srcs):
    """dst.copy_(src) for a list of GPU tensor pairs in ONE launch (wsmg_copy_multi) — same shape, dtype and device per pair, both
    contiguous; anything else falls back to torch's copy for that pair."""
    pairs = []
    for d, s in zip(dsts, srcs):
        if (d.is_cuda and s.is_cuda and d.device == s.device and d.dtype == s.dtype and d.shape == s.shape and d.is_contiguous()
                and s.is_contiguous()):
            if d.numel():
                pairs.append((d, s))
        else:
            d.copy_(s)
    if not pairs:
        return
    arr = (_abi.CopyDesc * len(pairs))()
    for i, (d, s) in enumerate(pairs):
        arr[i].dst, arr[i].src, arr[i].bytes = d.data_ptr(), s.data_ptr(), d.numel() * d.element_size()
    _abi.call("wsmg_copy_multi", ctypes.cast(arr, ctypes.c_void_p), len(pairs), _stream())


def conv_transpose2d_infer_bf16(x, w_ihwo_bf16, bias, stride, pad, relu):
    """Inference-only nn.ConvTranspose2d on a bf16 NHWC activation with a pre-laid-out IHWO bf16 weight ([Cout_t, KH, KW, Cin_t]
    of the module's [Cin_t, Cout_t, KH, KW] parameter), float32 bias and optional fused ReLU — one launch
    (wsmg_conv_transpose2d_infer_bf16; eval-mode BatchNorm folded into weight and bias by FoldCache)."""
    _req(x, w_ihwo_bf16, bias)
    if x.dtype != torch.bfloat16 or w_ihwo_bf16.dtype != torch.bfloat16:
        raise _abi.WsmgError("conv_transpose2d_infer_bf16 needs bf16 activations and weights")
    B, Hs, Ws, Ct_in = x.shape
    I, KH, KW, O = w_ihwo_bf16.shape
    assert O == Ct_in, (x.shape, w_ihwo_bf16.shape)
    Hb, Wb = (Hs - 1) * stride - 2 * pad + KH, (Ws - 1) * stride - 2 * pad + KW
    y = torch.empty(B, Hb, Wb, I, device=x.device, dtype=torch.bfloat16)
    _launch("wsmg_conv_transpose2d_infer_bf16", 2.0 * B * Hs * Ws * O * I * KH * KW, _p(x), _p(w_ihwo_bf16), _p(bias), _p(y),
            2 if relu else 0, B, Hb, Wb, I, O, KH, KW, stride, pad, Hs, Ws, _stream())
    return y


def conv_transpose2d(x, weight_iohw, stride=2, pad=1, stats=None):
    """nn.ConvTranspose2d weight is [Cin_t, Cout_t, KH, KW] = the adjoint conv's OIHW."""
    return _ConvT2d.apply(x, weight_iohw, stride, pad, stats)
