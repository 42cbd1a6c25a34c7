"""wsmgmap.ops.norm — channel reductions, train-mode BatchNorm (+ residual)(+ ReLU) with statistics from the producing convolution's
epilogue, inference GroupNorm.
"""
import ctypes

import torch

from .. import _abi
from ..debug import sw
from .core import _p, _raw_stream, _stream, _req, _f32, _sfx, _workspace, _rows_of, _conv_out, _launch, _zeros_f32, _prelaid, _join_side_at_end, TokenGradSink


def channel_sum(x2d):
    _req(x2d)
    rows, C = x2d.shape
    out = torch.empty(C, device=x2d.device, dtype=torch.float32)
    ws = _workspace(x2d.device)
    _abi.call("wsmg_channel_sum" + _sfx(x2d), _p(x2d), rows, C, _p(out), _p(ws), ws.numel() * 8, _stream())
    return out


# ----------------------------------------------------------------------------- batch norm (+res)(+relu)
class GradLoSink:
    """COMPUTE_DTYPE = "bf16+f32grad" (round 6): links the BatchNorm of a conv + BatchNorm layer to its convolution.  The BatchNorm's
    backward leaves here the low half of its float32 input gradient (dx_lo = bf16(dx_f32 - bf16(dx_f32))); the convolution's
    backward, which runs next on the same gradient, adds the weight gradient of (x, dx_lo) to that of (x, dx): the weight gradient of
    a 16-mantissa-bit dY from two launches of the bf16 kernel."""

    def __init__(self):
        self.lo = None

    def take(self):
        lo, self.lo = self.lo, None
        return lo


class _BnAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, gamma, beta, running_mean, running_var, train, relu, momentum, eps, stats=None, lo_sink=None):
        _req(x, residual, gamma, beta, running_mean, running_var)
        _f32(gamma, beta, running_mean, running_var)
        sfx = _sfx(x)
        if residual is not None and residual.dtype != x.dtype:
            raise _abi.WsmgError("residual must have the activation dtype")
        C = x.shape[-1]
        rows = x.numel() // C
        y = torch.empty_like(x)
        mean = torch.empty(C, device=x.device, dtype=torch.float32)
        invstd = torch.empty(C, device=x.device, dtype=torch.float32)
        if stats is not None and train and sfx:
            # the producing convolution left the sums of this tensor in `stats`: finalize (+ clear) and apply, no pass over x
            _abi.call("wsmg_bn_act_fwd_bf16_pre", _p(x), _p(residual), _p(gamma), _p(beta), _p(running_mean), _p(running_var),
                      float(momentum), float(eps), int(relu), rows, C, _p(y), _p(mean), _p(invstd), _p(stats), stats.shape[0],
                      _stream())
        else:
            ws = _workspace(x.device)
            _abi.call("wsmg_bn_act_fwd" + sfx, _p(x), _p(residual), _p(gamma), _p(beta), _p(running_mean), _p(running_var),
                      float(momentum), float(eps), int(train), int(relu), rows, C, _p(y), _p(mean), _p(invstd),
                      _p(ws), ws.numel() * 8, _stream())
        if train:   # the kernel wrote the running statistics through raw pointers: tell autograd's version counters, which
            #         the rollout route's FoldCache keys its folded operands on
            torch.autograd.graph.increment_version([running_mean, running_var])
        # without a residual the ReLU mask is recomputed from x in the backward kernels: y is neither kept nor read
        keep_y = relu and residual is not None
        ctx.save_for_backward(x, y if keep_y else None, gamma, beta, mean, invstd)
        ctx.cfg = (rows, C, int(relu), residual is not None, bool(train), sfx)
        ctx.lo_sink = lo_sink if (sfx and train and C % 8 == 0) else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, beta, mean, invstd = ctx.saved_tensors
        rows, C, relu, has_res, train, sfx = ctx.cfg
        if not train:
            raise _abi.WsmgError("backward through eval-mode BatchNorm is not part of the reference's path")
        dy, ld = _rows_of(dy, C)
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if has_res else None
        dgamma = torch.empty(C, device=x.device, dtype=torch.float32)
        dbeta = torch.empty(C, device=x.device, dtype=torch.float32)
        ws = _workspace(x.device)
        if ctx.lo_sink is not None:    # + the low half of the float32 gradient for the convolution's weight gradient (GradLoSink)
            lo = torch.empty_like(x)
            _abi.call("wsmg_bn_act_bwd_ld_bf16_lo", _p(dy), ld, _p(x), _p(y), _p(gamma), _p(beta), _p(mean), _p(invstd), relu, rows, C,
                      _p(dx), _p(lo), _p(dres), _p(dgamma), _p(dbeta), _p(ws), ws.numel() * 8, _stream())
            ctx.lo_sink.lo = lo
            return dx, dres, dgamma, dbeta, None, None, None, None, None, None, None, None
        if ld != C:     # a channel slice of a concatenation's gradient, read in place
            _abi.call("wsmg_bn_act_bwd_ld" + sfx, _p(dy), ld, _p(x), _p(y), _p(gamma), _p(beta), _p(mean), _p(invstd), relu, rows, C,
                      _p(dx), _p(dres), _p(dgamma), _p(dbeta), _p(ws), ws.numel() * 8, _stream())
        else:
            _abi.call("wsmg_bn_act_bwd" + sfx, _p(dy), _p(x), _p(y), _p(gamma), _p(beta), _p(mean), _p(invstd), relu, rows, C,
                      _p(dx), _p(dres), _p(dgamma), _p(dbeta), _p(ws), ws.numel() * 8, _stream())
        return dx, dres, dgamma, dbeta, None, None, None, None, None, None, None, None


def bn_act(x, gamma, beta, running_mean, running_var, train, relu=True, residual=None, momentum=0.1, eps=1e-5, stats=None, lo_sink=None):
    return _BnAct.apply(x, residual, gamma, beta, running_mean, running_var, train, relu, momentum, eps, stats, lo_sink)


BN_SLABS = 64
_bn_slabs = {}


def bn_stats_slabs(key, C, device):
    """Float64 [BN_SLABS, 2, C] accumulator a convolution's epilogue adds its output's per-channel sums into and the
    following train-mode BatchNorm consumes AND CLEARS (wsmg_bn_act_fwd_bf16_pre) — one persistent buffer per BatchNorm
    layer, zero between uses.  Returns None when the fused statistics are off (debug.sw.bn_fused_stats).  If a forward pass
    died between the two launches the buffer is dirty: `in_use` catches that and it is zeroed again."""
    if not sw.bn_fused_stats:
        return None
    k = (key, C, device.index)
    e = _bn_slabs.get(k)
    if e is None:
        e = _bn_slabs[k] = dict(buf=torch.zeros(BN_SLABS, 2, C, device=device, dtype=torch.float64), in_use=False)
    if e["in_use"]:
        e["buf"].zero_()
    e["in_use"] = True
    return e["buf"]


def bn_stats_done(key, C, device):
    e = _bn_slabs.get((key, C, device.index))
    if e is not None:
        e["in_use"] = False


@torch.no_grad()
def group_norm_nhwc(x, gamma, beta, groups, eps, relu, residual=None):
    """nn.GroupNorm(groups, C) [+ residual] [+ ReLU] on an NHWC bf16 activation (inference only: the frozen depth backbone)."""
    _req(x, gamma, beta, residual)
    _f32(gamma, beta)
    if x.dtype not in (torch.bfloat16, torch.float32) or (
            residual is not None and (residual.dtype != torch.bfloat16 or residual.shape != x.shape)):
        raise _abi.WsmgError("group_norm_nhwc: float32 / bf16 NHWC input, bf16 residual of the same shape")
    B, H, W, C = x.shape
    y = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    _abi.call("wsmg_group_norm_nhwc_bf16", _p(x), int(x.dtype == torch.float32), _p(residual), _p(gamma), _p(beta), B, H * W, C,
              int(groups), float(eps), int(bool(relu)), _p(y), _stream())
    return y
