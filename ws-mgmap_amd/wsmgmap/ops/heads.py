"""wsmgmap.ops.heads — losses and heads as fused launches: NHWC cross-entropy, the contrastive monitor's KL, the classifier tail, the
update path's heads / auxiliary-loss reduction / DAgger loss, and the rollout's one-launch dense layers and heads.
"""
import ctypes

import torch

from .. import _abi
from ..debug import sw
from .core import _p, _raw_stream, _stream, _req, _f32, _sfx, _workspace, _rows_of, _conv_out, _launch, _zeros_f32, _prelaid, _join_side_at_end, TokenGradSink


# ----------------------------------------------------------------------------- persistent masked GRU
class _CrossEntropyNHWC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, classes):
        _req(logits, target)
        if logits.shape[-1] != 32 or target.dtype != torch.int64:
            raise _abi.WsmgError("cross_entropy_nhwc: logits [..., 32] (padded classes), target int64")
        rows = logits.numel() // 32
        loss = torch.empty(target.shape, device=logits.device, dtype=torch.float32)
        _abi.call("wsmg_ce_nhwc_fwd" + _sfx(logits), _p(logits), _p(target), rows, int(classes), _p(loss), _stream())
        ctx.save_for_backward(logits, target)
        ctx.classes = int(classes)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        logits, target = ctx.saved_tensors
        g = gloss.contiguous().float()
        d = torch.empty_like(logits)
        _abi.call("wsmg_ce_nhwc_bwd" + _sfx(logits), _p(logits), _p(target), _p(g), logits.numel() // 32, ctx.classes, _p(d), _stream())
        return d, None, None


def cross_entropy_nhwc(logits, target, classes):
    """F.cross_entropy(logits_nchw[:, :classes], target, reduction='none') computed from NHWC logits [..., 32]."""
    return _CrossEntropyNHWC.apply(logits.contiguous(), target.contiguous(), classes)


class _PathKL(torch.autograd.Function):
    """kl[b] of the contrastive monitor (policy.py:72-82 of the reference) in one launch per direction (csrc/wsmg_loss.hip)."""

    @staticmethod
    def forward(ctx, dis, att, size, tau):
        _req(dis, att)
        _f32(dis, att)
        B, H, W = dis.shape
        if att.shape != (B, size * size):
            raise _abi.WsmgError(f"path_kl: attention {tuple(att.shape)} does not match {B} x {size}^2")
        lo, hi = torch.aminmax(dis)      # batch-global normalisation, as the reference does (dis.max(), dis.min())
        target = torch.empty(B, size * size, device=dis.device, dtype=torch.float32)
        kl = torch.empty(B, device=dis.device, dtype=torch.float32)
        _abi.call("wsmg_path_kl_fwd", _p(dis), _p(lo), _p(hi), _p(att), B, H, W, size, float(tau), _p(target), _p(kl), _stream())
        ctx.save_for_backward(target, att)
        return kl

    @staticmethod
    def backward(ctx, gkl):
        target, att = ctx.saved_tensors
        B, n = target.shape
        datt = torch.empty_like(att)
        _abi.call("wsmg_path_kl_bwd", _p(gkl.contiguous().float()), _p(target), _p(att), B, n, _p(datt), _stream())
        return None, datt, None, None


def path_kl(dis, att, size, tau):
    """F.kl_div(log(att), softmax(area_resize((hi - dis) / (hi - lo), size) / tau), reduction='none').mean(-1) with lo, hi the
    batch-global extremes of dis [B, H, W]; att [B, size*size] (a probability row); -> [B]."""
    return _PathKL.apply(dis.contiguous(), att.contiguous(), int(size), float(tau))


# ----------------------------------------------------------------------------- the semantic classifier's tail, fused
class _ClsTail(torch.autograd.Function):
    """BatchNorm2d(32, batch statistics) + ReLU + Conv2d(32, classes, 1) + { per-sample cross-entropy against the nearest-resized
    ground truth, AvgPool2d(2), the logits } in one pass per direction over the 32-channel activation (csrc/wsmg_cls_tail.hip):
    mg_map_policy.py:78-86,93-96,195 and policy.py:61-66 of the reference.  bf16 training mode only."""

    @staticmethod
    def forward(ctx, y2, stats, gamma, beta, running_mean, running_var, momentum, eps, w6, b6, gt):
        _req(y2, stats, gamma, beta, running_mean, running_var, w6, b6, gt)
        _f32(gamma, beta, running_mean, running_var, w6, b6, gt)
        B, H, W, C = y2.shape
        classes = w6.shape[0]
        if y2.dtype != torch.bfloat16 or C != 32 or w6.numel() != classes * 32 or b6.numel() != classes or stats.dtype != torch.float64:
            raise _abi.WsmgError("cls_tail: bf16 [B,H,W,32] activation, [classes,32,1,1] weight, float64 statistics slabs")
        if gt is not None and (gt.dim() != 3 or gt.shape[0] != B):
            raise _abi.WsmgError("cls_tail: ground truth [B, Hg, Wg] float32")
        dev = y2.device
        mean, invstd = torch.empty(32, device=dev), torch.empty(32, device=dev)
        _abi.call("wsmg_bn_stats_finalize", _p(stats), stats.shape[0], 32, B * H * W, float(momentum), float(eps), _p(running_mean),
                  _p(running_var), _p(mean), _p(invstd), _stream())
        torch.autograd.graph.increment_version([running_mean, running_var])
        sem = torch.empty(B, H, W, 32, device=dev, dtype=torch.bfloat16)
        pooled = torch.empty(B, H // 2, W // 2, 32, device=dev, dtype=torch.bfloat16)
        ce = torch.empty(B, device=dev, dtype=torch.float32) if gt is not None else None
        Hg, Wg = (gt.shape[1], gt.shape[2]) if gt is not None else (0, 0)
        _abi.call("wsmg_cls_tail_fwd_bf16", _p(y2), _p(gamma), _p(beta), _p(mean), _p(invstd), _p(w6), _p(b6), classes, _p(gt), Hg, Wg,
                  B, H, W, _p(sem), _p(pooled), _p(ce), _stream())
        ctx.save_for_backward(y2, gamma, beta, mean, invstd, w6, b6, gt)
        ctx.mark_non_differentiable(sem)
        ctx.set_materialize_grads(False)
        return sem, pooled, ce

    @staticmethod
    def backward(ctx, _dsem, dpooled, dce):
        y2, gamma, beta, mean, invstd, w6, b6, gt = ctx.saved_tensors
        B, H, W, _ = y2.shape
        classes = w6.shape[0]
        dev = y2.device
        dpooled = None if dpooled is None else dpooled.contiguous()
        dce = None if dce is None else dce.contiguous().float()
        if dpooled is not None and dpooled.dtype != torch.bfloat16:
            raise _abi.WsmgError("cls_tail: the pooled map's gradient must be bf16")
        Hg, Wg = (gt.shape[1], gt.shape[2]) if gt is not None else (0, 0)
        dbn = torch.empty_like(y2)
        nws = int(_abi.lib().wsmg_cls_tail_workspace_floats(B))
        ws = torch.empty(nws, device=dev, dtype=torch.float32)
        dgamma, dbeta = torch.empty(32, device=dev), torch.empty(32, device=dev)
        dw6, db6 = torch.empty_like(w6), torch.empty_like(b6)
        _abi.call("wsmg_cls_tail_bwd_bf16", _p(y2), _p(gamma), _p(beta), _p(mean), _p(invstd), _p(w6), _p(b6), classes,
                  _p(gt if dce is not None else None), Hg, Wg, _p(dce), _p(dpooled), B, H, W, _p(dbn), _p(ws), nws, _p(dgamma), _p(dbeta),
                  _p(dw6), _p(db6), _stream())
        # BatchNorm's apply pass, in place (element i of dx depends on element i of dy and x only)
        _abi.call("wsmg_bn_bwd_apply_bf16", _p(dbn), _p(y2), _p(gamma), _p(mean), _p(invstd), _p(dgamma), _p(dbeta), B * H * W, 32, _p(dbn),
                  _stream())
        return dbn, None, dgamma, dbeta, None, None, None, None, dw6, db6, None


def cls_tail_ok(y2, classes):
    """Can `cls_tail` take this activation?  bf16 [B, H, W, 32] with H even and W a multiple of 16, at most 32 classes."""
    return (y2.is_cuda and y2.dtype == torch.bfloat16 and y2.dim() == 4 and y2.shape[3] == 32 and y2.shape[1] % 2 == 0
            and y2.shape[2] % 16 == 0 and classes <= 32 and sw.fused_cls_tail)


def cls_tail(y2, stats, bn, conv1x1, gt=None):
    """-> (logits [B,H,W,32] bf16 — not differentiable, for inspection / pred_sem_map —, pooled [B,H/2,W/2,32] bf16, ce_rows [B] or
    None): train-mode `bn` (nn.BatchNorm2d(32), statistics in `stats` from the producing convolution's epilogue) + ReLU + `conv1x1`
    (nn.Conv2d(32, classes, 1)) + the prediction monitor's per-sample cross-entropy against gt [B,Hg,Wg] + AvgPool2d(2).
    A label outside [0, classes) — `F.cross_entropy` of the reference faults on it — makes that sample's loss row NaN (and with it
    the loss and the gradients behind it): loud, not a finite wrong number.  The loss is taken from the float32 logits; the
    unfused route takes it from the stored bf16 logits (the difference is inside the 5e-3 bar of the tests)."""
    w, b = conv1x1.weight, conv1x1.bias
    return _ClsTail.apply(y2.contiguous(), stats, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps,
                          w.contiguous(), b, None if gt is None else gt.contiguous())


# ----------------------------------------------------------------------------- update-path heads, auxiliary reduction, trainer loss
class _UpdateHeads(torch.autograd.Function):
    """(pred [B,A], prog [B,1], progress-loss rows [B]) of the update path in one launch per direction (csrc/wsmg_heads.hip):
    `action_distribution.fc_mean`, `tanh(prog_pred(.))` and `mse_loss(prog, progress, 'none').mean(-1)` of the reference's
    BasePolicy.forward / aux_prediction (models/policy.py:59,86-88,96-97)."""

    @staticmethod
    def forward(ctx, x, wm, bm, wp, bp, progress):
        _req(x, wm, bm, wp, bp, progress)
        _f32(x, wm, bm, wp, bp, progress)
        B, K = x.shape
        A = wm.shape[0]
        if wm.shape != (A, K) or wp.numel() != K or bm.numel() != A or bp.numel() != 1 or (progress is not None and progress.numel() != B):
            raise _abi.WsmgError("update_heads: head shapes do not fit the features")
        pred = torch.empty(B, A, device=x.device, dtype=torch.float32)
        prog = torch.empty(B, 1, device=x.device, dtype=torch.float32)
        rows = torch.empty(B, device=x.device, dtype=torch.float32) if progress is not None else None
        _abi.call("wsmg_update_heads_fwd", _p(x), _p(wm), _p(bm), _p(wp), _p(bp), _p(progress), B, K, A, _p(pred), _p(prog), _p(rows), _stream())
        ctx.save_for_backward(x, wm, wp, prog, progress)
        ctx.set_materialize_grads(False)
        return pred, prog, rows

    @staticmethod
    def backward(ctx, dpred, dprog, drows):
        x, wm, wp, prog, progress = ctx.saved_tensors
        B, K = x.shape
        A = wm.shape[0]
        c = lambda t: None if t is None else t.contiguous().float()   # noqa: E731
        dpred, dprog, drows = c(dpred), c(dprog), c(drows)
        dx = torch.empty_like(x)
        dwm, dbm = torch.empty_like(wm), torch.empty(A, device=x.device, dtype=torch.float32)
        dwp, dbp = torch.empty_like(wp), torch.empty(1, device=x.device, dtype=torch.float32)
        _abi.call("wsmg_update_heads_bwd", _p(x), _p(wm), _p(wp), _p(prog), _p(progress), _p(dpred), _p(dprog), _p(drows), B, K, A,
                  _p(dx), _p(dwm), _p(dbm), _p(dwp), _p(dbp), _stream())
        return dx, dwm, dbm, dwp, dbp, None


def update_heads(features, fc_mean, prog_pred, progress=None):
    """-> (pred [B,A], prog [B,1], rows [B] or None): action mean, tanh progress head and — with `progress` [B,1] — the progress
    monitor's per-row squared error, one launch; fc_mean / prog_pred are the nn.Linear modules."""
    return _UpdateHeads.apply(features.contiguous(), fc_mean.weight, fc_mean.bias, prog_pred.weight, prog_pred.bias,
                              None if progress is None else progress.contiguous())


class _AuxReduce(torch.autograd.Function):
    """_AuxLosses.reduce(mask) (common/aux_losses.py:24-35) over up to 4 per-row loss vectors: one launch per direction."""

    @staticmethod
    def forward(ctx, mask, alphas, *rows):
        _req(mask, *rows)
        _f32(*rows)
        B = rows[0].numel()
        if mask.dtype != torch.bool or mask.numel() != B or any(r.numel() != B for r in rows) or not 1 <= len(rows) <= 4:
            raise _abi.WsmgError("aux_reduce: 1-4 float32 loss vectors and a bool mask of one length")
        L = len(rows)
        ptrs = (ctypes.c_void_p * L)(*[r.data_ptr() for r in rows])
        al = (ctypes.c_float * L)(*[float(a) for a in alphas])
        out = torch.empty(2, device=mask.device, dtype=torch.float32)
        _abi.call("wsmg_aux_reduce_fwd", ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(al, ctypes.c_void_p), L, _p(mask), B, _p(out), _stream())
        ctx.save_for_backward(mask, out)
        ctx.alphas, ctx.shapes = tuple(float(a) for a in alphas), [r.shape for r in rows]
        return out[0]

    @staticmethod
    def backward(ctx, daux):
        mask, out = ctx.saved_tensors
        L, B = len(ctx.alphas), mask.numel()
        al = (ctypes.c_float * L)(*ctx.alphas)
        drows = torch.empty(L, B, device=mask.device, dtype=torch.float32)
        _abi.call("wsmg_aux_reduce_bwd", ctypes.cast(al, ctypes.c_void_p), L, _p(mask), _p(out[1:]), _p(daux.contiguous().float()), B, _p(drows),
                  _stream())
        return (None, None) + tuple(drows[k].view(shp) for k, shp in enumerate(ctx.shapes))


def aux_reduce(rows, alphas, mask):
    """sum_k alphas[k] * mean(rows[k][mask]) as a 0-dim tensor (rows: list of [B] float32 CUDA tensors, mask [B] bool)."""
    return _AuxReduce.apply(mask.contiguous(), tuple(alphas), *[r.contiguous() for r in rows])


class _DaggerLoss(torch.autograd.Function):
    """The trainer's loss of one update (dagger_trainer.py:526-534): weighted squared error of tanh(pred) against the waypoint,
    per-episode weight normalisation, mean over episodes, + the auxiliary loss — one launch per direction."""

    @staticmethod
    def forward(ctx, pred, waypoint, weights, aux):
        _req(pred, waypoint, weights, aux)
        _f32(pred, waypoint, weights, aux)
        T, N = weights.shape
        A = pred.shape[-1]
        if pred.numel() != T * N * A or waypoint.shape[0] != T * N or waypoint.shape[-1] < A or waypoint.dim() != 2:
            raise _abi.WsmgError("dagger_loss: pred [T*N, A], waypoint [T*N, >= A], weights [T, N]")
        out = torch.empty(2, device=pred.device, dtype=torch.float32)
        den = torch.empty(N, device=pred.device, dtype=torch.float32)
        _abi.call("wsmg_dagger_loss_fwd", _p(pred), _p(waypoint), waypoint.shape[-1], _p(weights), _p(aux), T, N, A, _p(out), _p(den), _stream())
        ctx.save_for_backward(pred, waypoint, weights, den)
        ctx.has_aux = aux is not None
        loss, action = out[0], out[1]
        ctx.mark_non_differentiable(action)
        return loss, action

    @staticmethod
    def backward(ctx, dloss, _daction):
        pred, waypoint, weights, den = ctx.saved_tensors
        T, N = weights.shape
        A = pred.shape[-1]
        dloss = dloss.contiguous().float()
        dpred = torch.empty_like(pred)
        _abi.call("wsmg_dagger_loss_bwd", _p(pred), _p(waypoint), waypoint.shape[-1], _p(weights), _p(den), _p(dloss), T, N, A, _p(dpred), _stream())
        return dpred, None, None, (dloss.reshape(1) if ctx.has_aux else None)


def dagger_loss(pred, aux_loss, waypoint, weights):
    """-> (loss, action_loss) as 0-dim tensors; pred [T*N, A], waypoint [T*N, >= A] (its first A columns are the target),
    weights [T, N]; aux_loss a 0-dim tensor, a Python number (added on the host side of the graph) or None."""
    aux_t = aux_loss if torch.is_tensor(aux_loss) else None
    loss, action = _DaggerLoss.apply(pred.contiguous(), waypoint.contiguous(), weights.contiguous(),
                                     None if aux_t is None else aux_t.reshape(1).float())
    if aux_t is None and aux_loss is not None:
        loss = loss + float(aux_loss)
    return loss, action


ROWS_MAX = 16     # rollout-size dense layers: up to this many rows go through linear_rows / act_heads


def rows_route(x):
    """True when a dense layer on x [B, ...] should take the one-launch rollout route: no autograd, float32 on the GPU, at most
    ROWS_MAX rows.  debug.sw.rows_linear = False turns it off (the nn.Linear modules run)."""
    return (not torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and x.shape[0] <= ROWS_MAX
            and sw.rows_linear)


@torch.no_grad()
def linear_rows(x, weight, bias, act=None, pool=1):
    """act(x @ weight.T + bias) for a few rows in one launch (wsmg_linear_rows).  x [B, K] — or [B, K, pool], averaged over its
    last axis first; weight [O, K]; act None / "relu" / "tanh"."""
    x = x.contiguous()
    _req(x, weight, bias)
    _f32(x, weight, bias)
    B = x.shape[0]
    O, K = weight.shape
    if x.numel() != B * K * pool or (bias is not None and bias.numel() != O):
        raise _abi.WsmgError(f"linear_rows: x {tuple(x.shape)} does not fit weight {tuple(weight.shape)} (pool {pool})")
    y = torch.empty(B, O, device=x.device, dtype=torch.float32)
    _abi.call("wsmg_linear_rows", _p(x), _p(weight), _p(bias), _p(y), B, K, O, {None: 0, "relu": 1, "tanh": 2}[act], int(pool), _stream())
    return y


@torch.no_grad()
def act_heads(features, prog_pred, fc_mean, logstd, critic_fc, noise=None):
    """(prog [B,1], value [B,1], action [B,A], log-probability [B]) of one rollout step in one launch (wsmg_act_heads).
    prog_pred / fc_mean / critic_fc: nn.Linear modules; logstd: the [A, 1] parameter of DiagGaussian.logstd; noise: [B, A]
    standard normals for a sampled action, None for the mode."""
    features = features.contiguous()
    B, K = features.shape
    A = fc_mean.weight.shape[0]
    ls = logstd.reshape(-1)
    _req(features, prog_pred.weight, prog_pred.bias, fc_mean.weight, fc_mean.bias, ls, critic_fc.weight, critic_fc.bias, noise)
    _f32(features, prog_pred.weight, prog_pred.bias, fc_mean.weight, fc_mean.bias, ls, critic_fc.weight, critic_fc.bias, noise)
    if prog_pred.weight.shape != (1, K) or critic_fc.weight.shape != (1, K) or fc_mean.weight.shape[1] != K or ls.numel() != A or (
            noise is not None and noise.shape != (B, A)):
        raise _abi.WsmgError("act_heads: head shapes do not fit the features")
    dev = features.device
    prog, value = torch.empty(B, 1, device=dev), torch.empty(B, 1, device=dev)
    action, logp = torch.empty(B, A, device=dev), torch.empty(B, device=dev)
    _abi.call("wsmg_act_heads", _p(features), B, K, _p(prog_pred.weight), _p(prog_pred.bias), _p(fc_mean.weight), _p(fc_mean.bias),
              _p(ls), A, _p(critic_fc.weight), _p(critic_fc.bias), _p(noise), _p(prog), _p(value), _p(action), _p(logp), _stream())
    return prog, value, action, logp


@torch.no_grad()
def colsum_multi(mats):
    """[x.sum(0) for x in mats] for up to 16 contiguous float32 [rows, cols] GPU matrices in ONE launch (wsmg_colsum_multi): the bias
    gradients of the recurrent core's dense layers.  Fixed summation order (bit-reproducible)."""
    import ctypes
    _req(*mats)
    _f32(*mats)
    outs = [torch.empty(m.shape[1], device=m.device, dtype=torch.float32) for m in mats]
    for i in range(0, len(mats), 16):
        grp = list(zip(mats[i:i + 16], outs[i:i + 16]))
        arr = (_abi.ColsumDesc * len(grp))()
        for j, (m, o) in enumerate(grp):
            arr[j].x, arr[j].out, arr[j].rows, arr[j].cols = m.data_ptr(), o.data_ptr(), m.shape[0], m.shape[1]
        _abi.call("wsmg_colsum_multi", ctypes.cast(arr, ctypes.c_void_p), len(grp), _stream())
    return outs
