"""wsmgmap.ops.attention — the cross-modal single-query attention (operator 3): per-row, shared-set and folded-key forms, and the
e4m3 forms of BASELINE configs[4].
"""
import ctypes

import torch

from .. import _abi
from ..debug import sw
from .core import _p, _raw_stream, _stream, _req, _f32, _sfx, _workspace, _rows_of, _conv_out, _launch, _zeros_f32, _prelaid, _join_side_at_end, TokenGradSink


# ----------------------------------------------------------------------------- attention
class _Attn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, mask, scale):
        _req(q, k, v, mask)
        _f32(q)
        sfx = _sfx(k)
        if v.dtype != k.dtype:
            raise _abi.WsmgError("attention keys and values must share a dtype")
        B, I, C = k.shape
        if mask is not None and mask.dtype != torch.uint8:
            raise _abi.WsmgError("attention mask must be uint8")
        out = torch.empty(B, C, device=q.device, dtype=torch.float32)
        attn = torch.empty(B, I, device=q.device, dtype=torch.float32)
        _abi.call("wsmg_attn_fwd" + sfx, _p(q), _p(k), _p(v), _p(mask), float(scale), B, I, C, _p(out), _p(attn), _stream())
        ctx.save_for_backward(q, k, v, attn)
        ctx.scale = float(scale)
        return out, attn

    @staticmethod
    def backward(ctx, dout, dattn):
        q, k, v, attn = ctx.saved_tensors
        B, I, C = k.shape
        dout = torch.zeros_like(q) if dout is None else dout.contiguous().float()
        dattn = None if dattn is None else dattn.contiguous().float()
        dq = torch.empty_like(q)
        dk = torch.empty_like(k)
        dv = torch.empty_like(v)
        _abi.call("wsmg_attn_bwd" + _sfx(k), _p(q), _p(k), _p(v), _p(attn), _p(dout), _p(dattn), ctx.scale, B, I, C,
                  _p(dq), _p(dk), _p(dv), _stream())
        return dq, dk, dv, None, None


def attention(q, k, v, mask=None, scale=1.0 / 16):
    """q [B,C]; k, v [B,I,C] token-major; mask [B,I] bool/uint8 (True = padded token)."""
    if mask is not None and mask.dtype == torch.bool:
        mask = mask.to(torch.uint8)
    return _Attn.apply(q, k, v, mask, scale)


class _AttnShared(torch.autograd.Function):
    """Single-query attention of B rows over U << B shared key / value sets (row b uses set inverse[b]): the update path
    repeats every instruction T times; the reference (and `attention` above) would need per-row copies of the
    instruction keys and values.  Forward reads the sets in place; backward gets d logits from the kernel and forms
    dK_u = sum_{b in u} dl_b^T q_b and dV_u = sum_{b in u} attn_b^T dout_b with two batched GEMMs over a one-hot
    membership matrix — no [B, I, C] tensor exists in either direction."""

    @staticmethod
    def forward(ctx, q, k_sets, v_sets, mask_sets, inverse, scale):
        _req(q, k_sets, v_sets, mask_sets, inverse)
        _f32(q)
        if v_sets.dtype != k_sets.dtype or inverse.dtype != torch.int64:
            raise _abi.WsmgError("attention_shared: k/v sets share a dtype, inverse is int64")
        if mask_sets is not None and mask_sets.dtype != torch.uint8:
            raise _abi.WsmgError("attention mask must be uint8")
        U, I, C = k_sets.shape
        B = q.shape[0]
        out = torch.empty(B, C, device=q.device, dtype=torch.float32)
        attn = torch.empty(B, I, device=q.device, dtype=torch.float32)
        _abi.call("wsmg_attn_shared_fwd" + _sfx(k_sets), _p(q), _p(k_sets), _p(v_sets), _p(mask_sets), _p(inverse), float(scale),
                  B, I, C, _p(out), _p(attn), _stream())
        ctx.save_for_backward(q, k_sets, v_sets, attn, inverse)
        ctx.scale = float(scale)
        ctx.set_materialize_grads(False)
        return out, attn

    @staticmethod
    def backward(ctx, dout, dattn):
        q, k_sets, v_sets, attn, inverse = ctx.saved_tensors
        U, I, C = k_sets.shape
        B = q.shape[0]
        dout = torch.zeros_like(q) if dout is None else dout.contiguous().float()
        dattn = None if dattn is None else dattn.contiguous().float()
        dq = torch.empty_like(q)
        dl = torch.empty(B, I, device=q.device, dtype=torch.float32)
        _abi.call("wsmg_attn_shared_bwd" + _sfx(k_sets), _p(q), _p(k_sets), _p(v_sets), _p(attn), _p(dout), _p(dattn), _p(inverse),
                  ctx.scale, B, I, C, _p(dq), _p(dl), _stream())
        member = torch.nn.functional.one_hot(inverse, U).to(torch.float32).t()            # [U, B]
        dk = torch.matmul((member.unsqueeze(2) * dl.unsqueeze(0)).transpose(1, 2), q)       # [U, I, B] x [B, C]
        dv = torch.matmul((member.unsqueeze(2) * attn.unsqueeze(0)).transpose(1, 2), dout)
        return dq, dk.to(k_sets.dtype), dv.to(v_sets.dtype), None, None, None


def attention_shared(q, k_sets, v_sets, mask_sets, inverse, scale):
    """(context [B,C], weights [B,I]); row b attends over k_sets[inverse[b]], v_sets[inverse[b]], mask_sets[inverse[b]]."""
    return _AttnShared.apply(q, k_sets, v_sets, mask_sets, inverse, scale)


class _AttnFolded(torch.autograd.Function):
    """Single-query attention whose keys are a k=1 Conv1d of the values (mg_map_policy.py:126-132,173-178):
    q.(W x_i + b) = (W^T q).x_i + q.b, and q.b is the same for every token, so it cancels in the softmax.
    The projection is therefore folded into the query ([B,C] x [C,C]) and the tokens x are read once as both
    keys and values; no key tensor is ever materialised."""

    @staticmethod
    def forward(ctx, q, w, b, x, mask, scale, sink=None):
        _req(q, x, mask)
        _f32(q)
        B, I, C = x.shape
        ctx.sink = sink
        if mask is not None:
            mask = mask.to(torch.uint8).contiguous()
        wf = w.reshape(w.shape[0], -1).float()
        qf = (q @ wf).contiguous()
        out = torch.empty(B, C, device=q.device, dtype=torch.float32)
        attn = torch.empty(B, I, device=q.device, dtype=torch.float32)
        _abi.call("wsmg_attn_fwd" + _sfx(x), _p(qf), _p(x), _p(x), _p(mask), float(scale), B, I, C, _p(out), _p(attn), _stream())
        ctx.save_for_backward(q, wf, qf, x, attn)
        ctx.scale = float(scale)
        ctx.wshape = w.shape
        ctx.has_b = b is not None
        return out, attn

    @staticmethod
    def backward(ctx, dout, dattn):
        q, wf, qf, x, attn = ctx.saved_tensors
        B, I, C = x.shape
        dout = torch.zeros_like(q) if dout is None else dout.contiguous().float()
        dattn = None if dattn is None else dattn.contiguous().float()
        dqf = torch.empty_like(qf)
        dx = torch.empty_like(x)
        _abi.call("wsmg_attn_bwd" + _sfx(x), _p(qf), _p(x), _p(x), _p(attn), _p(dout), _p(dattn), ctx.scale, B, I, C,
                  _p(dqf), _p(dx), _p(dx), _stream())
        dq = dqf @ wf.t()
        dw = (q.t() @ dqf).reshape(ctx.wshape)
        db = torch.zeros(ctx.wshape[0], device=q.device, dtype=torch.float32) if ctx.has_b else None   # exactly zero
        if ctx.sink is not None:     # parked: the token mean's backward merges its row in and returns the sum (TokenGradSink)
            ctx.sink.park(dx)
            dx = None
        return dq, dw, db, dx, None, None, None


def attention_folded(q, w, b, x, mask, scale, sink=None):
    """(context [B,C], weights [B,I]) of softmax(scale * (q . (W x_i + b) - 1e8 mask_i)) over x [B,I,C]."""
    return _AttnFolded.apply(q, w, b, x, mask, scale, sink)


def quantize_e4m3(x, scale):
    """float32 tensor -> uint8 tensor of OCP e4m3 codes of x / scale (saturating, round to nearest even)."""
    _req(x)
    _f32(x)
    if x.numel() % 4:
        raise _abi.WsmgError("quantize_e4m3 needs a multiple of 4 elements")
    y = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
    _abi.call("wsmg_quantize_e4m3", _p(x), x.numel(), 1.0 / float(scale), _p(y), _stream())
    return y


_fp8_tickets = {}


def _fp8_scratch(B, L, device):
    """(workspace, ticket) of the split-row fp8 attention: the ticket words must be zero before a launch and every launch
    leaves them zero, so one zero-initialised buffer per (stream, B) serves all calls."""
    L_ = _abi.lib()
    ws = torch.empty(int(L_.wsmg_attn_fp8_workspace_bytes(B, L)) // 4, device=device, dtype=torch.float32)
    key = (device.index, _raw_stream(), B)
    t = _fp8_tickets.get(key)
    if t is None:
        t = _fp8_tickets[key] = torch.zeros(B, device=device, dtype=torch.int32)
    return ws, t


def _fp8_fold(x, w, transpose):
    B, C = x.shape
    out = torch.empty(B, C, device=x.device, dtype=torch.float32)
    _abi.call("wsmg_attn_fp8_fold", _p(x), _p(w), B, C, int(transpose), _p(out), _stream())
    return out


last_fp8_row_launches = 0       # launches the last single-query fp8 forward took (1: wsmg_attn_fp8_row_fwd)


def _fp8_forward(q, w2d, x_q, xs_t, lengths, scale):
    global last_fp8_row_launches
    B, L, C = x_q.shape
    if sw.fp8_row_fused and L <= 224 and C == 256:
        # round 6: fold + attention of a row in ONE launch, one workgroup per row (SURVEY 8d's configs[4]: every row its own tokens)
        qf = torch.empty(B, C, device=q.device, dtype=torch.float32)
        out = torch.empty(B, C, device=q.device, dtype=torch.float32)
        attn = torch.empty(B, L, device=q.device, dtype=torch.float32)
        _abi.call("wsmg_attn_fp8_row_fwd", _p(q), _p(w2d), _p(x_q), _p(xs_t), _p(lengths), float(scale), B, L, C, _p(qf), _p(out), _p(attn),
                  _stream())
        last_fp8_row_launches = 1
        return qf, out, attn
    last_fp8_row_launches = 2
    qf = _fp8_fold(q, w2d, False)                       # q W_k on the matrix cores (float32 MFMA)
    out = torch.empty(B, C, device=q.device, dtype=torch.float32)
    attn = torch.empty(B, L, device=q.device, dtype=torch.float32)
    ws, ticket = _fp8_scratch(B, L, q.device)
    _abi.call("wsmg_attn_fp8_fwd", _p(qf), _p(x_q), _p(xs_t), _p(lengths), float(scale), B, L, C, _p(out), _p(attn), _p(ws),
              _p(ticket), _stream())
    return qf, out, attn


def attn_fp8_fused(q, w_k, b_k, x_q, x_scale, lengths, scale):
    """Text attention of BASELINE configs[4] on pre-quantised tokens (no autograd): q [B,C] float32, w_k [C,C] / b_k [C] the
    k=1 Conv1d key projection (mg_map_policy.py:126-127; b_k cancels in the softmax and is not read), x_q [B,L,C] uint8
    e4m3 codes of the instruction embedding / x_scale (float or device scalar), lengths [B] int32.  Returns
    (out [B,C], attn [B,L]) = softmax((q.(W_k x + b_k) - 1e8 mask) * scale) applied to x."""
    _req(q, x_q, lengths)
    _f32(q)
    if x_q.dtype != torch.uint8 or (lengths is not None and lengths.dtype != torch.int32):
        raise _abi.WsmgError("attn_fp8_fused: x_q must be uint8 (e4m3 codes), lengths int32")
    xs_t = x_scale if torch.is_tensor(x_scale) else torch.full((1,), float(x_scale), device=q.device, dtype=torch.float32)
    w2d = w_k.reshape(w_k.shape[0], -1).float().contiguous()
    _, out, attn = _fp8_forward(q.contiguous(), w2d, x_q.contiguous(), xs_t, lengths, scale)
    return out, attn


class _AttnFp8(torch.autograd.Function):
    """Trainable form: x float32 [B,L,C] is quantised to e4m3 with one per-tensor scale (amax / 448, computed on the device),
    forward and backward read the BYTES; the gradient of x is the straight-through gradient of the de-quantised tokens."""

    @staticmethod
    def forward(ctx, q, w_k, b_k, x, lengths, scale):
        _req(q, x, lengths)
        _f32(q, x)
        B, L, C = x.shape
        xs_t = (x.detach().abs().amax() / 448.0).clamp_min(1e-30).reshape(1).float()
        x_q = torch.empty(B, L, C, device=x.device, dtype=torch.uint8)
        _abi.call("wsmg_quantize_e4m3_dev", _p(x), x.numel(), _p(xs_t), _p(x_q), _stream())
        w2d = w_k.reshape(w_k.shape[0], -1).float().contiguous()
        qf, out, attn = _fp8_forward(q.contiguous(), w2d, x_q, xs_t, lengths, scale)
        ctx.save_for_backward(q, w2d, qf, x_q, xs_t, attn)
        ctx.scale, ctx.wshape, ctx.has_b = float(scale), w_k.shape, b_k is not None
        ctx.set_materialize_grads(False)
        return out, attn

    @staticmethod
    def backward(ctx, dout, dattn):
        q, w2d, qf, x_q, xs_t, attn = ctx.saved_tensors
        B, L, C = x_q.shape
        dout = torch.zeros_like(q) if dout is None else dout.contiguous().float()
        dattn = None if dattn is None else dattn.contiguous().float()
        dqf = torch.empty_like(qf)
        dx = torch.empty(B, L, C, device=q.device, dtype=torch.float32) if ctx.needs_input_grad[3] else None
        _abi.call("wsmg_attn_fp8_bwd", _p(qf), _p(x_q), _p(xs_t), _p(attn), _p(dout), _p(dattn), ctx.scale, B, L, C, _p(dqf), _p(dx),
                  _stream())
        dq = _fp8_fold(dqf, w2d, True) if ctx.needs_input_grad[0] else None          # d q_f W_k^T, same MFMA kernel
        dw = (q.t() @ dqf).reshape(ctx.wshape) if ctx.needs_input_grad[1] else None   # [C_out, C_in] = q^T d q_f
        db = torch.zeros(ctx.wshape[0], device=q.device, dtype=torch.float32) if (ctx.has_b and ctx.needs_input_grad[2]) else None
        return dq, dw, db, dx, None, None


def attention_fp8(q, w_k, b_k, x, lengths, scale=1.0 / 16):
    """(context [B,C], weights [B,L]) of the state -> instruction attention with e4m3 token storage; differentiable in q, W_k, x."""
    return _AttnFp8.apply(q, w_k, b_k, x.contiguous(), lengths, scale)


last_fp8_shared_launches = 0      # launches the last attention_fp8_shared call took (1: the fused kernel)
_fp8_fused_ws = {}


def _fp8_fused_state(dev):
    """[workspace (16 zeroed words), arrivals so far, launches that took maxima so far] of wsmg_attn_fp8_mfma_fused for the current
    stream: the kernel's grid barrier counts arrivals on a monotonic counter, so the host tells every launch where the counter
    stands (one workspace per stream: launches on one stream do not overlap)."""
    key = (dev.index, _raw_stream())
    st = _fp8_fused_ws.get(key)
    if st is None:
        st = _fp8_fused_ws[key] = [torch.zeros(16, device=dev, dtype=torch.int32), 0, 0]
    return st


@torch.no_grad()
def attention_fp8_shared(q, k_sets, v_sets, lengths, inverse, scale=1.0 / 16, scales=None):
    """BASELINE configs[4] on the matrix cores (csrc/wsmg_attn_fp8_mfma.hip): `_attn` (mg_map_policy.py:173-178) of B rows over U
    shared instruction sets with e4m3 storage — S = Q K^T on v_mfma_f32_32x32x16_fp8_fp8, float32 softmax, O = P V on the bf16
    matrix pipe.  q [B,256] float32; k_sets, v_sets [U,L,256] float32 (token-major keys and values of each unique instruction);
    lengths [U] int (valid tokens; the rest are masked) or None; inverse [B] int64 (row b uses set inverse[b]).
    scales: (q_scale, k_scale, v_scale) Python floats for the quantisation; None: amax / 448 per tensor, computed on the device.
    Forward only (rollout / evaluation); -> (context [B,256], weights [B,L])."""
    _req(q, k_sets, v_sets, lengths, inverse)
    _f32(q, k_sets, v_sets)
    B, C = q.shape
    U, L, _ = k_sets.shape
    if C != 256 or k_sets.shape != v_sets.shape or k_sets.shape[2] != C or inverse.numel() != B or L > 224:
        raise _abi.WsmgError("attention_fp8_shared: q [B,256], k / v sets [U,L<=224,256], inverse [B]")
    dev = q.device
    if inverse.dtype != torch.int64:
        inverse = inverse.long()
    q, k_sets, v_sets, inverse = q.contiguous(), k_sets.contiguous(), v_sets.contiguous(), inverse.contiguous()
    s3 = [float(x) if x is not None else 0.0 for x in (scales or (None, None, None))]
    lens = None if lengths is None else lengths.to(torch.int32).contiguous()
    global last_fp8_shared_launches
    # round 5: ONE launch (scales, codes, row grouping and the attention: wsmg_attn_fp8_mfma_fused) whenever the maxima pass's grid
    # barrier is safe (<= 128 attention workgroups) or the caller fixed the scales; not under a HIP-graph capture (the arrival target
    # of the barrier is a launch argument)
    need = not all(x > 0.0 for x in s3)
    arrivals = int(_abi.lib().wsmg_attn_fp8_mfma_fused_arrivals(B, U, L, C)) if need else 0
    if sw.fp8_fused and (not need or arrivals > 0) and not torch.cuda.is_current_stream_capturing():
        st = _fp8_fused_state(dev)
        out = torch.empty(B, C, device=dev, dtype=torch.float32)
        attn = torch.empty(B, L, device=dev, dtype=torch.float32)
        _abi.call("wsmg_attn_fp8_mfma_fused", _p(q), _p(k_sets), _p(v_sets), _p(inverse), _p(lens), s3[0], s3[1], s3[2], float(scale),
                  B, U, L, C, _p(st[0]), st[1], st[2], None, _p(out), _p(attn), _stream())
        if need:
            st[1] = (st[1] + arrivals) & 0xffffffff
            st[2] += 1
        last_fp8_shared_launches = 1
        return out, attn
    last_fp8_shared_launches = 3 if need else 2
    # scales, codes and the row grouping in two launches (wsmg_attn_fp8_prep) instead of ~25 stock ones
    qc = torch.empty(B, C, device=dev, dtype=torch.uint8)
    kc = torch.empty(U, L, C, device=dev, dtype=torch.uint8)
    vc = torch.empty(U, L, C, device=dev, dtype=torch.uint8)
    sc = torch.empty(3, device=dev, dtype=torch.float32)
    order = torch.empty(B, device=dev, dtype=torch.int32)
    start = torch.empty(U + 1, device=dev, dtype=torch.int32)
    ws = torch.zeros(4, device=dev, dtype=torch.int32)
    _abi.call("wsmg_attn_fp8_prep", _p(q), _p(k_sets), _p(v_sets), _p(inverse), B, U, L, C, s3[0], s3[1], s3[2], _p(qc), _p(kc), _p(vc),
              _p(sc), _p(order), _p(start), _p(ws), _stream())
    qs, ks, vs = sc[0:1], sc[1:2], sc[2:3]
    out = torch.empty(B, C, device=dev, dtype=torch.float32)
    attn = torch.empty(B, L, device=dev, dtype=torch.float32)
    _abi.call("wsmg_attn_fp8_mfma_fwd", _p(qc), _p(qs), _p(kc), _p(ks), _p(vc), _p(vs), _p(lens), _p(order), _p(start), float(scale),
              B, U, L, C, _p(out), _p(attn), _stream())
    return out, attn
