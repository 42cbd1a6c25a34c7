"""Host-side operators of the hot path: thin torch.autograd.Function wrappers around the
C ABI of libwsmgmap.so.  PyTorch is used only for device memory, streams and the autograd
tape; every FLOP of the three named operators runs in the hand-written gfx950 kernels.

All tensors are contiguous and resident on the GPU; anything else raises (there is no CPU or
eager fallback).  Activations are NHWC ([B,H,W,C]) stored float32 (parity mode: f32 MFMA, exact
float32 products) or bf16 (BASELINE configs[1]: bf16 MFMA, float32 accumulation); parameters,
statistics and weight gradients are always float32.
"""
import ctypes

import torch

from . import _abi
from .debug import sw

_ws_cache = {}


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


if hasattr(torch._C, "_cuda_getCurrentRawStream") and hasattr(torch._C, "_cuda_getDevice"):
    def _raw_stream():
        """The current HIP stream of the current device as an integer handle (two C calls: torch.cuda.current_stream() walks
        five Python frames per call, ≈100 times per rollout step — a fifth of the step's host time)."""
        return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
else:   # a torch build without the two entry points: the public route
    def _raw_stream():
        return torch.cuda.current_stream().cuda_stream


def _stream():
    return ctypes.c_void_p(_raw_stream())


def _req(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _abi.WsmgError("wsmgmap operators need GPU tensors: the HIP path is the only path (no CPU fallback)")
        if not t.is_contiguous():
            raise _abi.WsmgError(f"non-contiguous tensor passed to a wsmgmap kernel: shape {tuple(t.shape)} strides {t.stride()}")


def _f32(*ts):
    for t in ts:
        if t is not None and t.dtype != torch.float32:
            raise _abi.WsmgError(f"this wsmgmap argument must be float32, got {t.dtype}")


def _sfx(t):
    """C-ABI suffix for the storage type of an activation tensor (float32 or bf16)."""
    if t.dtype == torch.float32:
        return ""
    if t.dtype == torch.bfloat16:
        return "_bf16"
    raise _abi.WsmgError(f"wsmgmap activations are float32 or bfloat16, got {t.dtype}")


def _workspace(device):
    """float64 scratch for the chip-wide column reductions (4 MiB covers 1024 blocks x 2 x 256)."""
    # one buffer per STREAM: launches of one stream use it one after the other, but reductions on two streams (the decoder's
    # side-stream branch, the instruction branch's bias gradient) run at the same time
    key = (device.type, device.index, _raw_stream())
    ws = _ws_cache.get(key)
    if ws is None:
        ws = torch.empty(1024 * 2 * 256, dtype=torch.float64, device=device)
        _ws_cache[key] = ws
    return ws


def _rows_of(dy, C):
    """(tensor, row stride in elements) for a kernel that reads `dy` as rows of C channels: contiguous tensors as they are; a
    channel slice of a contiguous wider tensor — what autograd returns for the halves of a concatenation's gradient — in
    place, when rows stay 16-byte aligned; anything else through one contiguous copy."""
    if dy.is_contiguous():
        return dy, C
    ld = dy.stride(-2) if dy.dim() >= 2 else 0
    ok = (dy.stride(-1) == 1 and ld >= C and ld % 8 == 0 and dy.data_ptr() % 16 == 0 and C % 8 == 0
          and all(dy.stride(i) == dy.stride(i + 1) * dy.shape[i + 1] for i in range(dy.dim() - 2))
          and sw.strided_grads)
    return (dy, ld) if ok else (dy.contiguous(), C)


def _conv_out(h, k, s, p):
    return (h + 2 * p - k) // s + 1


# ----------------------------------------------------------------------------- live kernel timing
# bench.py brackets the conv-engine launches with HIP events on the launch stream (torch's
# current stream IS the stream handed to the C ABI) to report per-kernel roofline figures.
_prof = None

def _prof_key(name):
    """The entry points of one kernel family share a key: *_stats launch the same kernels with the statistics epilogue, *_slabs
    the same weight-gradient kernels with stores into slabs instead of atomics."""
    return name.replace("_stats", "").replace("_slabs", "")


KERNEL_OF = {  # C-ABI entry -> device kernel symbol (as rocprofv3 --kernel-trace names it)
    "wsmg_conv2d_fwd_bf16_stats": "conv_igemm_bf16_kernel<false, *>",
    "wsmg_conv2d_bwd_data_bf16_stats": "conv_igemm_bf16_kernel<true, *>",
    "wsmg_conv2d_fwd": "conv_igemm_kernel<false, false>",
    "wsmg_conv2d_bwd_data": "conv_igemm_kernel<true, false>",
    "wsmg_conv2d_bwd_weight": "conv_wgrad_kernel<false>",
    "wsmg_conv2d_fwd_bf16": "conv_igemm_bf16_kernel<false, *>",
    "wsmg_conv2d_fwd_bf16_splitk": "conv_igemm_bf16_kernel<false, 64, *, 1, true>",
    "wsmg_conv2d_bwd_data_bf16": "conv_igemm_bf16_kernel<true, *>",
    "wsmg_conv2d_bwd_weight_bf16": "conv_wgrad_bf16_kernel",
}


_prof_only = None


def profile_begin(only=None):
    """Start bracketing conv-engine launches with HIP events.  only: set of C-ABI entry names to time (None = all six
    conv entry points); every timed launch costs two event records on the host, so bench.py times only the dominant
    kernel family inside its timed region and learns which one that is during the warm-up updates."""
    global _prof, _prof_only
    _prof = {}
    _prof_only = set(only) if only else None


def profile_end():
    """-> {kernel: dict(launches, ms_total, flops_total)}; synchronises."""
    global _prof, _prof_only
    rec, _prof, _prof_only = _prof, None, None
    torch.cuda.synchronize()
    out = {}
    for name, items in (rec or {}).items():
        ms = sum(s.elapsed_time(e) for s, e, _, _ in items)
        o = out.setdefault(KERNEL_OF[_prof_key(name)], dict(launches=0, ms_total=0.0, flops_total=0.0, entry=_prof_key(name)))
        o["launches"] += sum(n for _, _, _, n in items)      # (the *_stats entry points launch the same kernels: one family)
        o["ms_total"] += ms
        o["flops_total"] += float(sum(f for _, _, f, _ in items))
    return out


def _launch(name, flops, *args, prof_as=None):
    """prof_as: time this launch WITH the family of that entry point, as part of its launches (the ordered slab reduction behind a
    weight-gradient kernel: its time belongs to the family's total, it is not a launch of the family's kernel)."""
    key = prof_as or name
    if _prof is None or (_prof_only is not None and _prof_key(key) not in _prof_only):
        _abi.call(name, *args)
        return
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    _abi.call(name, *args)
    e.record()
    _prof.setdefault(key, []).append((s, e, flops, 0 if prof_as else 1))


# ----------------------------------------------------------------------------- section marks (diagnostics)
# tools/section_times.py: HIP events at the stage boundaries of one update (forward marks from MGMapNet.forward, backward
# marks from gradient hooks on the boundary tensors), to see where the critical path of an update goes without a profiler
# attached.  Off (None) in every product run.
_marks = None


def marks_begin():
    global _marks
    _marks = []


def marks_end():
    global _marks
    out, _marks = _marks, None
    return out


def mark(name, tensor=None):
    """Record an event now (forward); with `tensor`, also when its gradient arrives (backward)."""
    if _marks is None:
        return
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    _marks.append(("f:" + name, e))
    if tensor is not None and tensor.requires_grad:
        def hook(g, name=name):
            if _marks is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                _marks.append(("b:" + name, ev))
            return g
        tensor.register_hook(hook)


# ----------------------------------------------------------------------------- convolution
_prelaid = {}     # (data_ptr, version, cin_pad, dtype) -> (OHWI, IHWO) laid out by prelayout_conv_weights for THIS forward pass


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


# Zero-initialised float32 scratch of ONE backward pass (the split-K accumulators of the weight-gradient kernels, the
# exactly-zero bias gradients): carved from one buffer per stream that is cleared by a single fill, instead of one
# 3.5-us fill launch per request (29 per update).  The first backward pass measures how much is needed; a callback at
# the end of every pass (autograd engine) retires the buffer, so the next pass starts from a fresh, cleared one.
_zero_pool = {}      # stream id -> dict(buf, off, used, cap, armed)


def _zero_pool_retire():
    for z in _zero_pool.values():
        z["cap"] = max(z["cap"], z["used"])
        z["buf"], z["off"], z["used"], z["armed"] = None, 0, 0, False


def reset_pass_state():
    """Called at the entry of every policy forward pass.  The end-of-backward callbacks that retire the zero pool and
    join the weight-gradient side stream do not run when backward() raises (a WsmgError from a kernel, OOM): without
    this reset one failed backward would leave them 'armed' for ever — every later request would fall back to a
    torch.zeros launch, and the optimizer could race gradients that a leaf stream is still writing."""
    TokenGradSink.check_none_pending()
    _prelaid.clear()
    if any(z["armed"] for z in _zero_pool.values()):
        _zero_pool_retire()
    if _side_join_armed:
        for main_id, side_id in list(_side_join_armed):
            for side in _leaf_streams:
                if side.cuda_stream == side_id:
                    torch.cuda.current_stream().wait_stream(side)
        _side_join_armed.clear()


def _zeros_f32(shape, device):
    numel = 1
    for d in shape:
        numel *= int(d)
    n = (numel + 63) // 64 * 64   # 256-byte granules
    engine = torch.autograd.Variable._execution_engine
    z = _zero_pool.setdefault(_raw_stream(), dict(buf=None, off=0, used=0, cap=0, armed=False))
    z["used"] += n
    if not z["armed"]:
        try:
            engine.queue_callback(_zero_pool_retire)   # only legal while a backward pass is running
            z["armed"] = True
        except RuntimeError:
            z["used"] -= n
            return torch.zeros(shape, device=device, dtype=torch.float32)
    if z["buf"] is None and z["cap"] >= n:
        z["buf"], z["off"] = torch.zeros(z["cap"], device=device, dtype=torch.float32), 0
    if z["buf"] is None or z["off"] + n > z["buf"].numel() or z["buf"].device != device:
        return torch.zeros(shape, device=device, dtype=torch.float32)
    out = z["buf"][z["off"]:z["off"] + numel].view(shape)
    z["off"] += n
    return out


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


_side_join_armed = set()
_leaf_streams = []     # streams that join the main stream at the end of a backward pass (wsmgmap.recurrent's leaf stream)


def _join_side_at_end(main, side, strict=False):
    """main waits for side once, when the running backward pass ends.  strict: outside a backward pass raise (the caller
    then does not use the side stream at all) instead of joining now."""
    key = (main.cuda_stream, side.cuda_stream)
    if key in _side_join_armed:
        return

    def join():
        # (final callbacks run on the stream that called backward(), AFTER the engine has joined the streams of the backward
        #  nodes into it: if `main` is one of those — the decoder's side branch — waiting on `main` alone would come too late)
        _side_join_armed.discard(key)
        main.wait_stream(side)
        cur = torch.cuda.current_stream(main.device)
        if cur.cuda_stream != main.cuda_stream:
            cur.wait_stream(side)
    try:
        torch.autograd.Variable._execution_engine.queue_callback(join)
        _side_join_armed.add(key)
    except RuntimeError:      # not inside a backward pass: join now
        if strict:
            raise
        main.wait_stream(side)


class TokenGradSink:
    """Links the backward passes around the map tokens (mg_map_policy.py:99-100,217,233-235): the tokens are the output of a
    convolution with a fused ReLU and feed (a) their token mean and (b) the map attention.  Autograd would add the two
    gradients (materialising the mean's broadcast) and the convolution would then mask the sum with a third pass.  With a
    sink, the attention's backward PARKS its gradient here and returns nothing; the token mean's backward — which data
    dependence puts later (its gradient comes through GRU 1, which needs the attention's query gradient first) — merges its
    broadcast row into the parked tensor and applies the ReLU mask in the same pass (wsmg_token_grad_merge), and the
    convolution sees `masked` and skips its own mask.  One object per forward pass."""
    _pending = []

    def __init__(self):
        self.dx = None        # the attention's gradient of the tokens, parked
        self.relu = False     # set by the producing convolution: the tokens are relu(conv)
        self.masked = False   # set by the merge: the gradient handed to the convolution is already masked

    def park(self, dx):
        self.dx = dx
        TokenGradSink._pending.append(self)

    def take(self):
        dx, self.dx = self.dx, None
        if self in TokenGradSink._pending:
            TokenGradSink._pending.remove(self)
        return dx

    @staticmethod
    def check_none_pending():
        if TokenGradSink._pending:
            TokenGradSink._pending.clear()
            raise _abi.WsmgError("a map-token gradient parked by the attention's backward was never merged (the token mean's "
                                 "backward did not run): gradients of the previous pass are incomplete")


class _Conv2d(torch.autograd.Function):
    """y = conv2d(x, w) + b on NHWC x; w is the reference's OIHW float32 parameter (its .grad comes back OIHW
    float32).  x float32 -> f32 MFMA engine; x bf16 -> bf16 operands, float32 accumulation and dW.  If x has more
    channels than w (the engine pads activations to multiples of 32) the weight is zero-padded to match."""

    @staticmethod
    def forward(ctx, x, w_oihw, bias, stride, pad, bias_grad_zero=False, relu=False, stats=None, relu_sink=None):
        _req(x, w_oihw, bias)
        _f32(w_oihw, bias)
        sfx = _sfx(x)
        B, H, W, Cin = x.shape
        Cout, Cin_w, KH, KW = w_oihw.shape
        assert Cin >= Cin_w, (x.shape, w_oihw.shape)
        w, w_ihwo = _weight_layouts(w_oihw.contiguous(), Cin, x.dtype, ctx.needs_input_grad[0])
        OH, OW = _conv_out(H, KH, stride, pad), _conv_out(W, KW, stride, pad)
        y = torch.empty(B, OH, OW, Cout, device=x.device, dtype=x.dtype)
        fl = 2.0 * B * OH * OW * Cout * Cin * KH * KW
        dims = (B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW)
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
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w_ihwo, y_relu = ctx.saved_tensors
        *dims, has_bias, sfx, Cin_w = ctx.cfg
        B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW = dims
        if ctx.relu_sink is not None and ctx.relu_sink.masked:
            y_relu = None        # the token-gradient merge applied the ReLU mask already (TokenGradSink)
        if y_relu is not None:   # fused ReLU: mask the incoming gradient with the saved output first
            masked = torch.empty_like(y_relu)
            dy, ld = _rows_of(dy, Cout) if sfx else (dy.contiguous(), Cout)
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
        if has_bias and ctx.needs_input_grad[2]:
            # a bias in front of a train-mode BatchNorm cancels in (x - mean): its gradient is sum(dy) = 0 exactly;
            # the caller says so and the channel reduction over dy is skipped
            db = _zeros_f32((Cout,), dy.device) if ctx.bias_grad_zero else channel_sum(dy.view(-1, Cout))
        return dx, dw, db, None, None, None, None, None, None


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


def conv2d(x, weight_oihw, bias, stride=1, pad=0, bias_grad_zero=False, relu=False, stats=None, relu_sink=None):
    """x NHWC; weight = the reference's OIHW parameter (laid out for the engine in one launch inside the autograd
    node, so the parameter's .grad comes back OIHW float32).  bias_grad_zero: the output feeds a train-mode
    BatchNorm, so d(loss)/d(bias) is identically zero and is returned as zeros.  relu: y = relu(conv + bias), fused into
    the conv epilogue in bf16 mode."""
    return _Conv2d.apply(x, weight_oihw, bias, stride, pad, bias_grad_zero, relu, stats, relu_sink)


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


def conv_transpose2d(x, weight_iohw, stride=2, pad=1, stats=None):
    """nn.ConvTranspose2d weight is [Cin_t, Cout_t, KH, KW] = the adjoint conv's OIHW."""
    return _ConvT2d.apply(x, weight_iohw, stride, pad, stats)


def channel_sum(x2d):
    _req(x2d)
    rows, C = x2d.shape
    out = torch.empty(C, device=x2d.device, dtype=torch.float32)
    ws = _workspace(x2d.device)
    _abi.call("wsmg_channel_sum" + _sfx(x2d), _p(x2d), rows, C, _p(out), _p(ws), ws.numel() * 8, _stream())
    return out


# ----------------------------------------------------------------------------- batch norm (+res)(+relu)
class _BnAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, gamma, beta, running_mean, running_var, train, relu, momentum, eps, stats=None):
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
        if ld != C:     # a channel slice of a concatenation's gradient, read in place
            _abi.call("wsmg_bn_act_bwd_ld" + sfx, _p(dy), ld, _p(x), _p(y), _p(gamma), _p(beta), _p(mean), _p(invstd), relu, rows, C,
                      _p(dx), _p(dres), _p(dgamma), _p(dbeta), _p(ws), ws.numel() * 8, _stream())
        else:
            _abi.call("wsmg_bn_act_bwd" + sfx, _p(dy), _p(x), _p(y), _p(gamma), _p(beta), _p(mean), _p(invstd), relu, rows, C,
                      _p(dx), _p(dres), _p(dgamma), _p(dbeta), _p(ws), ws.numel() * 8, _stream())
        return dx, dres, dgamma, dbeta, None, None, None, None, None, None, None


def bn_act(x, gamma, beta, running_mean, running_var, train, relu=True, residual=None, momentum=0.1, eps=1e-5, stats=None):
    return _BnAct.apply(x, residual, gamma, beta, running_mean, running_var, train, relu, momentum, eps, stats)


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
    (nn.Conv2d(32, classes, 1)) + the prediction monitor's per-sample cross-entropy against gt [B,Hg,Wg] + AvgPool2d(2)."""
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


def _fp8_forward(q, w2d, x_q, xs_t, lengths, scale):
    B, L, C = x_q.shape
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
    lens = None if lengths is None else lengths.to(torch.int32).contiguous()
    out = torch.empty(B, C, device=dev, dtype=torch.float32)
    attn = torch.empty(B, L, device=dev, dtype=torch.float32)
    _abi.call("wsmg_attn_fp8_mfma_fwd", _p(qc), _p(qs), _p(kc), _p(ks), _p(vc), _p(vs), _p(lens), _p(order), _p(start), float(scale),
              B, U, L, C, _p(out), _p(attn), _stream())
    return out, attn


def _rnn_workspace(nbytes, device):
    """Barrier words + exchange image of the persistent RNN kernels.  debug.sw.rnn_poison (stress tool) fills
    it with NaN first so that any stale or missed hand-off read poisons the results visibly."""
    ws = torch.empty((int(nbytes) + 3) // 4, device=device, dtype=torch.float32)
    if sw.rnn_poison:
        ws.fill_(float("nan"))
    return ws


check_rnn_status = _abi.check_rnn_status


def _rnn_launched():
    if sw.rnn_check:          # debug: synchronise and check after every persistent launch
        torch.cuda.current_stream().synchronize()
        _abi.check_rnn_status()


class _MaskedGRU(torch.autograd.Function):
    """gi [T,N,3H] (input projections), w_hh [3H,H], b_hh [3H], h0 [N,H], masks [T,N] -> y [T,N,H]."""

    @staticmethod
    def forward(ctx, gi, w_hh, b_hh, h0, masks):
        _req(gi, w_hh, b_hh, h0, masks)
        _f32(gi, w_hh, b_hh, h0, masks)
        T, N, H3 = gi.shape
        H = H3 // 3
        dev = gi.device
        y = torch.empty(T, N, H, device=dev, dtype=torch.float32)
        saves = [torch.empty(T, N, H, device=dev, dtype=torch.float32) for _ in range(4)]
        sync = _rnn_workspace(_abi.lib().wsmg_gru_workspace_bytes(T), dev)
        _abi.call("wsmg_gru_fwd", _p(gi), _p(w_hh), _p(b_hh), _p(h0), _p(masks), T, N, H, _p(y),
                  *[_p(s) for s in saves], _p(sync), _stream())
        _rnn_launched()
        ctx.save_for_backward(w_hh, h0, masks, y, *saves)
        return y

    @staticmethod
    def backward(ctx, dy):
        w_hh, h0, masks, y, sr, sz, sn, sghn = ctx.saved_tensors
        T, N, H = y.shape
        dev = y.device
        dy = dy.contiguous()
        dgi = torch.empty(T, N, 3 * H, device=dev, dtype=torch.float32)
        dgh = torch.empty(T, N, 3 * H, device=dev, dtype=torch.float32)
        dh0 = torch.empty(N, H, device=dev, dtype=torch.float32)
        sync = _rnn_workspace(_abi.lib().wsmg_gru_workspace_bytes(T), dev)
        _abi.call("wsmg_gru_bwd", _p(dy), None, _p(w_hh), _p(h0), _p(masks), _p(y), _p(sr), _p(sz), _p(sn), _p(sghn),
                  T, N, H, _p(dgi), _p(dgh), _p(dh0), _p(sync), _stream())
        _rnn_launched()
        hprev = torch.cat([h0.unsqueeze(0), y[:-1]], dim=0) * masks.unsqueeze(-1)
        g2 = dgh.view(T * N, 3 * H)
        dw_hh = g2.t() @ hprev.view(T * N, H)
        db_hh = g2.sum(dim=0)
        return dgi, dw_hh, db_hh, dh0, None


def masked_gru(gi, w_hh, b_hh, h0, masks):
    """Whole-sequence masked GRU in one persistent launch.  Returns y [T,N,H]; final state = y[-1]."""
    return _MaskedGRU.apply(gi.contiguous(), w_hh.contiguous(), b_hh.contiguous(), h0.contiguous(), masks.contiguous())


# ----------------------------------------------------------------------------- persistent packed bi-LSTM
class _BiLSTM(torch.autograd.Function):
    """gi [U,L,2,4H], w_hh [2,4H,H], b_hh [2,4H], lengths int32 [U] -> out [U,L,2H]."""

    @staticmethod
    def forward(ctx, gi, w_hh, b_hh, lengths):
        _req(gi, w_hh, b_hh, lengths)
        _f32(gi, w_hh, b_hh)
        U, L, _, H4 = gi.shape
        H = H4 // 4
        dev = gi.device
        out = torch.empty(U, L, 2 * H, device=dev, dtype=torch.float32)
        sg = torch.zeros(2, U, L, 4, H, device=dev, dtype=torch.float32)
        sc = torch.zeros(2, U, L, H, device=dev, dtype=torch.float32)
        ws = _rnn_workspace(_abi.lib().wsmg_lstm_workspace_bytes(L), dev)
        _abi.call("wsmg_lstm_fwd", _p(gi), _p(w_hh), _p(b_hh), _p(lengths), U, L, H, _p(out), _p(sg), _p(sc), _p(ws), _stream())
        _rnn_launched()
        ctx.save_for_backward(w_hh, lengths, out, sg, sc)
        return out

    @staticmethod
    def backward(ctx, dout):
        w_hh, lengths, out, sg, sc = ctx.saved_tensors
        U, L, H2 = out.shape
        H = H2 // 2
        dev = out.device
        dout = dout.contiguous()
        dg = torch.empty(U, L, 2, 4 * H, device=dev, dtype=torch.float32)
        ws = _rnn_workspace(_abi.lib().wsmg_lstm_workspace_bytes(L), dev)
        _abi.call("wsmg_lstm_bwd", _p(dout), _p(w_hh), _p(lengths), _p(sg), _p(sc), U, L, H, _p(dg), _p(ws), _stream())
        _rnn_launched()
        zero = torch.zeros(U, 1, H, device=dev, dtype=torch.float32)
        hprev_f = torch.cat([zero, out[:, :-1, :H]], dim=1)       # state before step t (forward direction)
        hprev_r = torch.cat([out[:, 1:, H:], zero], dim=1)        # state before step t (reverse direction)
        # one direction's gate gradients as a contiguous [U L, 4H] matrix first: on the strided view dg[:, :, d] (row pitch 8H)
        # the GEMM library picked a 32 x 16 tile kernel that took 340 us for this 0.7 GFLOP product (beside the map stack's
        # backward, on the instruction stream)
        dgd = dg.permute(2, 0, 1, 3).contiguous().view(2, U * L, 4 * H)
        dw = torch.stack([dgd[0].t() @ hprev_f.reshape(U * L, H), dgd[1].t() @ hprev_r.reshape(U * L, H)])
        db = dg.sum(dim=(0, 1))
        return dg, dw, db, None


def bilstm(gi, w_hh, b_hh, lengths):
    """Packed bidirectional LSTM over <= 8 sequences in one persistent launch."""
    return _BiLSTM.apply(gi.contiguous(), w_hh.contiguous(), b_hh.contiguous(), lengths.contiguous())


# ----------------------------------------------------------------------------- BEV (no autograd: rollout only)
@torch.no_grad()
def bev_index(depth, Hf, Wf, E, depth_scale=10.0, local_scale=0.12):
    """depth [B,Hd,Wd] -> lin_idx int32 [B,Hf*Wf] (-1 = invalid source)."""
    _req(depth)
    _f32(depth)
    B, Hd, Wd = depth.shape
    lin = torch.empty(B, Hf * Wf, device=depth.device, dtype=torch.int32)
    _abi.call("wsmg_bev_index", _p(depth), B, Hd, Wd, float(depth_scale), Hf, Wf, E, float(local_scale), _p(lin), _stream())
    return lin


@torch.no_grad()
def bev_scatter_max(feat, lin, C, E):
    """feat [B,Cf,Hf,Wf] NCHW -> [B,C,E,E] NCHW planes."""
    _req(feat, lin)
    _f32(feat)
    B, Cf, Hf, Wf = feat.shape
    out = torch.empty(B, C, E, E, device=feat.device, dtype=torch.float32)
    _abi.call("wsmg_bev_scatter_max", _p(feat), _p(lin), B, Cf, Hf, Wf, C, E, _p(out), _stream())
    return out


@torch.no_grad()
def bev_rotate(planes, heading, sign):
    """planes [B,C,E,E] -> rotated NHWC [B,E,E,C]."""
    _req(planes, heading)
    B, C, E, _ = planes.shape
    out = torch.empty(B, E, E, C, device=planes.device, dtype=torch.float32)
    _abi.call("wsmg_bev_rotate", _p(planes), _p(heading), float(sign), B, C, E, _p(out), _stream())
    return out


@torch.no_grad()
def bev_scatter_rotate(feat, lin, heading, sign, C, E):
    """bev_scatter_max + bev_rotate in one launch; the rotated map stays in NCHW planes [B,C,E,E] (for map_fuse(..., planes=True))."""
    _req(feat, lin, heading)
    _f32(feat, heading)
    B, Cf, Hf, Wf = feat.shape
    out = torch.empty(B, C, E, E, device=feat.device, dtype=torch.float32)
    _abi.call("wsmg_bev_scatter_rotate", _p(feat), _p(lin), _p(heading), float(sign), B, Cf, Hf, Wf, C, E, _p(out), _stream())
    return out


def bev_planes_ok(C, E):
    """Shapes the one-launch scatter + rotation and the plane-consuming fuse take."""
    return C % 4 == 0 and C <= 64 and E > 1 and E * E * 4 <= 160 * 1024


@torch.no_grad()
def _check_global_map(global_map, B, C, *f32s):
    """The kernels index global_map[b] for b < B: the reference slices `full_global_map[:bs]` (rgb_mapping.py:43) and
    fails with a shape error when the batch has more rows than num_proc — here that would be an out-of-bounds access."""
    _f32(global_map, *f32s)
    if global_map.dim() != 4 or global_map.shape[1] != global_map.shape[2]:
        raise _abi.WsmgError(f"full_global_map must be [num_proc, G, G, C], got {tuple(global_map.shape)}")
    if global_map.shape[0] < B:
        raise _abi.WsmgError(f"batch of {B} rows but full_global_map holds {global_map.shape[0]} maps (num_proc): "
                             "construct the policy with RGBMAPPING.num_proc >= the rollout batch")
    if global_map.shape[3] != C:
        raise _abi.WsmgError(f"full_global_map has {global_map.shape[3]} channels, the ego map {C}")


def map_fuse(ego_rot, global_map, gps, masks, resolution=0.12, planes=False):
    """planes: ego_rot is [B,C,E,E] (bev_scatter_rotate's output) instead of NHWC [B,E,E,C]; same result bit for bit."""
    _req(ego_rot, global_map, gps, masks)
    if planes:
        B, C, E, _ = ego_rot.shape
    else:
        B, E, _, C = ego_rot.shape
    _check_global_map(global_map, B, C, ego_rot, gps, masks)
    if gps.shape[0] != B or masks.numel() != B:
        raise _abi.WsmgError("map_fuse: gps [B,2] and masks [B] must match the ego maps' batch")
    G = global_map.shape[1]
    _abi.call("wsmg_map_fuse_planes" if planes else "wsmg_map_fuse", _p(ego_rot), _p(global_map), _p(gps), _p(masks), B, C, E, G,
              float(resolution), _stream())


@torch.no_grad()
def map_retrieve(global_map, gps, compass, E, resolution=0.12, fused=None):
    """fused: crop + rotation in one launch, bit-identical to the two.  Default: for small batches only (B E^2 C / 4 <= 500 000
    work items: 11 vs 16 us at B = 1; at B = 8 the 16 gathers per item already cost more than the crop's round trip through memory —
    35 vs 26 us, 311 vs 219 us at cfg4); `fused` = True / False forces either."""
    _req(global_map, gps, compass)
    B = gps.shape[0]
    _check_global_map(global_map, B, global_map.shape[3] if global_map.dim() == 4 else -1, gps, compass)
    if compass.numel() != B:
        raise _abi.WsmgError("map_retrieve: compass [B] must match gps [B,2]")
    G, C = global_map.shape[1], global_map.shape[3]
    out = torch.empty(B, E, E, C, device=gps.device, dtype=torch.float32)
    if fused is None:
        fused = B * E * E * (C // 4) <= 500_000
    if fused:
        _abi.call("wsmg_map_retrieve_fused", _p(global_map), _p(gps), _p(compass), B, C, E, G, float(resolution), _p(out), _stream())
        return out
    scratch = torch.empty(B, E, E, C, device=gps.device, dtype=torch.float32)
    _abi.call("wsmg_map_retrieve", _p(global_map), _p(gps), _p(compass), B, C, E, G, float(resolution), _p(scratch), _p(out), _stream())
    return out
