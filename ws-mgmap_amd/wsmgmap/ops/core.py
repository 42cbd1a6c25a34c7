"""wsmgmap.ops.core — shared plumbing of the host-side operators:
argument checks, raw-pointer / stream helpers, live kernel timing (bench.py), section marks, the per-pass zero pool, the
end-of-backward stream joins and the token-gradient sink.

Host-side operators of the hot path: thin torch.autograd.Function wrappers around the
C ABI of libwsmgmap.so.  PyTorch is used only for device memory, streams and the autograd
tape; every FLOP of the three named operators runs in the hand-written gfx950 kernels.

All tensors are contiguous and resident on the GPU; anything else raises (there is no CPU or
eager fallback).  Activations are NHWC ([B,H,W,C]) stored float32 (parity mode: f32 MFMA, exact
float32 products) or bf16 (BASELINE configs[1]: bf16 MFMA, float32 accumulation); parameters,
statistics and weight gradients are always float32.
"""
import ctypes

import torch

from .. import _abi
from ..debug import sw

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
    return name.replace("_stats", "").replace("_slabs", "").replace("_ex", "")


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


def profile_set_only(only):
    """Change which entry points are timed WITHOUT dropping what has been collected (bench.py times a second family on a sample
    of its timed region's updates: every timed launch is two event records on the stream, ~1.5 us of GPU time each)."""
    global _prof_only
    if _prof is not None:
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


# ----------------------------------------------------------------------------- helper streams (round 4)
# A process has four hardware queues by default (GPU_MAX_HW_QUEUES) and HIP deals its streams onto them round-robin.  The update
# uses the caller's stream plus three helpers — "instruction" (instruction branch, weight layout, dense inputs, recurrent-core
# stages), "decoder" (the decoder's full-resolution branch, map_encoded_linear) and "early" (input preprocessing behind the
# producer's event) — and these are PROCESS-WIDE, one per device and name: a second policy object in the same process (bench.py's
# float32 leg behind its bf16 leg, an evaluation policy beside a training one) shares them instead of creating three more and
# landing two "independent" streams on one queue (the float32 leg ran 69 ms per update behind the bf16 leg, 56.5 alone).
_helper_streams = {}


def helper_stream(name, device=None, priority=0):
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    key = (dev, name)
    s = _helper_streams.get(key)
    if s is None:
        s = _helper_streams[key] = torch.cuda.Stream(device=dev, priority=priority)
    return s


def ranks_share_gpu():
    """True under a process group whose ranks of this node outnumber its visible GPUs (the launcher's LOCAL_WORLD_SIZE): the
    functional tests of the data-parallel path on a 1-GPU box.  Two processes' persistent RNN kernels and helper streams on ONE
    GPU oversubscribe its hardware queues (round 1: 49 ms -> 4.3 s per update with one extra stream per process; round 4, with
    the recurrent core pipelined over three streams: 36 ms -> 0.24-1.9 s): there the update runs on the caller's stream plus
    the instruction branch's, one persistent kernel at a time, as in round 3.  One process per GPU — the target's only
    configuration — is never affected."""
    import os
    if not (torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1):
        return False
    lws = int(os.environ.get("LOCAL_WORLD_SIZE", "0") or 0)
    return not (0 < lws <= torch.cuda.device_count())


# ----------------------------------------------------------------------------- input readiness (round 4)
# A tensor handed to BasePolicy.forward is, by PyTorch's stream rules, complete only in the ORDER of the caller's stream — behind the
# previous update's optimizer step.  The one host read-back of a forward pass (the number of distinct instructions) therefore made the
# host wait, every update, until the GPU had finished the update before: the host never ran more than a fraction of an update
# ahead, and any hiccup of a few milliseconds on the host (scheduler, page faults: 17-25 ms outliers on some boxes) reached the GPU.
# A producer that KNOWS when an input was complete — the feeder's collate stream, a trainer that built the batch long ago — says so
# here; the policy then runs the data-dependent, parameter-free part of its forward (the instruction dedup) on a stream that
# waits for THAT event only, and the read-back returns while the GPU is still busy with the previous update.
import weakref

_inputs_ready = {}      # id(tensor) -> (weak reference to the tensor, event): by identity (tensors compare element-wise)


def mark_inputs_ready(tensor, event=None):
    """`tensor` (an observation about to be passed to the policy) is complete once `event` has happened; default: an event
    recorded now on the current stream.  Returns the event."""
    if not tensor.is_cuda:
        return None
    if event is None:
        event = torch.cuda.Event()
        event.record(torch.cuda.current_stream(tensor.device))
    key = id(tensor)
    _inputs_ready[key] = (weakref.ref(tensor, lambda _r, key=key: _inputs_ready.pop(key, None)), event)
    return event


_inputs_dedup = {}      # id(instruction tensor) -> (weak reference, dedup tuple): see attach_instruction_dedup


def attach_instruction_dedup(tensor, dedup):
    """A producer that has the tokens on the host (DeviceCollator: plan_batch computes it in the decode worker) hands the policy the
    instruction dedup of `tensor` — (unique rows [U, L] int64, inverse [B] int64, lengths on the host, lengths on the device), what
    InstructionEncoder.dedup returns: the forward pass then runs neither the dedup kernel nor its host read-back.  Kept beside the
    tensor (by identity), not inside the observations dict, whose keys stay the reference's."""
    key = id(tensor)
    _inputs_dedup[key] = (weakref.ref(tensor, lambda _r, key=key: _inputs_dedup.pop(key, None)), dedup)


def attached_instruction_dedup(tensor):
    e = _inputs_dedup.get(id(tensor))
    return e[1] if e is not None and e[0]() is tensor else None


def inputs_ready_event(tensor):
    e = _inputs_ready.get(id(tensor))
    return e[1] if e is not None and e[0]() is tensor else None


# ----------------------------------------------------------------------------- state of one forward / backward pass
_prelaid = {}     # (data_ptr, version, cin_pad, dtype) -> (OHWI, IHWO) laid out by conv.prelayout_conv_weights for THIS forward pass


# Zero-initialised float32 scratch of ONE backward pass (the split-K accumulators of the weight-gradient kernels, the
# exactly-zero bias gradients): carved from one buffer per stream that is cleared by a single fill, instead of one
# 3.5-us fill launch per request (29 per update).  The first backward pass measures how much is needed; a callback at
# the end of every pass (autograd engine) retires the buffer, so the next pass starts from a fresh, cleared one.
_zero_pool = {}      # stream id -> dict(buf, off, used, cap, armed)


def _zero_pool_retire():
    for z in _zero_pool.values():
        z["cap"] = max(z["cap"], z["used"])
        z["buf"], z["off"], z["used"], z["armed"] = None, 0, 0, False


# True from a CHAINED recurrent core's forward pass (wsmgmap/recurrent.py) until its backward pass has queued its kernels: the
# gradient exchange holds its buckets meanwhile (wsmgmap/parallel.py::_launch_ready)
chain_in_flight = False


def reset_pass_state():
    """Called at the entry of every policy forward pass.  The end-of-backward callbacks that retire the zero pool and
    join the weight-gradient side stream do not run when backward() raises (a WsmgError from a kernel, OOM): without
    this reset one failed backward would leave them 'armed' for ever — every later request would fall back to a
    torch.zeros launch, and the optimizer could race gradients that a leaf stream is still writing."""
    global chain_in_flight
    chain_in_flight = False
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
