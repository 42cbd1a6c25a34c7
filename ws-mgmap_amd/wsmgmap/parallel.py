"""Data-parallel gradient exchange for the teacher-forcing / DAgger update: one process per
GPU, one averaged all-reduce of the LIVE gradients per update over RCCL/xGMI (backend "nccl"),
bucketed and launched from autograd hooks so it overlaps the rest of backward.  Per update and bucket: one
multi-tensor pack, one all-reduce, one scale; the averaged gradients stay in the bucket (`p.grad` becomes a view of it).

Replaces `DistributedDataParallel(find_unused_parameters=True)` at the reference's
common_trainer.py:61-66 (stock DDP around `BasePolicy` also works — tests/test_gpu_policy.py wraps it — but
all-reduces all 19.76 M trainable floats (79 MB) although 11.5 M of them never receive a gradient (unused
resnet18 layers 2-4 / fc, critic, logstd); here the first backward discovers which parameters are live
(8.23 M, 32.9 MB) and only those are exchanged, in buckets filled in gradient-ready order (xGMI ring
all-reduce is per-link bound, so a few ~8 MB buckets keep every link busy while backward continues).

What DDP guarantees at construction — every rank reduces the same tensors in the same order
(common_trainer.py:60-66 → torch's `_verify_param_shape_across_processes`) — is established here after the
discovery pass: rank 0's gradient-ready order is broadcast and becomes EVERY rank's bucket layout, and a 64-bit
digest of each rank's live SET (parameter index, numel, dtype) is all-reduced (MIN and MAX): ranks whose backward
passes touched different parameters raise `GradExchangeError` together instead of averaging unrelated tensors.
Buckets are issued strictly in index order on every rank (a bucket that is ready waits for its predecessors),
also on the error path, because NCCL/RCCL and gloo pair collectives by issue order.

BatchNorm statistics stay per rank (no SyncBN), like the reference; `broadcast_buffers()` gives every rank
rank 0's buffers (what DDP's per-forward buffer broadcast, C4 in SURVEY.md, amounts to at checkpoint time).

Errors are agreed on across ranks BEFORE the optimizer can use the gradients: a rank that sees a local problem (a
gradient outside the discovered live set, a live parameter without a gradient, a persistent-RNN timeout) still
takes part in every collective of the update — so no peer blocks in an all-reduce — and raises a flag that one
4-byte all-reduce sums inside `finish()`; the averaged gradients of ALL ranks are then replaced by zeros on the
device (`masked_fill_(flag, 0)`: no host synchronisation), so the `optimizer.step()` that follows applies the same
(momentum-only, finite) step everywhere and the ranks' parameters stay identical.  Every rank reads the summed
flag at the start of the NEXT `finish()` (by then the copy to pinned memory is long complete: no stall) — or in
this one with `finish(check_now=True)`, at the price of a host synchronisation per update — and all of them raise
together, after resetting to a fresh discovery pass.  `resync()` re-broadcasts rank 0's parameters, buffers and
optimizer state for callers that want to continue after such an error.

Works with any torch.distributed backend (gloo on CPU for tests).
"""
import time
import zlib

import torch
import torch.distributed as dist


class GradExchangeError(RuntimeError):
    pass


class GradAllReducer:
    def __init__(self, params, bucket_bytes=8 << 20, group=None, single_rank_exchange=False, module=None, broadcast_buffers_every=0,
                 exchange_stream=None):
        """single_rank_exchange: run the whole exchange even in a one-rank group (by default one rank does nothing at all) — the
        way to put the real backend (RCCL) through this code on a box with one GPU (tests/test_gpu_round2.py).
        module, broadcast_buffers_every = n > 0: every n-th `finish()` ends with rank 0's buffers (BatchNorm running statistics,
        counters) broadcast to every rank, two coalesced collectives — opt-in parity with DDP's `broadcast_buffers=True`, which
        re-broadcasts rank 0's buffers before EVERY forward (common_trainer.py:61-66): n = 1 gives a trainer that
        evaluates or checkpoints from any rank mid-epoch the statistics the reference's DDP would show it.  Default 0: the
        statistics stay per rank between explicit `broadcast_buffers()` calls."""
        # exchange_stream (round 5): a CUDA stream THE CALLER ALREADY USES (the policy's instruction-branch stream, idle for most
        # of the backward pass: ops.helper_stream("instruction")), or the string naming it.  The buckets are then packed AND
        # all-reduced on that stream with synchronous collectives (`async_op=False`: this torch runs those on the caller's
        # current stream, without ProcessGroupNCCL's internal stream), the compute streams never wait for a bucket — only
        # `finish()` joins the exchange stream into the caller's — and the process has no stream beyond the policy's own.
        # None: the round-2 form (pack on the stream of the hook that completes a bucket, asynchronous collective on the
        # backend's internal stream).  DESIGN.md section 4.3 / 5.
        self._xstream = exchange_stream
        self._restore = []          # process-wide settings this reducer changed: close() puts them back
        if exchange_stream is not None:
            # with no stream beyond the policy's own in the process, the policy may keep its early instruction dedup under a process
            # group too (mg_map_policy.MGMapNet._encode_instruction; measured: profiles/r05_dp_exchange_ab.txt)
            from . import debug
            self._restore.append((debug.sw, "early_dedup_dp", debug.sw.early_dedup_dp))
            debug.sw.early_dedup_dp = True
        self._module = module
        self._buffers_every = int(broadcast_buffers_every)
        if self._buffers_every > 0 and module is None:
            raise ValueError("broadcast_buffers_every needs the module whose buffers are to be broadcast")
        self._updates = 0
        self.params = [p for p in params if p.requires_grad]
        seen, uniq = set(), []
        for p in self.params:
            if id(p) not in seen:
                seen.add(id(p))
                uniq.append(p)
        self.params = uniq
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self.bucket_bytes = bucket_bytes
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._off = self.world == 1 and not (single_rank_exchange and dist.is_initialized())
        self._order = []          # discovery pass: params in gradient-ready order
        self._buckets = None      # list of dicts: params, flat buffer, pending count
        self._where = {}          # id(param) -> (bucket index, offset)
        self._works = []
        self._seen = set()        # parameters whose gradient-ready hook ran in the running update
        self._next = 0            # buckets are issued in index order: the next one to launch
        self._local_error = None  # first local problem of the running update
        self._flag = None         # device int32 [1]: this rank's error flag, summed over the ranks
        self._flag_host = None    # pinned copy of the summed flag of the previous update (+ event)
        self._flag_event = None
        self._flag_pending = False
        self._prev_error = None
        self._timing = []         # per update: (host seconds inside finish(), event at finish() entry, event after the last wait)
        self._stats = dict(updates=0, host_wait_ms=0.0, exposed_ms=0.0, exposed_max_ms=0.0)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        for p in self.params:
            p._wsmg_reducer = True      # wsmgmap.recurrent: this parameter's gradient-ready hook is ours (it follows _wsmg_grad_stream)
        self._bucket_events = None  # time_buckets(): per update [(bucket, event before, event after the collective)]
        self._bucket_ms = {}
        if not self._off and self.params and self.params[0].is_cuda:
            # a status check that raises on ONE rank in the middle of a forward pass would leave the peers blocked in their collectives:
            # while this reducer lives, only finish() reads the persistent kernels' status word, and the ranks agree on what it said
            from . import _abi
            self._restore.append((_abi, "defer_rnn_status", _abi.defer_rnn_status))
            _abi.defer_rnn_status = True

    def close(self):
        """Remove the gradient hooks and put back the process-wide settings the constructor changed (the deferred status checks, the
        early dedup under a process group): a dropped reducer must not leave them behind."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for p in self.params:
            if hasattr(p, "_wsmg_reducer"):
                del p._wsmg_reducer
        for obj, name, val in reversed(self._restore):
            setattr(obj, name, val)
        self._restore = []
        self._off = True

    def time_buckets(self, on=True):
        """Bracket every bucket's pack + collective with two events on the stream it runs on (exchange_stream forms only);
        `stats()["per_bucket_allreduce_ms"]` then lists the mean per bucket — what a ring step over xGMI takes, peers' skew included."""
        self._bucket_events = [] if on else None
        self._bucket_ms = {}

    # -- setup ---------------------------------------------------------------------
    def broadcast_parameters(self, module, src=0):
        """rank-`src` parameters and buffers to every rank (DDP constructor behaviour)."""
        if self._off:
            return
        for t in list(module.parameters()) + [b for b in module.buffers() if b.is_floating_point() or b.dtype == torch.int64]:
            dist.broadcast(t.data, src=src, group=self.group)
            torch.autograd.graph.increment_version(t)

    def broadcast_buffers(self, module, src=0):
        """rank-`src` buffers (BatchNorm running statistics, counters) to every rank: the state DDP's
        `broadcast_buffers=True` keeps identical by re-broadcasting before every forward (common_trainer.py:61-66
        default).  Statistics here stay per rank during training; call this before evaluating or checkpointing from a
        rank other than `src` (the reference checkpoints rank 0's, common_trainer.py:99), or construct the reducer with
        `module=..., broadcast_buffers_every=n`.  One collective per dtype: the buffers travel packed."""
        if self._off:
            return
        by_dtype = {}
        for b in module.buffers():
            if b.is_floating_point() or b.dtype == torch.int64:
                by_dtype.setdefault((b.dtype, b.device), []).append(b)
        g_src = dist.get_global_rank(self.group, src) if self.group is not None else src
        for bufs in by_dtype.values():
            flat = torch.cat([b.data.reshape(-1) for b in bufs])
            dist.broadcast(flat, src=g_src, group=self.group)
            off = 0
            views = []
            for b in bufs:
                views.append(flat[off:off + b.numel()].view_as(b))
                off += b.numel()
            torch._foreach_copy_([b.data for b in bufs], views)
            for b in bufs:
                torch.autograd.graph.increment_version(b)

    def resync(self, module, optimizer=None, src=0):
        """After a GradExchangeError: rank-`src` parameters, buffers and (if given) optimizer state tensors to every rank, so
        training can continue from one consistent state (a rank whose forward pass produced NaN — a persistent-RNN timeout —
        may hold NaN BatchNorm statistics; the zeroed gradients kept its parameters in step with the others)."""
        self.broadcast_parameters(module, src)
        if optimizer is not None and self.world > 1:
            for group in optimizer.param_groups:
                for p in group["params"]:
                    st = optimizer.state.get(p)
                    if not st:
                        continue
                    for k in sorted(st):
                        v = st[k]
                        if torch.is_tensor(v) and v.device == p.device:
                            dist.broadcast(v, src=src, group=self.group)

    def _agree_on_layout(self):
        """After the discovery pass: every rank must have found the same live SET; the ORDER (bucket layout) is rank 0's."""
        dev = self.params[0].device
        items = sorted((self._index[id(p)], p.numel(), str(p.dtype)) for p in self._order)
        digest = zlib.crc32(repr(items).encode()) | (len(items) << 32)          # < 2^63
        t = torch.tensor([digest, -digest], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        hi, lo = int(t[0]), -int(t[1])
        if hi != lo:
            mine = len(items)
            self.reset()
            raise GradExchangeError(f"the ranks' backward passes produced gradients for different parameter sets (this rank: {mine} "
                                    f"live parameters, digest {digest:#x}; over the ranks {lo:#x} .. {hi:#x}): refusing to average "
                                    "unrelated tensors")
        order = torch.tensor([self._index[id(p)] for p in self._order], dtype=torch.int64, device=dev)
        dist.broadcast(order, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        self._order = [self.params[i] for i in order.tolist()]

    def _build_buckets(self):
        live = self._order
        self._buckets = []
        cur, cur_bytes = [], 0
        for p in live:
            nb = p.numel() * p.element_size()
            if cur and cur_bytes + nb > self.bucket_bytes:
                self._buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nb
        if cur:
            self._buckets.append(cur)
        out = []
        for bi, ps in enumerate(self._buckets):
            # every gradient starts on a 16-byte boundary of the bucket (the averaged gradient IS the bucket view, and the
            # multi-tensor Adam kernel takes its 16-byte path only when all four of a tensor's pointers allow it); the few
            # padding elements are zero and travel with the all-reduce
            al = max(1, 16 // ps[0].element_size())
            offs, off = [], 0
            for p in ps:
                offs.append(off)
                off += -(-p.numel() // al) * al
            flat = torch.zeros(off, dtype=ps[0].dtype, device=ps[0].device)
            for p, o in zip(ps, offs):
                self._where[id(p)] = (bi, o)
            views = [flat[o:o + p.numel()].view_as(p) for p, o in zip(ps, offs)]
            out.append(dict(params=ps, flat=flat, views=views, pending=len(ps), total=len(ps), launched=False))
        self._buckets = out
        self._next = 0

    @property
    def live_bytes(self):
        return sum(p.numel() * p.element_size() for b in (self._buckets or []) for p in b["params"])     # (without the alignment padding)

    @property
    def num_buckets(self):
        return len(self._buckets or [])

    # -- per-update ------------------------------------------------------------------
    def _note(self, msg):
        if self._local_error is None:
            self._local_error = msg

    def _exchange_stream(self):
        xs = self._xstream
        if isinstance(xs, str):
            from . import ops
            xs = self._xstream = ops.helper_stream(xs, device=self.params[0].device)
        return xs

    def _launch(self, bi):
        b = self._buckets[bi]
        xs = self._exchange_stream() if b["params"][0].is_cuda else None
        if b["params"][0].is_cuda:
            # parts of the backward graph run on side streams (instruction branch, decoder branch): the stream that packs the bucket
            # waits for every OTHER stream a gradient of the bucket was produced on.  One event per such stream, recorded now:
            # the hook of each of those gradients ran after its producer was queued, so "everything queued on that stream so
            # far" covers it (an event per gradient was 102 records per update: 1 ms of host time in the backward pass).
            cur = xs if xs is not None else torch.cuda.current_stream()
            for sid, st in b.pop("streams", {}).items():
                if sid != cur.cuda_stream:
                    ev = torch.cuda.Event()
                    ev.record(st)
                    cur.wait_event(ev)
        grads = []
        for q, v in zip(b["params"], b["views"]):
            if q.grad is None:        # only on the error path (a live parameter without a gradient): exchange zeros
                grads.append(torch.zeros_like(v))
            elif q.grad.data_ptr() == v.data_ptr():
                grads.append(None)    # the optimizer kept the bucket view as .grad and autograd accumulated in place
            else:
                grads.append(q.grad)
        # the whole bucket is ready: ONE multi-tensor copy packs it (instead of a copy kernel per parameter), then
        # the exchange starts while backward continues
        dst = [v for v, g in zip(b["views"], grads) if g is not None]
        b["launched"] = True
        if xs is not None:
            with torch.cuda.stream(xs):
                e0 = e1 = None
                if self._bucket_events is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(xs)
                if dst:
                    torch._foreach_copy_(dst, [g for g in grads if g is not None])
                dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group)     # synchronous form: runs on `xs` itself
                if e1 is not None:
                    e1.record(xs)
                    self._bucket_events.append((bi, e0, e1))
            self._works.append((bi, None))
            return
        if dst:
            torch._foreach_copy_(dst, [g for g in grads if g is not None])
        self._works.append((bi, dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)))

    def _launch_ready(self):
        """Issue every ready bucket whose predecessors have been issued: all ranks issue bucket 0, 1, 2, ... in that order."""
        if self._xstream is not None and self._buckets and self._buckets[0]["params"][0].is_cuda:
            # The exchange stream is the attention-stage stream of the chained recurrent core (recurrent.py: `sa`).  A blocking
            # cross-rank collective enqueued there BEFORE the core's backward pass would sit in front of the attention kernels the
            # resident, spinning recurrences wait for — a slow peer then runs their bounded spins out.  While a chained
            # forward pass is waiting for its backward, buckets are held; the core's backward releases them when it has queued its
            # kernels (its leaf pass calls these hooks), and finish() issues whatever is left.
            from .ops import core as _core
            if _core.chain_in_flight:
                return
        while self._next < len(self._buckets) and self._buckets[self._next]["pending"] == 0:
            self._launch(self._next)
            self._next += 1

    def _on_grad(self, p):
        if self._off:
            return
        # once per parameter and update: wsmgmap.recurrent assigns the gradients it produces on its leaf stream itself and calls this
        # hook from its backward; autograd's AccumulateGrad node may call it again for the same parameter afterwards
        if id(p) in self._seen:
            return
        self._seen.add(id(p))
        if self._buckets is None:
            self._order.append(p)
            return
        loc = self._where.get(id(p))
        if loc is None:
            self._note("a parameter that had no gradient in the discovery update received one later")
            return
        bi, off = loc
        b = self._buckets[bi]
        b["pending"] -= 1
        if p.is_cuda:
            st = torch.cuda.current_stream()
            b.setdefault("streams", {})[st.cuda_stream] = st     # see _launch
            leaf = getattr(p, "_wsmg_grad_stream", None)         # wsmgmap.recurrent computes parameter gradients on a leaf stream
            if leaf is not None:
                b["streams"][leaf.cuda_stream] = leaf
        if b["pending"] == 0:
            self._launch_ready()

    def reset(self):
        self._order, self._buckets, self._where, self._works = [], None, {}, []
        self._seen = set()
        self._next = 0
        self._local_error = None

    def _raise_if_flagged(self):
        """The ranks' summed error flag of the PREVIOUS update: every rank reads the same number at the same point."""
        if not self._flag_pending:
            return
        self._flag_pending = False
        if self._flag_event is not None:
            self._flag_event.synchronize()
        n = int(self._flag_host[0])
        if n:
            mine = self._prev_error
            self.reset()
            raise GradExchangeError(f"{n} of {self.world} ranks reported an error in the previous update"
                                    + (f" (this rank: {mine})" if mine else " (not this rank)")
                                    + "; that update's gradients were replaced by zeros on every rank (parameters stayed in step); "
                                      "the live-gradient set will be re-discovered on the next update")

    def _exchange_flag(self, dev):
        """-> device bool [1]: did any rank flag an error in THIS update (stream-ordered, no host synchronisation)."""
        if self._flag is None:
            self._flag = torch.zeros(1, dtype=torch.int32, device=dev)
            self._flag_host = torch.zeros(1, dtype=torch.int32, pin_memory=dev.type == "cuda")
        self._flag.fill_(1 if self._local_error else 0)
        if dev.type == "cuda" and self._exchange_stream() is not None:
            dist.all_reduce(self._flag, op=dist.ReduceOp.SUM, group=self.group)        # on the current stream (see exchange_stream)
        else:
            w = dist.all_reduce(self._flag, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            w.wait()    # NCCL/RCCL: the current stream waits, not the host
        self._flag_host.copy_(self._flag, non_blocking=True)
        if dev.type == "cuda":
            self._flag_event = torch.cuda.Event()
            self._flag_event.record(torch.cuda.current_stream())
        self._prev_error, self._local_error = self._local_error, None
        self._flag_pending = True
        return self._flag > 0

    def finish(self, check_now=False):
        """Call after backward() and before optimizer.step(): waits for the exchanges (the compute stream waits, not the host)
        and leaves the averaged gradients in `p.grad` — zeros on every rank if any rank flagged an error in this update.
        check_now=True: read the ranks' error flag of THIS update before returning (one host synchronisation per update) and
        raise GradExchangeError here; default: the flag is read at the start of the next finish()."""
        if self._off:
            return
        self._raise_if_flagged()
        t_host = time.perf_counter()
        dev = self.params[0].device
        ev0 = ev1 = None
        if dev.type == "cuda":
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record(torch.cuda.current_stream())      # the backward pass ends here on the compute stream
            from . import _abi
            try:
                _abi.check_rnn_status(force=True)
            except _abi.WsmgError as e:   # agreed on with the other ranks below: no rank may leave the collectives alone
                # in a data-parallel run the first suspect is co-residency: a persistent GRU / LSTM kernel needs all of its 32 / 16
                # workgroups running at once and spins (bounded) for its peers — beside the collective library's kernels and up to
                # three persistent kernels of the pipelined recurrent core, a workgroup that is not scheduled in time looks like this
                self._note(str(e) + " [data-parallel run: most likely the persistent kernels' workgroups were not co-resident beside "
                           "the collective's kernels — fall back to one persistent kernel at a time with policy.net.recurrent_chunks = 0 "
                           "(WSMG_RECURRENT_CHUNKS=0) and WSMG_DECODER_STREAMS=0, and check GPU_MAX_HW_QUEUES (8 for one process per GPU, "
                           "unset when ranks share a GPU)]")
        if dev.type == "cuda":
            from .ops import core as _core
            _core.chain_in_flight = False        # backward is over: nothing of the chained core is pending any more
            if self._buckets is not None:
                self._launch_ready()
        if self._buckets is None:  # first update: discovery pass, exchange synchronously
            self._agree_on_layout()
            self._build_buckets()
            for b in self._buckets:
                torch._foreach_copy_(b["views"], [p.grad for p in b["params"]])
                dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group)
            done = range(len(self._buckets))
        else:
            if self._next < len(self._buckets):
                # (only on an error path: a bucket never became ready.  The rest go out now, in index order — the order the
                # peers issue theirs in — with zeros for the missing gradients.)
                missing = [(self._index[id(q)], tuple(q.shape)) for b in self._buckets[self._next:] for q in b["params"] if q.grad is None]
                self._note("a live parameter received no gradient in this update"
                           + (f" (parameter index, shape: {missing[:4]}{' ...' if len(missing) > 4 else ''})" if missing else
                              f" (every gradient exists, but bucket {self._next} still waits for {self._buckets[self._next]['pending']} "
                              "gradient-ready hook call(s))"))
                while self._next < len(self._buckets):
                    self._launch(self._next)
                    self._next += 1
            for _, w in self._works:
                if w is not None:
                    w.wait()
            if dev.type == "cuda" and self._exchange_stream() is not None:
                torch.cuda.current_stream().wait_stream(self._exchange_stream())     # the one join of the exchange into the compute stream
            done = [bi for bi, _ in self._works]
        bad = self._exchange_flag(dev)
        inv = 1.0 / self.world
        for bi in done:
            b = self._buckets[bi]
            b["flat"].mul_(inv)
            b["flat"].masked_fill_(bad, 0.0)             # an error on any rank: the same zeros everywhere (NaN included)
            for p, v in zip(b["params"], b["views"]):
                p.grad = v                           # the averaged gradient lives in the bucket: no copy back
            b["pending"] = b["total"]
            b["launched"] = False
        self._works = []
        self._seen = set()
        self._next = 0
        if ev0 is not None:
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record(torch.cuda.current_stream())
        self._updates += 1
        if self._buffers_every > 0 and self._updates % self._buffers_every == 0:
            self.broadcast_buffers(self._module)
        self._timing.append((time.perf_counter() - t_host, ev0, ev1))
        if len(self._timing) > 64:
            self._fold_timing(keep=8)
        if check_now:
            self._raise_if_flagged()

    def _fold_timing(self, keep=0):
        """Fold completed updates' timings into the running statistics (events of the newest `keep` updates may be in flight)."""
        todo, self._timing = self._timing[:len(self._timing) - keep], self._timing[len(self._timing) - keep:]
        if self._bucket_events:
            # (the events of the newest `keep` updates may be in flight; older ones are folded here too — kept until stats() they were
            #  10 live events per update, 15 000 over a 1 500-update run, and the run's 50-update windows grew jittery: 10.3-12.4 ms)
            nkeep = keep * max(1, self.num_buckets)
            done = self._bucket_events[:len(self._bucket_events) - nkeep] if nkeep else self._bucket_events
            self._bucket_events = self._bucket_events[len(done):]
            for bi, e0, e1 in done:
                e1.synchronize()
                acc = self._bucket_ms.setdefault(bi, [0.0, 0])
                acc[0] += e0.elapsed_time(e1)
                acc[1] += 1
        for host_s, ev0, ev1 in todo:
            exposed = 0.0
            if ev0 is not None and ev1 is not None:
                ev1.synchronize()
                exposed = ev0.elapsed_time(ev1)
            s = self._stats
            s["updates"] += 1
            s["host_wait_ms"] += host_s * 1e3
            s["exposed_ms"] += exposed
            s["exposed_max_ms"] = max(s["exposed_max_ms"], exposed)

    def stats(self, reset=False):
        """Per-update means since construction (or the last reset): `host_ms_in_finish` = host time spent inside finish()
        (enqueueing the waits, the flag exchange and the scale kernels); `exposed_allreduce_ms` = time on the compute stream
        between the end of the backward pass and the last averaged bucket — the part of the gradient exchange that backward
        did NOT hide."""
        self._fold_timing()
        s = self._stats
        n = max(1, s["updates"])
        out = dict(updates=s["updates"], exposed_allreduce_ms=round(s["exposed_ms"] / n, 4), exposed_allreduce_max_ms=round(s["exposed_max_ms"], 4),
                   host_ms_in_finish=round(s["host_wait_ms"] / n, 4), buckets=self.num_buckets, live_gradient_bytes=self.live_bytes)
        if self._bucket_ms:
            out["per_bucket_allreduce_ms"] = [round(self._bucket_ms[bi][0] / max(1, self._bucket_ms[bi][1]), 4) for bi in sorted(self._bucket_ms)]
        if reset:
            self._stats = dict(updates=0, host_wait_ms=0.0, exposed_ms=0.0, exposed_max_ms=0.0)
            self._bucket_ms = {}
        return out

    def check(self):
        """Synchronous form of the deferred error check (end of training / tests)."""
        self._raise_if_flagged()
