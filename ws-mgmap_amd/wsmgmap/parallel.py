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

BatchNorm statistics stay per rank (no SyncBN), like the reference; `broadcast_buffers()` gives every rank
rank 0's buffers (what DDP's per-forward buffer broadcast, C4 in SURVEY.md, amounts to at checkpoint time).

Errors are agreed on across ranks: a rank that sees a local problem (a gradient outside the discovered live
set, a live parameter without a gradient, a persistent-RNN timeout) still takes part in every collective of the
update — so no peer blocks in an all-reduce — and raises a flag that is summed over the ranks by one
extra 4-byte all-reduce per update; every rank reads the sum at the start of the NEXT `finish()` (by then the
copy to pinned memory is long complete: no stall) and all of them raise there together, after resetting to a
fresh discovery pass.

Works with any torch.distributed backend (gloo on CPU for tests).
"""
import torch
import torch.distributed as dist


class GradExchangeError(RuntimeError):
    pass


class GradAllReducer:
    def __init__(self, params, bucket_bytes=8 << 20, group=None):
        self.params = [p for p in params if p.requires_grad]
        seen, uniq = set(), []
        for p in self.params:
            if id(p) not in seen:
                seen.add(id(p))
                uniq.append(p)
        self.params = uniq
        self.bucket_bytes = bucket_bytes
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._order = []          # discovery pass: params in gradient-ready order
        self._buckets = None      # list of dicts: params, flat buffer, pending count
        self._where = {}          # id(param) -> (bucket index, offset)
        self._works = []
        self._local_error = None  # first local problem of the running update
        self._flag = None         # device int32 [1]: this rank's error flag, summed over the ranks
        self._flag_host = None    # pinned copy of the summed flag of the previous update (+ event)
        self._flag_event = None
        self._flag_pending = False
        self._prev_error = None
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    # -- setup ---------------------------------------------------------------------
    def broadcast_parameters(self, module, src=0):
        """rank-`src` parameters and buffers to every rank (DDP constructor behaviour)."""
        if self.world == 1:
            return
        for t in list(module.parameters()) + [b for b in module.buffers() if b.is_floating_point() or b.dtype == torch.int64]:
            dist.broadcast(t.data, src=src, group=self.group)

    def broadcast_buffers(self, module, src=0):
        """rank-`src` buffers (BatchNorm running statistics, counters) to every rank: the state DDP's
        `broadcast_buffers=True` keeps identical by re-broadcasting before every forward (common_trainer.py:61-66
        default).  Statistics here stay per rank during training; call this before evaluating or checkpointing from a
        rank other than `src` (the reference checkpoints rank 0's, common_trainer.py:99)."""
        if self.world == 1:
            return
        for b in module.buffers():
            if b.is_floating_point() or b.dtype == torch.int64:
                dist.broadcast(b.data, src=src, group=self.group)

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
            n = sum(p.numel() for p in ps)
            flat = torch.empty(n, dtype=ps[0].dtype, device=ps[0].device)
            off = 0
            for p in ps:
                self._where[id(p)] = (bi, off)
                off += p.numel()
            views, o2 = [], 0
            for p in ps:
                views.append(flat[o2:o2 + p.numel()].view_as(p))
                o2 += p.numel()
            out.append(dict(params=ps, flat=flat, views=views, pending=len(ps), total=len(ps), launched=False))
        self._buckets = out

    @property
    def live_bytes(self):
        return sum(b["flat"].numel() * b["flat"].element_size() for b in (self._buckets or []))

    @property
    def num_buckets(self):
        return len(self._buckets or [])

    # -- per-update ------------------------------------------------------------------
    def _note(self, msg):
        if self._local_error is None:
            self._local_error = msg

    def _launch(self, bi):
        b = self._buckets[bi]
        if b["params"][0].is_cuda:
            cur = torch.cuda.current_stream()
            for ev in b.pop("events", []):
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
        if dst:
            torch._foreach_copy_(dst, [g for g in grads if g is not None])
        b["launched"] = True
        self._works.append((bi, dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)))

    def _on_grad(self, p):
        if self.world == 1:
            return
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
            # parts of the backward graph run on side streams (instruction branch, decoder branch): the stream that packs
            # the bucket must wait for the streams that produced the other gradients in it
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            b.setdefault("events", []).append(ev)
        if b["pending"] == 0:
            self._launch(bi)

    def reset(self):
        self._order, self._buckets, self._where, self._works = [], None, {}, []
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
                                    + "; the live-gradient set will be re-discovered on the next update")

    def _exchange_flag(self, dev):
        if self._flag is None:
            self._flag = torch.zeros(1, dtype=torch.int32, device=dev)
            self._flag_host = torch.zeros(1, dtype=torch.int32, pin_memory=dev.type == "cuda")
        self._flag.fill_(1 if self._local_error else 0)
        w = dist.all_reduce(self._flag, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        w.wait()    # NCCL/RCCL: the current stream waits, not the host
        self._flag_host.copy_(self._flag, non_blocking=True)
        if dev.type == "cuda":
            self._flag_event = torch.cuda.Event()
            self._flag_event.record(torch.cuda.current_stream())
        self._prev_error, self._local_error = self._local_error, None
        self._flag_pending = True

    def finish(self):
        """Call after backward(): waits for the exchanges and writes averaged gradients back."""
        if self.world == 1:
            return
        self._raise_if_flagged()
        dev = self.params[0].device
        if dev.type == "cuda":
            from . import _abi
            try:
                _abi.check_rnn_status()
            except _abi.WsmgError as e:   # agreed on with the other ranks below: no rank may leave the collectives alone
                self._note(str(e))
        if self._buckets is None:  # first update: discovery pass, exchange synchronously
            self._build_buckets()
            for b in self._buckets:
                torch._foreach_copy_(b["views"], [p.grad for p in b["params"]])
                dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group)
            done = range(len(self._buckets))
        else:
            for bi, b in enumerate(self._buckets):
                if not b["launched"]:
                    self._note("a live parameter received no gradient in this update")
                    self._launch(bi)          # with zeros for the missing ones: the peers are waiting in this all-reduce
            for _, w in self._works:
                w.wait()
            done = [bi for bi, _ in self._works]
        inv = 1.0 / self.world
        for bi in done:
            b = self._buckets[bi]
            b["flat"].mul_(inv)                      # one kernel per bucket
            for p, v in zip(b["params"], b["views"]):
                p.grad = v                           # the averaged gradient lives in the bucket: no copy back
            b["pending"] = b["total"]
            b["launched"] = False
        self._works = []
        self._exchange_flag(dev)

    def check(self):
        """Synchronous form of the deferred error check (end of training / tests)."""
        self._raise_if_flagged()
