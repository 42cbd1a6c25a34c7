"""Data-parallel gradient exchange for the teacher-forcing / DAgger update: one process per
GPU, one averaged all-reduce of the LIVE gradients per update over RCCL/xGMI (backend "nccl"),
bucketed and launched from autograd hooks so it overlaps the rest of backward.  Per update and bucket: one
multi-tensor pack, one all-reduce, one scale; the averaged gradients stay in the bucket (`p.grad` becomes a view of it).

Replaces `DistributedDataParallel(find_unused_parameters=True)` at the reference's
common_trainer.py:61-66.  The reference all-reduces all 19.76 M trainable floats (79 MB) although
11.5 M of them never receive a gradient (unused resnet18 layers 2-4 / fc, critic, logstd);
here the first backward discovers which parameters are live (8.23 M, 32.9 MB) and only those
are exchanged, in buckets filled in gradient-ready order (xGMI ring all-reduce is per-link
bound, so a few ~8 MB buckets keep every link busy while backward continues).

BatchNorm statistics stay per rank (no SyncBN), like the reference.
Works with any torch.distributed backend (gloo on CPU for tests).
"""
import torch
import torch.distributed as dist


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
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    # -- setup ---------------------------------------------------------------------
    def broadcast_parameters(self, module, src=0):
        """rank-`src` parameters and buffers to every rank (DDP constructor behaviour)."""
        if self.world == 1:
            return
        for t in list(module.parameters()) + [b for b in module.buffers() if b.is_floating_point() or b.dtype == torch.int64]:
            dist.broadcast(t.data, src=src, group=self.group)

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
            out.append(dict(params=ps, flat=flat, views=views, pending=len(ps), total=len(ps)))
        self._buckets = out

    @property
    def live_bytes(self):
        return sum(b["flat"].numel() * b["flat"].element_size() for b in (self._buckets or []))

    # -- per-update ------------------------------------------------------------------
    def _on_grad(self, p):
        if self.world == 1:
            return
        if self._buckets is None:
            self._order.append(p)
            return
        loc = self._where.get(id(p))
        if loc is None:
            raise RuntimeError("a parameter that had no gradient in the first update received one later; "
                               "call reset() to re-discover the live set")
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
            if p.is_cuda:
                cur = torch.cuda.current_stream()
                for ev in b.pop("events", []):
                    cur.wait_event(ev)
            # the whole bucket is ready: ONE multi-tensor copy packs it (instead of a copy kernel per parameter), then
            # the exchange starts while backward continues
            torch._foreach_copy_(b["views"], [q.grad for q in b["params"]])
            self._works.append((bi, dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)))

    def reset(self):
        self._order, self._buckets, self._where, self._works = [], None, {}, []

    def finish(self):
        """Call after backward(): waits for the exchanges and writes averaged gradients back."""
        if self.world == 1:
            return
        if self._buckets is None:  # first update: discovery pass, exchange synchronously
            self._build_buckets()
            for b in self._buckets:
                torch._foreach_copy_(b["views"], [p.grad for p in b["params"]])
                dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group)
            done = range(len(self._buckets))
        else:
            for b in self._buckets:
                if b["pending"] != 0:
                    raise RuntimeError("a live parameter received no gradient in this update")
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
        self._works = []
