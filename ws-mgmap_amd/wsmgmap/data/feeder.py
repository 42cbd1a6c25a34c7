"""Prefetching feeder: record decode in worker PROCESSES (as the reference's `DataLoader(num_workers=...)` does,
dagger_trainer.py:560-575,585-594), batch assembly on the GPU (`DeviceCollator`) on a side stream, `prefetch` batches ahead of
the consumer.  What the consumer gets per iteration is exactly what the reference's training loop hands to `_update_agent`
(:606-625).

Two transports between the decode workers and this process:

* `workers="dataloader"`: torch's DataLoader — every decoded batch (735 MB of numpy arrays at BASELINE configs[1]) is pickled
  through a pipe by the worker and unpickled here, by ONE thread: ≈1 100 steps/s whatever the number of workers.
* `workers="ring"` (default with num_workers > 0): each worker packs its decoded batch straight into a slot of a shared-memory
  ring in the layout `DeviceCollator.launch` copies to the device (all episodes of all sensors back to back); only a few
  hundred bytes of metadata cross the queue.  This process registers the ring as pinned memory, so the host-to-device copy is
  asynchronous and reads the worker's bytes in place.  The decode itself (zlib inflate, ≈350 MB/s per core at 1.44 MB per
  policy step) is then the bound: ≈240 steps/s per worker.

Worker w of W reads the contiguous shard `shard_range(len, rank, world, W, w)` and its batches are taken in turn (w = 0, 1, ...,
W-1, 0, ...), which is the order DataLoader yields an IterableDataset's batches in."""
import collections
import multiprocessing as mp
import queue as _queue

import torch

from .collate import DeviceCollator, pack_batch, plan_batch

_STOP = "__stop__"


def _identity(batch):
    return batch


def _host_memory_available():
    """Bytes of host memory this process may still take: min(MemAvailable, cgroup limit - usage); None when unknown."""
    vals = []
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable:"):
                    vals.append(int(line.split()[1]) * 1024)
    except OSError:
        pass
    for lim, cur in (("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.current"),
                     ("/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes")):
        try:
            with open(lim) as f:
                v = f.read().strip()
            if v == "max":
                continue
            with open(cur) as f:
                used = int(f.read().strip())
            if int(v) < (1 << 60):
                vals.append(max(0, int(v) - used))
        except (OSError, ValueError):
            pass
    return min(vals) if vals else None


def _ring_worker(dataset, batch_size, wid, nworkers, slots, free_q, ready_q, seed):
    """Decode worker: its shard of the dataset, whole batches, packed into the shared slots it is handed."""
    import random

    import numpy as np
    try:
        torch.set_num_threads(1)
        random.seed(seed + wid)        # DataLoader seeds every worker with base_seed + worker_id
        np.random.seed((seed + wid) % (1 << 32))
        dataset._worker_override = (nworkers, wid)
        it = iter(dataset)
        while True:
            batch = []
            try:
                while len(batch) < batch_size:
                    batch.append(next(it))
            except StopIteration:
                pass
            if len(batch) < batch_size:          # drop_last=True, as the reference's loader (dagger_trainer.py:585-594)
                break
            plan, meta = plan_batch(batch)
            sid = free_q.get()
            if sid is None:
                return
            pack_batch(plan, meta, slots[sid].numpy())
            ready_q.put((sid, meta))
        ready_q.put(_STOP)
    except Exception as e:  # pragma: no cover - reported to the consumer
        import traceback
        ready_q.put(("__error__", traceback.format_exc() + repr(e)))


class DeviceFeeder:
    def __init__(self, dataset, batch_size, device="cuda", num_workers=0, prefetch=2, workers="ring", slot_bytes=None,
                 slots_per_worker=2, seed=0):
        """slot_bytes: capacity of one ring slot (default: sized from the first batch this process plans itself — it decodes
        one batch of the first shard for that — with 25 % headroom)."""
        self.dataset, self.batch_size, self.device = dataset, batch_size, torch.device(device)
        self.num_workers, self.prefetch = int(num_workers), max(1, prefetch)
        self.workers = workers if self.num_workers > 0 else "none"
        if self.workers not in ("none", "ring", "dataloader"):
            raise ValueError("workers: 'ring' or 'dataloader'")
        self.slot_bytes, self.slots_per_worker, self.seed = slot_bytes, max(2, int(slots_per_worker)), int(seed)
        self.pinned_ring = None       # True / False once a ring exists: could the shared slots be registered as pinned memory

    def _trace(self, msg):
        import os
        if os.environ.get("WSMG_FEEDER_TRACE") == "1":
            import sys
            import time
            print("[feeder %.1f] %s" % (time.time() % 10000, msg), file=sys.stderr, flush=True)

    # -- device side --------------------------------------------------------------------------------------------------
    def _hand_over(self, out, ev):
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        obs, prev, masks, corr, wts = out
        for t in list(obs.values()) + [prev, masks, corr, wts]:
            t.record_stream(cur)
        return out

    def __iter__(self):
        if self.workers == "ring":
            return self._iter_ring()
        return self._iter_loader()

    def _iter_loader(self):
        loader = torch.utils.data.DataLoader(self.dataset, batch_size=self.batch_size, collate_fn=_identity,
                                             num_workers=self.num_workers, drop_last=True)
        side = torch.cuda.Stream(self.device)
        slots = [[DeviceCollator(self.device), None] for _ in range(self.prefetch + 1)]   # [collator, last event]
        pending = collections.deque()
        for i, batch in enumerate(loader):
            slot = slots[i % len(slots)]
            if slot[1] is not None:
                slot[1].synchronize()            # the H2D copy out of this slot's pinned staging has finished
            out = slot[0](batch, stream=side)
            ev = torch.cuda.Event()
            ev.record(side)
            slot[1] = ev
            pending.append((out, ev))
            if len(pending) > self.prefetch:
                yield self._hand_over(*pending.popleft())
        while pending:
            yield self._hand_over(*pending.popleft())

    # -- shared-memory ring ---------------------------------------------------------------------------------------------
    def _probe_slot_bytes(self):
        import copy
        ds = copy.copy(self.dataset)
        ds._worker_override = (self.num_workers, 0)
        it, batch = iter(ds), []
        try:
            while len(batch) < self.batch_size:
                batch.append(next(it))
        except StopIteration:
            pass
        if not batch:
            return 1 << 20
        _, meta = plan_batch(batch)
        # episodes of other batches may be longer (up to the collate's 200-step cap): scale to the cap
        from .collate import LIMITED_LEN_BY_GPU
        scale = LIMITED_LEN_BY_GPU / max(1, meta["T"])
        return int(meta["total"] * min(scale, 4.0) * 1.25) + 4096

    def _iter_ring(self):
        W = self.num_workers
        # a worker's slot comes back when its batch leaves the prefetch queue: with `prefetch` batches queued (taken from the workers
        # in turn) a worker needs ceil(prefetch / W) slots there, one being handed over and one to fill meanwhile — fewer deadlocks
        # the ring (the worker waits for a slot, the consumer for the worker's next batch)
        self.slots_per_worker = max(self.slots_per_worker, -(-self.prefetch // W) + 2)
        nbytes = self.slot_bytes or self._probe_slot_bytes()
        need = nbytes * W * self.slots_per_worker
        try:     # the ring lives in /dev/shm (torch's shared-memory tensors): refuse up front what would die with a bus error later
            import os
            st = os.statvfs("/dev/shm")
            if need > st.f_bavail * st.f_frsize:
                raise RuntimeError(f"the feeder's shared-memory ring needs {need >> 20} MiB ({W} workers x {self.slots_per_worker} slots x "
                                   f"{nbytes >> 20} MiB) but /dev/shm has {(st.f_bavail * st.f_frsize) >> 20} MiB free: fewer workers or "
                                   "slots_per_worker, or a larger /dev/shm")
        except OSError:
            pass
        # ... and what would take the machine with it: the ring is page-locked host memory, and every worker is a process with its
        # own decode buffers on top — refuse a ring above 40 % of what this process may still use (the smaller of MemAvailable and
        # the control group's limit; a 96-worker ring of 700 MiB batches is 212 GB)
        avail = _host_memory_available()
        if avail is not None and need > 0.4 * avail:
            raise RuntimeError(f"the feeder's shared-memory ring needs {need >> 20} MiB ({W} workers x {self.slots_per_worker} slots x "
                               f"{nbytes >> 20} MiB), more than 40 % of the {avail >> 20} MiB of host memory this process may use: "
                               "fewer workers or slots_per_worker")
        ctx = mp.get_context("spawn")
        slots = [torch.empty(nbytes, dtype=torch.uint8).share_memory_() for _ in range(W * self.slots_per_worker)]
        rt = torch.cuda.cudart()
        import os
        self.pinned_ring = os.environ.get("WSMG_FEEDER_PIN", "1") != "0"
        for t in slots:       # pinned: the H2D copy is asynchronous and reads the worker's bytes in place
            if not self.pinned_ring:
                break
            if int(rt.cudaHostRegister(t.data_ptr(), t.numel(), 0)) != 0:
                self.pinned_ring = False
        self._trace("ring ready: %d slots of %d MiB, pinned=%s" % (len(slots), nbytes >> 20, self.pinned_ring))
        free_qs = [ctx.Queue() for _ in range(W)]
        ready_qs = [ctx.Queue() for _ in range(W)]
        procs = []
        for w in range(W):
            for k in range(self.slots_per_worker):
                free_qs[w].put(w * self.slots_per_worker + k)
            p = ctx.Process(target=_ring_worker, args=(self.dataset, self.batch_size, w, W, slots, free_qs[w], ready_qs[w], self.seed),
                            daemon=True)
            p.start()
            procs.append(p)
        self._trace("%d workers started" % W)
        side = torch.cuda.Stream(self.device)
        coll = DeviceCollator(self.device)
        pending = collections.deque()          # (out, event, worker, slot)
        live = [True] * W
        try:
            k = 0
            while any(live):
                w = k % W
                k += 1
                if not live[w]:
                    continue
                item, waited = None, 0.0
                while item is None:
                    try:
                        item = ready_qs[w].get(timeout=2.0)
                    except _queue.Empty:
                        waited += 2.0
                        if not procs[w].is_alive() and ready_qs[w].empty():
                            raise RuntimeError(f"feeder worker {w} died (exit code {procs[w].exitcode}) — a bus error (-7) means the "
                                               f"shared-memory ring ({len(slots)} slots of {nbytes >> 20} MiB) does not fit in /dev/shm")
                        if waited >= 900.0:
                            raise RuntimeError(f"feeder worker {w} produced nothing for 15 minutes")
                if item == _STOP:
                    live[w] = False
                    continue
                if item[0] == "__error__":
                    raise RuntimeError("feeder worker failed:\n" + item[1])
                sid, meta = item
                self._trace("batch from worker %d in slot %d" % (w, sid))
                out = coll.launch(meta, slots[sid], stream=side)
                ev = torch.cuda.Event()
                ev.record(side)
                pending.append((out, ev, w, sid))
                if len(pending) > self.prefetch:
                    out, ev, w0, s0 = pending.popleft()
                    ev.synchronize()                  # the copy out of the slot has finished: the worker may refill it
                    free_qs[w0].put(s0)
                    yield self._hand_over(out, ev)
            while pending:
                out, ev, w0, s0 = pending.popleft()
                ev.synchronize()
                free_qs[w0].put(s0)
                yield self._hand_over(out, ev)
        finally:
            for q in free_qs:
                q.put(None)
            for p in procs:
                p.join(timeout=5)
                if p.is_alive():
                    p.terminate()
            if self.pinned_ring:
                for t in slots:
                    rt.cudaHostUnregister(t.data_ptr())
            for q in free_qs + ready_qs:      # the queues' feeder threads and semaphores go with the ring, not with the interpreter
                q.close()
                q.cancel_join_thread()
