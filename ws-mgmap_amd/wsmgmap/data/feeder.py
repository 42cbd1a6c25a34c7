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

TRACE = False     # tools set this to get the feeder's progress lines on stderr

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


def _ring_worker(dataset, batch_size, wid, nworkers, slots, free_q, ready_q, seed, persistent=False):
    """Decode worker: its shard of the dataset, whole batches, packed into the shared slots it is handed.  A batch larger than a
    slot (the slots are sized from a probe batch; episodes vary in length) does NOT fail the epoch: it travels once as its own
    shared-memory block ("__big__") and the worker waits for the consumer's acknowledgement before it goes on.
    persistent: after an epoch's _STOP the worker waits for ("epoch", seed) on its free queue and runs the next one (round 5: a ring
    that outlives the epoch — no process spawn, no re-pinning of the slots); None ends it."""
    import random
    import time

    import numpy as np
    try:
        torch.set_num_threads(1)
        dataset._worker_override = (nworkers, wid)
        spare = []                     # slot ids that came back while this worker was waiting for something else

        def take(want_ack):
            while True:
                if not want_ack and spare:
                    return spare.pop()
                tok = free_q.get()
                if tok is None or (tok == "ack") == want_ack:
                    return tok
                if tok != "ack":
                    spare.append(tok)

        def next_epoch():
            while True:
                tok = free_q.get()
                if tok is None:
                    return None
                if isinstance(tok, tuple) and tok[0] == "epoch":
                    return tok[1]
                if tok != "ack":
                    spare.append(tok)

        while True:
            random.seed(seed + wid)        # DataLoader seeds every worker with base_seed + worker_id
            np.random.seed((seed + wid) % (1 << 32))
            it = iter(dataset)
            while True:
                batch = []
                t0 = time.perf_counter()
                try:
                    while len(batch) < batch_size:
                        batch.append(next(it))
                except StopIteration:
                    pass
                if len(batch) < batch_size:          # drop_last=True, as the reference's loader (dagger_trainer.py:585-594)
                    break
                t1 = time.perf_counter()
                plan, meta = plan_batch(batch)
                t2 = time.perf_counter()
                if meta["total"] > slots[0].numel():
                    big = torch.empty(meta["total"], dtype=torch.uint8).share_memory_()
                    pack_batch(plan, meta, big.numpy())
                    ready_q.put(("__big__", meta, big))
                    if take(True) is None:
                        return
                    continue
                sid = take(False)
                if sid is None:
                    return
                t3 = time.perf_counter()
                pack_batch(plan, meta, slots[sid].numpy())
                # where this worker's time went, in seconds: reading / decoding the records, planning, waiting for a slot, packing
                meta["worker_times"] = (t1 - t0, t2 - t1, t3 - t2, time.perf_counter() - t3)
                ready_q.put((sid, meta))
            ready_q.put(_STOP)
            if not persistent:
                return
            seed = next_epoch()
            if seed is None:
                return
    except Exception as e:  # pragma: no cover - reported to the consumer
        import traceback
        ready_q.put(("__error__", traceback.format_exc() + repr(e)))


class DeviceFeeder:
    def __init__(self, dataset, batch_size, device="cuda", num_workers=0, prefetch=2, workers="ring", slot_bytes=None,
                 slots_per_worker=2, seed=None, ego_map_nhwc_bf16=False, persistent=False):
        """slot_bytes: capacity of one ring slot (default: sized from one batch this process plans itself — it decodes one batch
        of the first shard for that — with 50 % headroom, capped at the true upper bound batch_size x 200 steps; a larger batch
        travels outside the ring, counted in `oversize_batches`).
        seed: None (default) draws a fresh base seed from torch's generator for every epoch (= every `iter()`), which is what
        DataLoader does for its workers (dagger_trainer.py:116-119,585-594: the shuffle order changes each epoch); an integer is
        a reproducibility override: epoch e seeds worker w with seed + e * num_workers + w.
        persistent (ring transport): the ring — worker processes, shared slots, their page-locking — outlives the epoch: an epoch that
        ran to its end leaves it standing and the next `iter()` re-arms the workers with the new seed (the analogue of DataLoader's
        persistent_workers; without it every epoch pays 2-5 s of process start + pinning and 0.6-4.8 s of teardown).  `close()`
        (or garbage collection) takes it down; an epoch that is abandoned half way or fails takes it down too.  The workers keep the
        dataset as pickled when the ring was built: a different dataset object, length or shard at the next `iter()` rebuilds the ring;
        in-place changes of the same object that keep its length are NOT seen — use a fresh feeder (or `close()`) after those."""
        self.dataset, self.batch_size, self.device = dataset, batch_size, torch.device(device)
        self.num_workers, self.prefetch = int(num_workers), max(1, prefetch)
        self.workers = workers if self.num_workers > 0 else "none"
        if self.workers not in ("none", "ring", "dataloader"):
            raise ValueError("workers: 'ring' or 'dataloader'")
        self.slot_bytes, self.slots_per_worker = slot_bytes, max(2, int(slots_per_worker))
        self.seed = None if seed is None else int(seed)
        self.ego_map_nhwc_bf16 = bool(ego_map_nhwc_bf16)   # DeviceCollator's option: rgb_ego_map as channels-last bf16 (bf16 policies)
        self.epoch = 0                # iterators created so far (ring transport)
        self.oversize_batches = 0     # batches that did not fit a ring slot and travelled as their own shared-memory block
        self.pinned_ring = None       # True / False once a ring exists: could the shared slots be registered as pinned memory
        self.persistent = bool(persistent)
        self.ring_opens = 0           # rings built so far (persistent: 1 however many epochs)
        self._ring = None             # the standing ring of a persistent feeder between epochs

    def _dataset_fingerprint(self):
        """What a persistent ring's workers were built for: the dataset object, its length and its shard."""
        d = self.dataset
        try:
            n = len(d)
        except TypeError:
            n = getattr(d, "length", None)
        return (id(d), n, getattr(d, "rank", None), getattr(d, "world_size", None), self.batch_size)

    def _trace(self, msg):
        if TRACE:
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
        slots = [[DeviceCollator(self.device, self.ego_map_nhwc_bf16), None] for _ in range(self.prefetch + 1)]   # [collator, last event]
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
        """Slot capacity from one probe batch: 1.5 x its packed size, at most the true upper bound (every episode of a batch at
        the collate's 200-step cap: per-step bytes from the probe's sensor shapes and dtypes x batch_size x 200).  The probe decodes
        in this process; the global `random` state it advances (block shuffle, tie-breaks) is put back."""
        import copy
        import random

        import numpy as np
        state = random.getstate()
        try:
            ds = copy.copy(self.dataset)
            ds._worker_override = (self.num_workers, 0)
            it, batch = iter(ds), []
            try:
                while len(batch) < self.batch_size:
                    batch.append(next(it))
            except StopIteration:
                pass
        finally:
            random.setstate(state)
        if not batch:
            return 1 << 20
        plan, meta = plan_batch(batch)
        from .collate import LIMITED_LEN_BY_GPU
        per_step = sum(int(np.prod(a[0].shape[1:], dtype=np.int64)) * a[0].dtype.itemsize for _, a, _ in plan)
        bound = (per_step * LIMITED_LEN_BY_GPU + 16 * len(plan)) * self.batch_size + 4096
        return min(bound, int(meta["total"] * 1.5) + 4096)

    def _ring_open(self, base_seed):
        """Build the ring: shared slots (page-locked when the runtime allows), two queues and one decode process per worker."""
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
        from ..debug import sw
        self.pinned_ring = bool(sw.feeder_pin)
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
            p = ctx.Process(target=_ring_worker, args=(self.dataset, self.batch_size, w, W, slots, free_qs[w], ready_qs[w], base_seed,
                                                       self.persistent), daemon=True)
            p.start()
            procs.append(p)
        self._trace("%d workers started" % W)
        self.ring_opens += 1
        return dict(slots=slots, nbytes=nbytes, free_qs=free_qs, ready_qs=ready_qs, procs=procs, pinned=self.pinned_ring)

    def _ring_close(self, ring):
        if ring is None:
            return
        for q in ring["free_qs"]:
            q.put(None)
        for p in ring["procs"]:
            p.join(timeout=5)
            if p.is_alive():
                p.terminate()
        if ring["pinned"]:
            rt = torch.cuda.cudart()
            for t in ring["slots"]:
                rt.cudaHostUnregister(t.data_ptr())
        for q in ring["free_qs"] + ring["ready_qs"]:      # the queues' feeder threads and semaphores go with the ring, not with the interpreter
            q.close()
            q.cancel_join_thread()

    def close(self):
        """Take a persistent feeder's standing ring down (worker processes, page-locked slots).  Idempotent."""
        ring, self._ring = self._ring, None
        self._ring_close(ring)

    def __del__(self):  # pragma: no cover - best effort at interpreter exit
        try:
            self.close()
        except Exception:
            pass

    def _iter_ring(self):
        W = self.num_workers
        # DataLoader draws a fresh base seed per iterator from torch's default generator (worker w gets base_seed + w): the block
        # shuffle and the tie-breaks of equal-length episodes differ from epoch to epoch.  An explicit seed stays reproducible.
        if self.seed is None:
            base_seed = int(torch.empty((), dtype=torch.int64).random_().item())
        else:
            base_seed = self.seed + self.epoch * W
        self.epoch += 1
        ring, self._ring = self._ring, None
        # a standing ring's workers hold the dataset as it was PICKLED when the ring was built: a dataset that has changed since (a DAgger
        # cache that grew, another shard) gets a new ring instead of silently serving the old snapshot
        fp = self._dataset_fingerprint()
        if ring is not None and (not all(p.is_alive() for p in ring["procs"]) or ring.get("fingerprint") != fp):
            self._ring_close(ring)
            ring = None
        if ring is None:
            ring = self._ring_open(base_seed)
            ring["fingerprint"] = fp
        else:               # a standing ring: every slot is back with its worker; re-arm the workers
            self.pinned_ring = ring["pinned"]
            for q in ring["free_qs"]:
                q.put(("epoch", base_seed))
        slots, nbytes, free_qs, ready_qs, procs = ring["slots"], ring["nbytes"], ring["free_qs"], ring["ready_qs"], ring["procs"]
        side = torch.cuda.Stream(self.device)
        coll = DeviceCollator(self.device, self.ego_map_nhwc_bf16)
        pending = collections.deque()          # (out, event, worker, slot)
        live = [True] * W
        import time as _time
        # where the consumer's and the workers' time goes, per batch (reported under WSMG_FEEDER_TRACE; live: a caller may zero it
        # once the pipeline is full)
        tacc = self.consumer_times = dict(get=0.0, launch=0.0, sync=0.0, n=0, w_read=0.0, w_plan=0.0, w_slot=0.0, w_pack=0.0)
        done = False
        try:
            k = 0
            while any(live):
                w = k % W
                k += 1
                if not live[w]:
                    continue
                item, waited = None, 0.0
                _t0 = _time.perf_counter()
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
                tacc["get"] += _time.perf_counter() - _t0
                if item == _STOP:
                    live[w] = False
                    continue
                if item[0] == "__error__":
                    raise RuntimeError("feeder worker failed:\n" + item[1])
                if item[0] == "__big__":               # a batch larger than a slot: its own (unpinned) shared block, copied now
                    _, meta, big = item
                    self.oversize_batches += 1
                    self._trace("oversize batch from worker %d: %d MiB (slot %d MiB)" % (w, meta["total"] >> 20, nbytes >> 20))
                    out = coll.launch(meta, big, stream=side)
                    ev = torch.cuda.Event()
                    ev.record(side)
                    ev.synchronize()                   # the pageable copy has left `big`
                    del big
                    free_qs[w].put("ack")
                    pending.append((out, ev, w, None))
                else:
                    sid, meta = item
                    self._trace("batch from worker %d in slot %d" % (w, sid))
                    _t1 = _time.perf_counter()
                    out = coll.launch(meta, slots[sid], stream=side)
                    tacc["launch"] += _time.perf_counter() - _t1
                    tacc["n"] += 1
                    for key, v in zip(("w_read", "w_plan", "w_slot", "w_pack"), meta.get("worker_times", ())):
                        tacc[key] += v
                    ev = torch.cuda.Event()
                    ev.record(side)
                    pending.append((out, ev, w, sid))
                if len(pending) > self.prefetch:
                    out, ev, w0, s0 = pending.popleft()
                    _t2 = _time.perf_counter()
                    ev.synchronize()                  # the copy out of the slot has finished: the worker may refill it
                    tacc["sync"] += _time.perf_counter() - _t2
                    if s0 is not None:
                        free_qs[w0].put(s0)
                    yield self._hand_over(out, ev)
            while pending:
                out, ev, w0, s0 = pending.popleft()
                ev.synchronize()
                if s0 is not None:
                    free_qs[w0].put(s0)
                yield self._hand_over(out, ev)
            done = True
        finally:
            if tacc["n"]:
                self._trace("consumer per batch: waiting for the worker's batch %.1f ms, launch() %.1f ms, waiting for the device %.1f ms; "
                            "a worker per batch: records %.1f ms, plan %.1f ms, waiting for a slot %.1f ms, pack %.1f ms (%d batches)"
                            % tuple([tacc[k] / tacc["n"] * 1e3 for k in ("get", "launch", "sync", "w_read", "w_plan", "w_slot", "w_pack")]
                                    + [tacc["n"]]))
            if done and self.persistent:
                self._ring = ring        # every worker has sent its _STOP and has every slot back: the ring stands for the next epoch
            else:
                self._ring_close(ring)
