#!/usr/bin/env python3
"""GPU-box tool: throughput of the trajectory-cache path for one BASELINE configs[1] batch (N = 8 episodes x T = 64
steps, on-disk dtypes of common_trainer.py:514-532): record decode (zlib + msgpack), and batch assembly — the
reference's way (host collate_fn -> .float() -> .to(device), dagger_trainer.py:40-113,614-625) against
DeviceCollator (compact dtypes over PCIe, pad/interleave/convert on the GPU)."""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ws-mgmap_amd"))
import numpy as np
import torch
from wsmgmap.data import pack_record, pack_record_raw, unpack_record, collate_fn, DeviceCollator

N, T = 8, 64
rng = np.random.RandomState(0)
def episode():
    ego = np.maximum(rng.randn(T, 64, 100, 100), 0).astype(np.float16)
    ego[rng.rand(*ego.shape) < 0.6] = 0       # real maps are 20-45 % non-zero
    obs = {
        "rgb_ego_map": ego,
        "rgb_features": rng.randn(T, 512, 7, 7).astype(np.float16),
        "depth_features": rng.randn(T, 128, 4, 4).astype(np.float16),
        "instruction": np.tile(np.pad(rng.randint(1, 2504, size=80), (0, 120)), (T, 1)).astype(np.int64),
        "gt_semantic_map": rng.randint(0, 27, size=(T, 100, 100)).astype(np.int64),
        "gt_path": (rng.rand(T, 100, 100) * 50).astype(np.float16),
        "progress": rng.rand(T, 1).astype(np.float32),
        "waypoint": rng.randn(T, 2).astype(np.float32),
    }
    return obs, rng.randn(T, 2).astype(np.float32), rng.randn(T, 2).astype(np.float32)
raw = 0


def host_side_and_collate():
    global raw
    eps = [episode() for _ in range(N)]
    raw = sum(sum(v.nbytes for v in e[0].values()) + e[1].nbytes + e[2].nbytes for e in eps)
    t0 = time.time(); blobs = [pack_record(*e, level=1) for e in eps]; t_pack = time.time() - t0
    print(f"batch: {N} episodes x {T} steps = {raw / 1e6:.0f} MB on-disk dtypes ({raw / (N * T) / 1e6:.2f} MB/step), {sum(map(len, blobs)) / 1e6:.0f} MB compressed; pack {t_pack:.2f} s")
    t0 = time.time(); recs = [unpack_record(b) for b in blobs]; t1 = time.time() - t0
    print(f"decode (zlib + msgpack): {t1 * 1e3:.0f} ms on 1 thread = {N * T / t1:.0f} steps/s")
    for nt in (8, 32):   # zlib releases the GIL; one record per thread, so more records than a batch are needed to use more threads
        many = blobs * (nt // 8)
        with ThreadPoolExecutor(nt) as ex:
            t0 = time.time(); out = list(ex.map(unpack_record, many)); dt = time.time() - t0
        print(f"decode on {nt} threads: {len(many)} records in {dt * 1e3:.0f} ms = {len(many) * T / dt:.0f} steps/s ({os.cpu_count()} host CPUs)")
    recs = out[:N]
    batch = [(r[0], r[1], r[2], torch.ones(T)) for r in recs]
    def ref_path():
        ob, prev, masks, corr, wts = collate_fn(batch)
        ob = {k: v.float().to("cuda", non_blocking=True) for k, v in ob.items()}
        out = (ob, prev.to("cuda"), masks.to("cuda"), corr.to("cuda"), wts.to("cuda"))
        torch.cuda.synchronize()
        return out
    coll = DeviceCollator("cuda")
    def dev_path():
        out = coll(batch)
        torch.cuda.synchronize()
        return out
    for name, f in (("reference-style host collate + float() + H2D", ref_path), ("DeviceCollator", dev_path)):
        f()
        t0 = time.time()
        for _ in range(3): out = f()
        dt = (time.time() - t0) / 3
        print(f"{name:46s} {dt * 1e3:8.1f} ms per batch = {N * T / dt:8.0f} steps/s")
    a, b = ref_path(), dev_path()
    assert all(torch.equal(a[0][k], b[0][k]) for k in a[0]) and all(torch.equal(x, y) for x, y in zip(a[1:], b[1:]))
    print("both paths produce identical tensors")
    # one batch through the bf16 / channels-last collate, dense and sparse ego map (round 5), synchronously: staging copy, H2D, kernels
    from wsmgmap.data import pack_record_raw as _pr, unpack_record as _ur
    from wsmgmap.data.collate import plan_batch, pack_batch
    for sp in (False, True):
        recs2 = [_ur(_pr(r[0], r[1], r[2], sparse_ego=sp)) for r in recs]
        batch2 = [({k: np.asarray(v) for k, v in r[0].items()}, np.asarray(r[1]), np.asarray(r[2]), torch.ones(T)) for r in recs2]
        c2 = DeviceCollator("cuda", ego_map_nhwc_bf16=True)
        plan, meta = plan_batch(batch2)
        host = c2._staging(meta["total"])
        t0 = time.time(); pack_batch(plan, meta, host.numpy()); t_pack = time.time() - t0
        c2.launch(meta, host); torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(5):
            th = time.time(); c2.launch(meta, host); t_host = time.time() - th
            torch.cuda.synchronize()
        dt = (time.time() - t0) / 5
        print(f"one batch, ego map {'sparse' if sp else 'dense '} -> NHWC bf16: {meta['total'] / 1e6:6.0f} MB staged; worker-side pack (memcpy into the slot) "
              f"{t_pack * 1e3:6.1f} ms; consumer-side launch() host {t_host * 1e3:5.1f} ms; H2D + kernels to completion {dt * 1e3:6.1f} ms "
              f"= {N * T / dt:7.0f} steps/s if nothing overlapped")




# ---- end to end: record store -> decode workers (processes) -> pinned shared-memory ring -> H2D + device collate --------------
# (VERDICT r02 #8.)  The store is in memory (8 distinct records, built once per process from a seed: nothing but the seed is pickled
# to the workers), every worker decodes whole batches of its shard, the consumer does what a training loop does with a batch:
# waits for it on its stream.  Reported next to the compute rate of the update (bench.py: ~42 k steps/s).
class SynthStore:
    def __init__(self, n_records, seed=0, raw_records=False, sparse_ego=False):
        self.n, self.seed, self._blobs, self.raw_records, self.sparse_ego = n_records, seed, None, raw_records, sparse_ego

    def __getstate__(self):
        return dict(n=self.n, seed=self.seed, _blobs=None, raw_records=self.raw_records, sparse_ego=self.sparse_ego)

    def __call__(self, i):
        if self._blobs is None:
            global rng
            rng = np.random.RandomState(self.seed)
            # raw_records: the recoded cache of tools/recode_cache.py (uncompressed, zero-copy decode)
            self._blobs = [(pack_record_raw(*episode(), sparse_ego=self.sparse_ego) if self.raw_records else pack_record(*episode(), level=1))
                           for _ in range(4)]
        return self._blobs[i % len(self._blobs)]


def feeder_rate(workers, batches_per_worker=6, transport="ring", raw_records=False, sparse_ego=False, nhwc_bf16=False):
    from wsmgmap.data import TrajectoryDataset, DeviceFeeder
    nw = max(workers, 1)
    ds = TrajectoryDataset(SynthStore(N * batches_per_worker * nw, raw_records=raw_records, sparse_ego=sparse_ego), N * batches_per_worker * nw,
                           batch_size=N)
    fd = DeviceFeeder(ds, N, "cuda", num_workers=workers, prefetch=2, workers=transport, slot_bytes=int(raw * 1.05) + (1 << 20),
                      ego_map_nhwc_bf16=nhwc_bf16 or sparse_ego)
    t_first, n, steps = None, 0, 0
    for ob, prev, masks, corr, wts in fd:
        torch.cuda.current_stream().synchronize()
        if n == nw - 1 or (workers == 0 and n == 0):     # every worker has delivered once: its store is built, the pipeline is full
            t_first, steps = time.time(), 0
            for key in getattr(fd, "consumer_times", None) or {}:
                fd.consumer_times[key] = 0
        elif t_first is not None:
            steps += prev.shape[0]
        t_last = time.time()
        n += 1
    # (up to the last batch: leaving the loop also tears the ring down — joins the workers, unpins 2 slots per worker of one batch
    # each, seconds for a 16-worker ring — which is per epoch, not per batch; reported separately)
    dt = t_last - t_first
    feeder_rate.last_teardown = time.time() - t_last
    ct = getattr(fd, "consumer_times", None)
    if ct and ct.get("n"):
        feeder_rate.last_consumer = ("consumer per batch: worker wait %.1f ms, launch %.1f ms, device wait %.1f ms; a worker per batch: records "
                                     "%.1f ms, plan %.1f ms, slot wait %.1f ms, pack %.1f ms; ring teardown after the epoch %.1f s") % (tuple(
            ct[k] / ct["n"] * 1e3 for k in ("get", "launch", "sync", "w_read", "w_plan", "w_slot", "w_pack")) + (feeder_rate.last_teardown,))
    return steps / dt, n, fd.pinned_ring


if __name__ == "__main__":      # (the decode workers are spawned: they import this file, and must not run the measurements)
    host_side_and_collate()
if __name__ == "__main__" and os.environ.get("WSMG_FEEDER_E2E", "1") != "0":
    print("end to end (in-memory record store -> decode processes -> pinned shared-memory ring -> H2D + device collate):")
    st_ = os.statvfs("/dev/shm")
    print(f"  /dev/shm free: {st_.f_bavail * st_.f_frsize / 2**30:.1f} GiB; one slot = one batch = {raw / 2**20:.0f} MiB, 2 slots per worker")
    from wsmgmap.data.feeder import _host_memory_available
    print(f"  host memory this process may use: {(_host_memory_available() or 0) / 2**30:.0f} GiB; CPUs: {len(os.sched_getaffinity(0))}")
    for w in [min(int(x), 32) for x in os.environ.get("WSMG_FEEDER_WORKERS", "1,8,32").split(",")]:   # (a 96-worker ring is 212 GB pinned)
        try:
            r, n, pinned = feeder_rate(w)
        except RuntimeError as e:
            print(f"  ring, {w:2d} decode processes: not run — {str(e).splitlines()[-1][:300]}")
            continue
        print(f"  ring, {w:2d} decode processes: {r:8.0f} steps/s  ({n} batches; ring pinned: {pinned}; {r / max(w, 1):.0f} steps/s per worker)")
    print("the same from a RECODED cache (tools/recode_cache.py: uncompressed raw records, no inflate / msgpack parse):")
    for w in [min(int(x), 32) for x in os.environ.get("WSMG_FEEDER_WORKERS_RAW", "1,4,8,16").split(",")]:
        try:
            r, n, pinned = feeder_rate(w, batches_per_worker=12, raw_records=True)
        except RuntimeError as e:
            print(f"  raw ring, {w:2d} worker processes: not run — {str(e).splitlines()[-1][:300]}")
            continue
        print(f"  raw ring, {w:2d} worker processes: {r:8.0f} steps/s  ({n} batches; ring pinned: {pinned}; {r / max(w, 1):.0f} steps/s per worker)")
    print("the same with the ego map SPARSE in the recoded cache (recode_cache.py --sparse-ego: presence bits + packed non-zeros; this synthetic map "
          "is 20 % non-zero, real ones 20-45 %) and channels-last bf16 out of the collate (what the bf16 policy reads):")
    e0 = episode()
    b_dense, b_sparse = len(pack_record_raw(*e0)), len(pack_record_raw(*e0, sparse_ego=True))
    print(f"  bytes per step over PCIe: dense {b_dense / T / 1e6:.3f} MB, sparse {b_sparse / T / 1e6:.3f} MB")
    for w in [min(int(x), 32) for x in os.environ.get("WSMG_FEEDER_WORKERS_RAW", "1,4,8,16").split(",")]:
        for sp in (False, True):
            try:
                r, n, pinned = feeder_rate(w, batches_per_worker=12, raw_records=True, sparse_ego=sp, nhwc_bf16=True)
            except RuntimeError as e:
                print(f"  raw ring, {w:2d} workers, sparse={sp}: not run — {str(e).splitlines()[-1][:300]}")
                continue
            print(f"  raw ring, {w:2d} worker processes, ego map {'sparse' if sp else 'dense '} -> NHWC bf16: {r:8.0f} steps/s  ({n} batches; "
                  f"{getattr(feeder_rate, 'last_consumer', '')})")
    # epoch to epoch: from the last batch of one epoch to the first batch of the next, with the ring rebuilt (the default: what the
    # reference's DataLoader does with its workers too) and with a ring that outlives the epoch (DeviceFeeder(persistent=True))
    from wsmgmap.data import TrajectoryDataset, DeviceFeeder
    for persistent in (False, True):
        w = 8
        ds = TrajectoryDataset(SynthStore(N * 4 * w, raw_records=True, sparse_ego=True), N * 4 * w, batch_size=N)
        fd = DeviceFeeder(ds, N, "cuda", num_workers=w, prefetch=2, slot_bytes=int(raw * 1.05) + (1 << 20), ego_map_nhwc_bf16=True,
                          persistent=persistent)
        gaps, t_end = [], None
        for ep in range(3):
            for k, batch in enumerate(fd):
                torch.cuda.current_stream().synchronize()
                if k == 0 and t_end is not None:
                    gaps.append(time.time() - t_end)
                t_last = time.time()
            t_end = t_last
        t0 = time.time()
        fd.close()
        print(f"  epoch -> epoch, {w} workers, sparse ego map, persistent={persistent}: last batch of an epoch to the first of the next "
              f"{' / '.join('%.2f' % g for g in gaps)} s (rings built: {fd.ring_opens}; close() {time.time() - t0:.2f} s)")
    r, n, _ = feeder_rate(8, transport="dataloader")
    print(f"  torch DataLoader transport, 8 workers (batches pickled through a pipe): {r:8.0f} steps/s")
