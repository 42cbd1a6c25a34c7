#!/usr/bin/env python3
"""One-off: rewrite a DAgger trajectory cache (the reference's LMDB: key str(i).encode(), value zlib(msgpack_numpy(record)),
dagger_trainer.py:336-343) into the uncompressed raw record layout of wsmgmap.data.codec, which `TrajectoryDataset` /
`DeviceFeeder` read with no decode at all (zero-copy views; the arrays keep their on-disk dtypes, so batches are bit-identical).

    python tools/recode_cache.py <src.lmdb> <dst.lmdb> [--map-size-gb 2000] [--workers 16]

Needs the `lmdb` package (as the reference does).  The raw cache is ~2.8 x the size of the compressed one (1.44 MB per step).
`recode_store(get, n, put)` is the store-agnostic form (any `get(i) -> bytes`, `put(i, bytes)`)."""
import argparse
import os
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ws-mgmap_amd"))
from wsmgmap.data.codec import recode_record  # noqa: E402


def recode_store(get, n, put, workers=8, chunk=64, sparse_ego=False):
    """Recode records 0 .. n-1 of `get` into `put`; zlib releases the GIL, so threads scale the inflate.  -> (bytes in, bytes out)."""
    nin = nout = 0
    with ThreadPoolExecutor(max(1, workers)) as ex:
        for base in range(0, n, chunk):
            ids = list(range(base, min(n, base + chunk)))
            blobs = [bytes(get(i)) for i in ids]
            for i, b, r in zip(ids, blobs, ex.map(lambda x: recode_record(x, sparse_ego=sparse_ego), blobs)):
                put(i, r)
                nin += len(b)
                nout += len(r)
    return nin, nout


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--map-size-gb", type=float, default=2000.0)
    ap.add_argument("--workers", type=int, default=16)
    ap.add_argument("--sparse-ego", action="store_true",
                    help="store rgb_ego_map as presence bits + packed non-zeros (codec.sparse_pack_ego): 0.4-0.6 x the bytes per step; "
                         "needs DeviceCollator / DeviceFeeder(ego_map_nhwc_bf16=True), i.e. a bf16 policy")
    a = ap.parse_args()
    import lmdb
    src = lmdb.open(a.src, readonly=True, lock=False)
    dst = lmdb.open(a.dst, map_size=int(a.map_size_gb * (1 << 30)))
    n = src.stat()["entries"]
    txn_r = src.begin(buffers=True)
    pending = []

    def put(i, blob):
        pending.append((str(i).encode(), blob))
        if len(pending) >= 32:
            with dst.begin(write=True) as t:
                for k, v in pending:
                    t.put(k, v)
            pending.clear()
    nin, nout = recode_store(lambda i: txn_r.get(str(i).encode()), n, put, a.workers, sparse_ego=a.sparse_ego)
    if pending:
        with dst.begin(write=True) as t:
            for k, v in pending:
                t.put(k, v)
    print(f"{n} records: {nin / 1e9:.2f} GB compressed -> {nout / 1e9:.2f} GB raw")


if __name__ == "__main__":
    main()
