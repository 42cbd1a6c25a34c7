"""Record codec of the DAgger trajectory cache.

A record is `[observations: {sensor: ndarray[T, ...]}, prev_actions: ndarray[T, 2], oracle_actions: ndarray[T, 2]]`
stored as `zlib(msgpack_numpy(record))` (reference: dagger_trainer.py:36-37 write, :177-179 read).  msgpack_numpy
is a thin layer over msgpack: an ndarray is the map {b"nd": True, b"type": dtype.str, b"kind": b"", b"shape":
[...], b"data": raw bytes}; this module speaks that wire format with plain `msgpack`, so caches written by the
reference load here and vice versa, without the extra dependency.
"""
import zlib

import msgpack
import numpy as np

# on-disk dtypes, common_trainer.py:514-532
DISK_DTYPES = {
    "vln_oracle_action_sensor": np.uint8,
    "rgb_ego_map": np.float16,
    "gt_path": np.float16,
    "rgb": np.uint8,
    "depth": np.float16,
    "rgb_features": np.float16,
    "depth_features": np.float16,
    "gt_semantic_map": np.int64,
}


def change_data_type(traj_obs):
    """Cast a trajectory's observations to their on-disk dtypes (the reference's CommonTrainer.change_data_type)."""
    out = {}
    for k, v in traj_obs.items():
        a = v.numpy() if hasattr(v, "numpy") else np.asarray(v)
        out[k] = a.astype(DISK_DTYPES[k]) if k in DISK_DTYPES else a
    return out


def _encode(obj):
    if isinstance(obj, np.ndarray):
        if obj.dtype.kind == "V":
            raise TypeError("structured arrays are not part of the trajectory cache")
        a = np.ascontiguousarray(obj)
        return {b"nd": True, b"type": a.dtype.str, b"kind": b"", b"shape": list(a.shape), b"data": a.tobytes()}
    if isinstance(obj, np.generic):
        return {b"nd": False, b"type": obj.dtype.str, b"data": obj.tobytes()}
    raise TypeError(f"cannot pack {type(obj)}")


def _decode(obj):
    nd = obj.get(b"nd", obj.get("nd"))
    if nd is None:
        return obj
    get = lambda k: obj[k] if k in obj else obj[k.decode()]  # noqa: E731  (raw=False readers see str keys for old files)
    dt = get(b"type")
    dt = np.dtype(dt.decode() if isinstance(dt, bytes) else dt)
    if nd:
        return np.frombuffer(get(b"data"), dtype=dt).reshape(get(b"shape"))
    return np.frombuffer(get(b"data"), dtype=dt)[0]


def pack_record(observations, prev_actions, oracle_actions, level=-1):
    payload = msgpack.packb([observations, prev_actions, oracle_actions], default=_encode, use_bin_type=True)
    return zlib.compress(payload, level)


# ---- the recoded ("raw") record: the same arrays, uncompressed, behind a small index ------------------------------------------
# The reference's value format costs one zlib inflate + one msgpack parse of 92 MB per episode on every epoch: 228 policy steps/s
# per host core (profiles/r03_feeder.txt), i.e. ~200 cores of decode to keep ONE MI355X fed (the update consumes > 45 k steps/s).
# `tools/recode_cache.py` rewrites a cache ONCE into this layout; reading a record is then a header parse and zero-copy views
# into the stored bytes — the arrays keep their on-disk dtypes (common_trainer.py:514-532), so a recoded record collates
# bit-identically to the original (tests/test_host_cpu.py, tests/test_gpu_round4.py).
#
#   bytes 0-7   b"WSMGRAW1"
#   bytes 8-11  little-endian uint32: length H of the index
#   12 .. 12+H  msgpack [[name, dtype.str, shape, offset, nbytes] ...] for the observations in their stored order, then the
#               entries "__prev" and "__oracle" (prev_actions, oracle_actions); offsets are relative to the payload
#   payload     starts at the next multiple of 64 bytes; every array starts on a multiple of 64 bytes
RAW_MAGIC = b"WSMGRAW1"


# ---- sparse ego map (round 5) ------------------------------------------------------------------------------------------------------
# `rgb_ego_map` is 89 % of a step's bytes (float16 [64, 100, 100] = 1.28 MB of 1.44) and it is a post-ReLU feature map scattered
# into a mostly empty grid: 55-80 % of its elements are exactly zero.  A raw record may therefore hold it as
#     rgb_ego_map__bits  uint8  [T, H*W, C/8]   one presence bit per (pixel, channel), pixel-major (channels-last): channel c of pixel p
#                                               is bit (c % 8) of byte c / 8
#     rgb_ego_map__off   uint32 [T, H*W]        non-zeros of the step in front of pixel p
#     rgb_ego_map__base  int64  [T + 1]         non-zeros of the episode in front of step t
#     rgb_ego_map__vals  float16 [nnz]          the non-zero values, step by step, pixel by pixel, channel by channel
#     rgb_ego_map__shape int64  [3]             C, H, W
# (-0.0 counts as non-zero: the expansion is bit-exact.)  `wsmg_collate_ego_sparse_nhwc_bf16` expands it on the device straight into
# the padded channels-last bf16 tensor the bf16 policy reads — one wave per pixel: the pixel's 8 presence bytes are one word, lane c's
# value sits at base + off + popcount(word & lanes below c): a coalesced gather.  PCIe bytes per step at 30 % non-zeros: 0.12 (bits
# + offsets) + 0.38 (values) MB instead of 1.28.
SPARSE_EGO = "rgb_ego_map"
SPARSE_SUFFIXES = ("__bits", "__off", "__base", "__vals", "__shape")


def sparse_pack_ego(ego):
    """float16 [T, C, H, W] -> the five arrays above (C == 64: one 8-byte word of presence bits per pixel)."""
    a = np.ascontiguousarray(np.asarray(ego))
    if a.dtype != np.float16 or a.ndim != 4 or a.shape[1] != 64:
        raise TypeError(f"sparse ego map: float16 [T, 64, H, W] expected, got {a.dtype} {a.shape}")
    T, C, H, W = a.shape
    nhwc = np.ascontiguousarray(a.transpose(0, 2, 3, 1)).reshape(T, H * W, C)
    present = nhwc.view(np.uint16) != 0                       # the bit pattern: -0.0 is kept
    bits = np.packbits(present, axis=2, bitorder="little")      # [T, HW, C/8]
    per_pixel = present.sum(axis=2, dtype=np.int64)             # [T, HW]
    off = np.zeros((T, H * W), dtype=np.uint32)
    if H * W > 1:
        off[:, 1:] = np.cumsum(per_pixel[:, :-1], axis=1)
    base = np.zeros(T + 1, dtype=np.int64)
    base[1:] = np.cumsum(per_pixel.sum(axis=1))
    vals = nhwc[present]                                        # C order: step, pixel, channel
    return {SPARSE_EGO + "__bits": bits, SPARSE_EGO + "__off": off, SPARSE_EGO + "__base": base,
            SPARSE_EGO + "__vals": np.ascontiguousarray(vals), SPARSE_EGO + "__shape": np.array([C, H, W], dtype=np.int64)}


def sparse_expand_ego(obs, steps=None):
    """The dense float16 [T, C, H, W] map of a record that holds the sparse form (host route: tests, the reference-semantics collate)."""
    C, H, W = (int(x) for x in obs[SPARSE_EGO + "__shape"])
    bits = np.asarray(obs[SPARSE_EGO + "__bits"])
    T = bits.shape[0] if steps is None else min(int(steps), bits.shape[0])
    present = np.unpackbits(bits[:T], axis=2, bitorder="little").astype(bool)   # [T, HW, C]
    nhwc = np.zeros((T, H * W, C), dtype=np.float16)
    nhwc[present] = np.asarray(obs[SPARSE_EGO + "__vals"])[:int(np.asarray(obs[SPARSE_EGO + "__base"])[T])]
    return np.ascontiguousarray(nhwc.reshape(T, H, W, C).transpose(0, 3, 1, 2))


def has_sparse_ego(obs):
    return SPARSE_EGO + "__bits" in obs


def densify(obs):
    """observations with the sparse ego map expanded back to `rgb_ego_map` (a copy of the dict; dense records pass through)."""
    if not has_sparse_ego(obs):
        return obs
    out = {k: v for k, v in obs.items() if not k.startswith(SPARSE_EGO + "__")}
    out[SPARSE_EGO] = sparse_expand_ego(obs)
    return out


def pack_record_raw(observations, prev_actions, oracle_actions, sparse_ego=False):
    """The recoded form of one record (see above).  Arrays are stored as they are (no dtype change: cast first with
    change_data_type, as the reference does before it writes).  sparse_ego: store a float16 [T, 64, H, W] `rgb_ego_map` in the
    sparse form."""
    if sparse_ego and SPARSE_EGO in observations:
        e = np.asarray(observations[SPARSE_EGO])
        if e.dtype == np.float16 and e.ndim == 4 and e.shape[1] == 64:
            observations = {k: v for k, v in observations.items() if k != SPARSE_EGO}
            observations.update(sparse_pack_ego(e))
    items = [(k, np.ascontiguousarray(np.asarray(v))) for k, v in observations.items() if k != "ep_id"]
    items += [("__prev", np.ascontiguousarray(np.asarray(prev_actions))), ("__oracle", np.ascontiguousarray(np.asarray(oracle_actions)))]
    index, off = [], 0
    for k, a in items:
        if a.dtype.kind == "V" or a.dtype.kind == "O":
            raise TypeError(f"cannot store {k} of dtype {a.dtype} in a raw record")
        index.append([k, a.dtype.str, list(a.shape), off, a.nbytes])
        off += (a.nbytes + 63) & ~63
    head = msgpack.packb(index, use_bin_type=True)
    start = (12 + len(head) + 63) & ~63
    out = bytearray(start + off)
    out[0:8] = RAW_MAGIC
    out[8:12] = len(head).to_bytes(4, "little")
    out[12:12 + len(head)] = head
    for (k, a), (_, _, _, o, nb) in zip(items, index):
        out[start + o:start + o + nb] = a.reshape(-1).view(np.uint8).data
    return bytes(out)


def is_raw_record(blob):
    return len(blob) >= 12 and bytes(blob[0:8]) == RAW_MAGIC


def _unpack_raw(blob):
    """Views straight into `blob` — zero copy.  LIFETIME: the arrays are valid as long as `blob` is; a `get` that hands back a
    transaction-scoped buffer (lmdb with `buffers=True`) must keep the transaction open until the batch has been collated, or
    hand back bytes.  The stored index is validated against the blob before any view is made."""
    mv = memoryview(blob)
    total = len(mv)
    hlen = int.from_bytes(mv[8:12], "little")
    if 12 + hlen > total:
        raise ValueError(f"raw trajectory record truncated: index of {hlen} bytes in a record of {total}")
    index = msgpack.unpackb(bytes(mv[12:12 + hlen]), raw=False)
    start = (12 + hlen + 63) & ~63
    obs, prev, oracle = {}, None, None
    for name, dt, shape, off, nbytes in index:
        dtype = np.dtype(dt)
        want = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        if off < 0 or nbytes != want or start + off + nbytes > total:
            raise ValueError(f"raw trajectory record corrupt or truncated: entry {name!r} ({dt}, shape {list(shape)}, {nbytes} bytes at "
                             f"payload offset {off}) does not fit a record of {total} bytes")
        a = np.frombuffer(mv, dtype=dtype, count=nbytes // dtype.itemsize, offset=start + off).reshape(shape)
        if name == "__prev":
            prev = a
        elif name == "__oracle":
            oracle = a
        else:
            obs[name] = a
    return [obs, prev, oracle]


def recode_record(blob, level=None, sparse_ego=False):
    """zlib(msgpack_numpy) value -> raw value (a raw value is returned unchanged).  sparse_ego: with the ego map in the sparse form."""
    if is_raw_record(blob):
        return bytes(blob)
    return pack_record_raw(*unpack_record(blob), sparse_ego=sparse_ego)


def unpack_record(blob):
    """-> [observations, prev_actions, oracle_actions]; arrays are read-only views of the decompressed buffer
    (no copy), which is what the pinned staging of DeviceCollator copies from.  Accepts the reference's value format
    (zlib(msgpack_numpy)) and the recoded raw format (views straight into `blob`)."""
    if is_raw_record(blob):
        return _unpack_raw(blob)
    rec = msgpack.unpackb(zlib.decompress(blob), object_hook=_decode, raw=False, strict_map_key=False)
    if isinstance(rec[0], dict):
        rec[0].pop("ep_id", None)   # dagger_trainer.py:180-181
    return rec
