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


def unpack_record(blob):
    """-> [observations, prev_actions, oracle_actions]; arrays are read-only views of the decompressed buffer
    (no copy), which is what the pinned staging of DeviceCollator copies from."""
    rec = msgpack.unpackb(zlib.decompress(blob), object_hook=_decode, raw=False, strict_map_key=False)
    if isinstance(rec[0], dict):
        rec[0].pop("ep_id", None)   # dagger_trainer.py:180-181
    return rec
