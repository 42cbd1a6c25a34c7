"""TEST INFRASTRUCTURE (oracle) — CPU restatement of the trajectory-cache path in front of the policy update
(SURVEY 8f-2).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.

Follows vlnce_baselines/dagger_trainer.py:
  :36-37   compress_data           zlib.compress(msgpack_numpy.packb(record, use_bin_type=True))
  :40-113  collate_fn              time-major pad (observations with 1.0, the rest with 0) + stack over episodes
  :116-119 _block_shuffle
  :122-238 IWTrajectoryDataset     rank / worker sharding, block shuffle, length-sorted preload, weights
and common_trainer.py:514-532 (on-disk dtypes).  Pinned against tests/golden/g6_g7_data.npz, captured from the
unmodified reference by tools/make_goldens_data.py.

The record codec is third-party: msgpack_numpy (requirements.txt:6, no version pinned; wire format unchanged
since 0.4.4) on top of msgpack, neither vendored under /root/reference and msgpack_numpy not installed here.
Its published encoding of an ndarray is restated below; parity is UNPINNED for the codec (no reference-side
bytes exist to compare with) — the round trip and the format's field names are what the tests check.
"""
import random
import zlib

import msgpack
import numpy as np
import torch

LIMITED_LEN_BY_GPU = 200   # dagger_trainer.py:82


# ----------------------------------------------------------------------------- codec (msgpack_numpy wire format)
def _enc(obj):
    if isinstance(obj, np.ndarray):
        if obj.dtype.kind == "V":
            raise TypeError("structured arrays are not part of the trajectory cache")
        return {b"nd": True, b"type": obj.dtype.str, b"kind": b"", b"shape": list(obj.shape),
                b"data": obj.tobytes() if not obj.flags["C_CONTIGUOUS"] else bytes(obj.data)}
    if isinstance(obj, np.generic):
        return {b"nd": False, b"type": obj.dtype.str, b"data": obj.tobytes()}
    return obj


def _dec(obj):
    if b"nd" in obj:
        if obj[b"nd"] is True:
            return np.frombuffer(obj[b"data"], dtype=np.dtype(obj[b"type"])).reshape(obj[b"shape"])
        return np.frombuffer(obj[b"data"], dtype=np.dtype(obj[b"type"]))[0]
    return obj


def pack_record(obs, prev_actions, oracle_actions):
    return zlib.compress(msgpack.packb([obs, prev_actions, oracle_actions], default=_enc, use_bin_type=True))


def unpack_record(blob):
    return msgpack.unpackb(zlib.decompress(blob), object_hook=_dec, raw=False, strict_map_key=False)


def change_data_type(traj_obs):
    """common_trainer.py:514-532 (np.int there is int64)."""
    casts = {"vln_oracle_action_sensor": np.uint8, "rgb_ego_map": np.float16, "gt_path": np.float16, "rgb": np.uint8,
             "depth": np.float16, "rgb_features": np.float16, "depth_features": np.float16, "gt_semantic_map": np.int64}
    return {k: (np.asarray(v).astype(casts[k]) if k in casts else np.asarray(v)) for k, v in traj_obs.items()}


# ----------------------------------------------------------------------------- collate (dagger_trainer.py:40-113)
def _pad(t, max_len, fill):
    n = max_len - t.size(0)
    if n <= 0:
        return t[:max_len]
    return torch.cat([t, torch.full_like(t[0:1], fill).expand(n, *t.size()[1:])], dim=0)


def collate(batch):
    """batch: list of (obs dict of [T_i, ...] tensors, prev_actions [T_i,2], oracle [T_i,2], weights [T_i])."""
    obs_l, prev_l, corr_l, w_l = (list(x) for x in zip(*batch))
    T = min(max(p.size(0) for p in prev_l), LIMITED_LEN_BY_GPU)
    obs = {}
    for k in obs_l[0]:
        s = torch.stack([_pad(o[k], T, 1.0) for o in obs_l], dim=1)     # [T, N, ...]
        obs[k] = s.view(-1, *s.size()[2:])
    prev = torch.stack([_pad(p, T, 0) for p in prev_l], dim=1)
    corr = torch.stack([_pad(c, T, 0) for c in corr_l], dim=1)
    wts = torch.stack([_pad(w, T, 0) for w in w_l], dim=1)
    masks = torch.ones_like(wts, dtype=torch.float)
    masks[0] = 0
    return obs, prev.view(-1, 2), masks.view(-1, 1), corr, wts


# ----------------------------------------------------------------------------- dataset order (dagger_trainer.py:116-238)
def block_shuffle(lst, block_size):
    blocks = [lst[i:i + block_size] for i in range(0, len(lst), block_size)]
    random.shuffle(blocks)
    return [e for b in blocks for e in b]


def shard_range(length, rank, world_size, num_workers=0, worker_id=0):
    per_proc = int(np.floor(length / world_size))
    if num_workers == 0:
        return per_proc * rank, per_proc * rank + per_proc, per_proc
    per_worker = int(np.floor(per_proc / num_workers))
    start = per_worker * worker_id + per_proc * rank
    return start, start + per_worker, per_worker * num_workers


def dataset_order(lengths, rank, world_size, num_workers, worker_id, batch_size):
    """Yields (record index, in the order __next__ returns them); uses the global `random` exactly as the reference
    does (one shuffle of the blocks, then one shuffle of the tie-break priorities per preload)."""
    start, end, _ = shard_range(len(lengths), rank, world_size, num_workers, worker_id)
    ordering = list(reversed(block_shuffle(list(range(start, end)), batch_size)))
    loaded = []
    while ordering:
        idx = [ordering.pop() for _ in range(min(batch_size, len(ordering)))]
        loaded += idx
        prio = list(range(len(idx)))
        random.shuffle(prio)
        order = sorted(range(len(idx)), key=lambda k: (lengths[idx[k]], prio[k]))
        for k in reversed(order):        # _preload.pop() takes from the end: longest first
            yield idx[k]
    return loaded
