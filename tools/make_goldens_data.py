#!/usr/bin/env python3
"""Capture golden vectors of the trajectory-cache path from the UNMODIFIED reference file
vlnce_baselines/dagger_trainer.py: collate_fn, _block_shuffle and IWTrajectoryDataset (sharding, block shuffle,
length-sorted preload).  Build container only.  The module's heavy imports (habitat, lmdb, msgpack_numpy, the
trainer base classes) are replaced by empty stand-in modules in sys.modules: none of them is executed by the
three objects captured here, except lmdb / msgpack_numpy / zlib inside IWTrajectoryDataset._load_next, which
are given an in-memory dict of pickled records (the codec itself is NOT captured from the reference:
msgpack_numpy is not installed here; oracle/data_ref.py restates its wire format, parity unpinned).

    python tools/make_goldens_data.py
"""
import hashlib
import importlib.util
import os
import pickle
import random
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import data_cases as dc  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
RECORDS = {}   # the fake LMDB: key bytes -> value bytes


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Txn:
    def __enter__(self): return self
    def __exit__(self, *a): return False
    def get(self, key): return RECORDS[bytes(key)]


class _Env:
    def __enter__(self): return self
    def __exit__(self, *a): return False
    def stat(self): return {"entries": len(RECORDS)}
    def begin(self, **kw): return _Txn()


def install_stubs():
    class _Registry:
        @staticmethod
        def register_trainer(name=None):
            return lambda cls: cls
    _mod("lmdb", open=lambda *a, **k: _Env())
    _mod("msgpack_numpy", unpackb=lambda b, raw=False: pickle.loads(b), packb=lambda d, use_bin_type=True: pickle.dumps(d))
    _mod("habitat", Config=dict, logger=types.SimpleNamespace(info=lambda *a, **k: None))
    _mod("habitat_baselines"); _mod("habitat_baselines.common")
    _mod("habitat_baselines.common.baseline_registry", baseline_registry=_Registry)
    _mod("habitat_baselines.common.environments", get_env_class=None)
    _mod("habitat_baselines.common.tensorboard_utils", TensorboardWriter=None)
    _mod("habitat_baselines.common.utils", batch_obs=None)
    _mod("vlnce_baselines"); _mod("vlnce_baselines.common")
    _mod("vlnce_baselines.common_trainer", CommonTrainer=object)
    _mod("vlnce_baselines.common.env_utils", construct_envs=None)
    _mod("vlnce_baselines.common.aux_losses", AuxLosses=None)
    _mod("vlnce_baselines.common.utils", transform_obs=None)
    import zlib as _z
    # records in the fake LMDB are stored uncompressed: zlib.decompress must pass them through
    sys.modules["zlib"] = types.SimpleNamespace(decompress=lambda b: bytes(b), compress=_z.compress)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    install_stubs()
    spec = importlib.util.spec_from_file_location("vlnce_baselines.dagger_trainer",
                                                  "/root/reference/vlnce_baselines/dagger_trainer.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    out = {}

    # ---- G6: collate_fn on ragged batches (and the 200-step cap)
    def sample(seed, length):
        obs, prev, oracle = dc.episode(seed, length)
        obs = {k: torch.from_numpy(v.copy()) for k, v in obs.items()}
        w = torch.ones(length)
        return obs, torch.from_numpy(prev.copy()), torch.from_numpy(oracle.copy()), w
    for tag, lengths in (("rag", dc.COLLATE_LENGTHS), ("long", dc.LONG_LENGTHS)):
        batch = [sample(100 + i, n) for i, n in enumerate(lengths)]
        ob, prev, masks, corr, wts = ref.collate_fn(batch)
        for k, v in ob.items():
            out[f"g6_{tag}_obs_{k}_shape"] = np.array(v.shape)
            out[f"g6_{tag}_obs_{k}_dtype"] = np.array(str(v.dtype))
            out[f"g6_{tag}_obs_{k}_sha"] = np.array(sha(v.float().numpy()))   # what the trainer ships: v.float()
        for name, v in (("prev", prev), ("masks", masks), ("corr", corr), ("wts", wts)):
            out[f"g6_{tag}_{name}_shape"] = np.array(v.shape)
            out[f"g6_{tag}_{name}_sha"] = np.array(sha(v.float().numpy()))
        if tag == "rag":
            out["g6_rag_progress"] = ob["progress"].numpy()
            out["g6_rag_masks"] = masks.numpy()
            out["g6_rag_wts"] = wts.numpy()

    # ---- G7: IWTrajectoryDataset ordering (rank / worker sharding, block shuffle, length-sorted preload)
    RECORDS.clear()
    for i, n in enumerate(dc.DATASET_LENGTHS):
        obs, prev, oracle = dc.episode(1000 + i, n)
        RECORDS[str(i).encode()] = pickle.dumps([obs, prev, oracle])
    for ci, (world, rank, nworkers, wid, bs, seed) in enumerate(dc.DATASET_CASES):
        ds = ref.IWTrajectoryDataset("unused", use_iw=True, inflection_weight_coef=3.2, batch_size=bs, rank=rank, world_size=world)
        info = None if nworkers == 0 else types.SimpleNamespace(num_workers=nworkers, id=wid)
        torch.utils.data.get_worker_info = lambda info=info: info
        random.seed(seed)
        lens, wsum, first = [], [], []
        for obs, prev, oracle, w in ds:   # ONE __iter__ call per epoch, as the DataLoader fetcher does
            lens.append(prev.shape[0]); wsum.append(float(w.sum())); first.append(float(prev[0, 0]))
        out[f"g7_{ci}_order"] = np.array(ds._preload_index)
        out[f"g7_{ci}_yield_lengths"] = np.array(lens)
        out[f"g7_{ci}_weight_sums"] = np.array(wsum)
        out[f"g7_{ci}_first_prev"] = np.array(first, dtype=np.float64)
        out[f"g7_{ci}_len"] = np.array(len(ds))
    random.seed(3)
    out["g7_block_shuffle"] = np.array(ref._block_shuffle(list(range(17)), 4))
    np.savez_compressed(os.path.join(OUT, "g6_g7_data.npz"), **out)
    print("wrote g6_g7_data.npz with", len(out), "entries")


if __name__ == "__main__":
    main()
