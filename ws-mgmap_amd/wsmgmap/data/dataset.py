"""Sharded, block-shuffled, length-sorted iteration over a trajectory cache (IWTrajectoryDataset,
dagger_trainer.py:122-238) over any `get(index) -> bytes` store (the reference's store is an LMDB whose keys are
str(index).encode(); the store is injected so that this module does not depend on lmdb)."""
import math
import random

import numpy as np
import torch

from .codec import unpack_record


def block_shuffle(lst, block_size):
    """dagger_trainer.py:116-119 (uses the global `random`, like the reference)."""
    blocks = [lst[i:i + block_size] for i in range(0, len(lst), block_size)]
    random.shuffle(blocks)
    return [e for b in blocks for e in b]


def shard_range(length, rank, world_size, num_workers=0, worker_id=0):
    """[start, end) of this rank's (and DataLoader worker's) contiguous slice, and len(dataset) (:208-238)."""
    per_proc = int(math.floor(length / world_size))
    if num_workers == 0:
        return per_proc * rank, per_proc * rank + per_proc, per_proc
    per_worker = int(math.floor(per_proc / num_workers))
    start = per_worker * worker_id + per_proc * rank
    return start, start + per_worker, per_worker * num_workers


class TrajectoryDataset(torch.utils.data.IterableDataset):
    def __init__(self, get, length, use_iw=True, inflection_weight_coef=1.0, batch_size=1, rank=0, world_size=1):
        super().__init__()
        self.get, self.length = get, int(length)
        self.preload_size = batch_size
        self.rank, self.world_size = rank, world_size
        self.inflec_weights = torch.tensor([1.0, inflection_weight_coef if use_iw else 1.0])
        self._preload, self.load_ordering, self.loaded_indices = [], [], []

    _worker_override = None     # (num_workers, worker_id) when the decode workers are not DataLoader's (data.feeder's ring)

    def _shard(self):
        if self._worker_override is not None:
            return shard_range(self.length, self.rank, self.world_size, *self._worker_override)
        info = torch.utils.data.get_worker_info()
        return shard_range(self.length, self.rank, self.world_size, 0 if info is None else info.num_workers,
                           0 if info is None else info.id)

    def __len__(self):
        return self._shard()[2]

    def __iter__(self):
        start, end, _ = self._shard()
        self.load_ordering = list(reversed(block_shuffle(list(range(start, end)), self.preload_size)))
        self._preload, self.loaded_indices = [], []
        return self

    def _load_next(self):
        if not self._preload:
            if not self.load_ordering:
                raise StopIteration
            new, lengths = [], []
            for _ in range(min(self.preload_size, len(self.load_ordering))):
                i = self.load_ordering.pop()
                new.append(unpack_record(self.get(i)))
                lengths.append(len(new[-1][-1]))
                self.loaded_indices.append(i)
            prio = list(range(len(new)))
            random.shuffle(prio)                                   # ties between equal lengths are broken at random
            order = sorted(range(len(new)), key=lambda k: (lengths[k], prio[k]))
            self._preload = [new[k] for k in order]               # popped from the end: longest first
        return self._preload.pop()

    def __next__(self):
        obs, prev_actions, oracle_actions = self._load_next()
        # the reference indexes inflec_weights with an all-zero list (dagger_trainer.py:203-205): every step gets
        # weight inflec_weights[0] = 1.0 whatever the coefficient; reproduced as is
        weights = self.inflec_weights[torch.zeros(len(prev_actions), dtype=torch.long)]
        return ({k: np.asarray(v) for k, v in obs.items()}, np.asarray(prev_actions), np.asarray(oracle_actions), weights)
