"""GPU tests, round 4 (VERDICT r03 "Next round" + ADVICE r03): feeder epochs / oversize batches, the data-parallel exchange beside
the persistent RNN kernels over RCCL, and the parity of the kernels added this round (each next to its section below)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _aux_losses_off():
    from wsmgmap.common.aux_losses import AuxLosses
    AuxLosses.deactivate()
    AuxLosses.clear()
    yield
    AuxLosses.deactivate()
    AuxLosses.clear()


class _Store:
    """picklable in-memory record store (stands in for the reference's LMDB)"""
    def __init__(self, blobs):
        self.blobs = blobs

    def __call__(self, i):
        return self.blobs[i]


# ----------------------------------------------------------------------------- feeder (ADVICE r03, medium x 2)
def test_feeder_ring_reshuffles_every_epoch_and_survives_oversize_batches():
    """(1) The ring's decode workers are re-seeded per epoch (a fresh base seed per iterator, as DataLoader draws one:
    dagger_trainer.py:116-119,585-594), so the batch order differs from epoch to epoch; an explicit seed reproduces a run.
    (2) A batch that does not fit the ring slot arrives anyway (its own shared block), with the same values."""
    from oracle import data_cases as dc
    from wsmgmap.data import TrajectoryDataset, DeviceFeeder, pack_record
    lengths = (dc.DATASET_LENGTHS * 3)[:64]
    store = _Store([pack_record(*dc.episode(1000 + i, n)) for i, n in enumerate(lengths)])
    mk = lambda: TrajectoryDataset(store, len(store.blobs), batch_size=2, rank=0, world_size=1)  # noqa: E731

    def epoch(fd):
        order, vals = [], {}
        for ob, prev, masks, corr, wts in fd:
            firsts = prev.view(-1, 2, 2)[0, :, 0].cpu().tolist()
            order += firsts
            for n, f in enumerate(firsts):
                vals[f] = float(ob["progress"].view(-1, 2, 1)[:, n].sum())
        return order, vals

    fd = DeviceFeeder(mk(), 2, "cuda", num_workers=2, prefetch=2)
    o1, v1 = epoch(fd)
    o2, v2 = epoch(fd)
    assert fd.epoch == 2 and sorted(o1) == sorted(o2) and o1 != o2, "the same batch order in two epochs"
    a, _ = epoch(DeviceFeeder(mk(), 2, "cuda", num_workers=2, prefetch=2, seed=9))
    b, _ = epoch(DeviceFeeder(mk(), 2, "cuda", num_workers=2, prefetch=2, seed=9))
    assert a == b
    fs = DeviceFeeder(mk(), 2, "cuda", num_workers=2, prefetch=2, seed=9)
    e1, _ = epoch(fs)
    e2, _ = epoch(fs)
    assert e1 == a and e2 != e1
    # oversize: slots of 4 KiB hold only the shortest batches
    small = DeviceFeeder(mk(), 2, "cuda", num_workers=2, prefetch=2, seed=9, slot_bytes=4096)
    c, vc = epoch(small)
    _, va = epoch(DeviceFeeder(mk(), 2, "cuda", num_workers=2, prefetch=2, seed=9))
    assert small.oversize_batches > 0 and c == a and vc == va
