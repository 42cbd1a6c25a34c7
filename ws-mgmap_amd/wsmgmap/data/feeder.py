"""Prefetching feeder: decode in DataLoader worker processes (as the reference does, dagger_trainer.py:560-575), batch
assembly on the GPU (`DeviceCollator`) on a side stream, `prefetch` batches ahead of the consumer.  What the consumer
gets per iteration is exactly what the reference's training loop hands to `_update_agent` (:606-625)."""
import collections

import torch

from .collate import DeviceCollator


def _identity(batch):
    return batch


class DeviceFeeder:
    def __init__(self, dataset, batch_size, device="cuda", num_workers=0, prefetch=2):
        self.dataset, self.batch_size, self.device = dataset, batch_size, torch.device(device)
        self.num_workers, self.prefetch = num_workers, max(1, prefetch)

    def __iter__(self):
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

    def _hand_over(self, out, ev):
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        obs, prev, masks, corr, wts = out
        for t in list(obs.values()) + [prev, masks, corr, wts]:
            t.record_stream(cur)
        return out
