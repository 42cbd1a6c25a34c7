"""In-process fallback of the update's recurrent core after a persistent-kernel timeout (round 6).

The two GRU recurrences and the instruction LSTM are persistent kernels whose cooperating workgroups wait for each other with BOUNDED
spins (csrc/wsmg_rnn.hip); the chained recurrent core adds kernels that spin on device counters (wsmgmap/recurrent.py).  A workgroup
that does not become resident in time — beside a collective library's ring kernels, on a CU-masked device, beside another process —
makes a spin run out: the kernel fills its outputs with NaN and sets a bit in a host-mapped status word.  `RecurrentCoreFallback`
turns that from a fatal error into a slower configuration, on every rank of a process group TOGETHER (one 4-byte MAX all-reduce), and
without ever replacing the process (a process that has touched the GPU must not exec):

    level 0   what the policy was built with (default: the chained core, three streams)
    level 1   the staged core: one persistent kernel at a time, no device-side chaining, no decoder side stream
    level 2   the stock (MIOpen) GRU / LSTM: no persistent kernel at all

Reference call sites replaced by those kernels: mg_map_policy.py:220-227,242-249, instruction_encoder.py:80-92; the data-parallel
wrapper whose collectives the ranks must stay in step for: common_trainer.py:35-38,61-66.
"""
import copy
import sys

import torch
import torch.distributed as dist

from . import _abi, debug
from .parallel import GradExchangeError


class RecurrentCoreFallback:
    def __init__(self, policy, optimizer=None, reducer=None, group=None, verbose=True):
        self.policy, self.optimizer, self.reducer, self.group, self.verbose = policy, optimizer, reducer, group, verbose
        net = policy.net
        self.level = 0
        self.reasons = []
        chained = getattr(net, "recurrent_chunks", 0) > 0
        self.levels = [("chained (one launch per recurrence, device-side chunk counters)" if chained and debug.sw.recurrent_chain
                        else "pipelined chunk launches" if chained else "staged"),
                       "staged (fallback: one persistent kernel at a time, no chaining, no decoder side stream)",
                       "stock MIOpen GRU / LSTM (fallback: no persistent kernel)"]
        self.read_status = _abi.take_rnn_status      # (a test substitutes its own)
        self._multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1 or (
            reducer is not None and not getattr(reducer, "_off", True))
        self.snapshot()

    def snapshot(self):
        """The state a fallback returns to (a timeout's NaN may have reached the parameters through an optimizer step)."""
        self._snap = ({k: v.detach().clone() for k, v in self.policy.state_dict().items()},
                      copy.deepcopy(self.optimizer.state_dict()) if self.optimizer is not None else None)

    @property
    def name(self):
        return self.levels[self.level]

    def agree(self, flag):
        """True on every rank if ANY rank passes True."""
        if not (dist.is_available() and dist.is_initialized()):
            return bool(flag)
        dev = next(self.policy.parameters()).device
        t = torch.tensor([1.0 if flag else 0.0], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return bool(t.item() > 0)

    def apply(self, why):
        """Switch to the next level (every rank calls this after `agree` said so), restore the snapshot, reset the exchange."""
        self.level += 1
        self.reasons.append(str(why)[:300])
        if self.level == 1:
            self.policy.net.recurrent_chunks = 0
            debug.sw.decoder_streams = 0
        elif self.level == 2:
            debug.sw.rnn_stock = True
        else:
            raise RuntimeError("persistent-kernel timeouts persist without any persistent kernel: " + "; ".join(self.reasons))
        if self.verbose:
            print("wsmgmap: persistent-kernel timeout (%s) -> recurrent core: %s" % (str(why)[:200], self.name), file=sys.stderr)
        if torch.cuda.is_available() and next(self.policy.parameters()).is_cuda:
            torch.cuda.synchronize()
            _abi.take_rnn_status()
        if self.optimizer is not None:
            self.optimizer.zero_grad(set_to_none=True)
            self.optimizer.load_state_dict(copy.deepcopy(self._snap[1]))
        self.policy.load_state_dict(self._snap[0])
        if self.reducer is not None:
            self.reducer.reset()
            self.reducer._flag_pending = False
            self.reducer.broadcast_parameters(self.policy)

    def guarded(self, phase):
        """Run `phase()` — whole updates — until it completes without a timeout on any rank; -> its result."""
        while True:
            err, out = None, None
            try:
                out = phase()
                if torch.cuda.is_available() and next(self.policy.parameters()).is_cuda:
                    torch.cuda.synchronize()
                if self.reducer is not None:
                    self.reducer.check()
            except GradExchangeError as e:        # raised by every rank at the same finish(): the ranks are in step
                err = str(e)
            except _abi.WsmgError as e:           # single process only (under a process group the forward's checks are deferred)
                if "timed out" not in str(e) or self._multi:
                    raise
                err = str(e)
            bits = self.read_status()
            if bits and err is None:
                err = "status word: " + ", ".join(_abi.status_names(bits))
            if not self.agree(err is not None):
                return out
            self.apply(err or "a peer rank reported a timeout")

    def report(self):
        return dict(recurrent_core=self.name, fallback_level=self.level, reasons=list(self.reasons))
