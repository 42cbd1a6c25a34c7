"""One teacher-forcing update as a HIP graph.

The reference's update (`DaggerTrainer._update_agent`, dagger_trainer.py:505-543) is
`optimizer.zero_grad(); pred, aux = policy(...); loss = ...; loss.backward(); optimizer.step()` — here ≈460 kernel launches of
which ≈250 are 2-6 µs long (heads, auxiliary losses, gradient bookkeeping): the host enqueues an update in 9 ms against 12.5 ms
of GPU time, so the GPU never starves on average, but inside the small-launch regions it does (50-140 µs gaps under a
profiler).  `GraphedUpdate` captures the whole update once per input signature (`torch.cuda.graph`: the custom launches, the side
streams with their event joins and the persistent RNN kernels all capture) and replays it; the only eager work left per update
is the instruction dedup, whose result (number of unique instructions, longest length) fixes the shapes the graph was captured
for.

Requirements: CUDA tensors that stay at the same addresses from call to call (a tensor at a new address is copied into the
captured one — correct, but the cached ego map alone is 1.3 GB per update); an optimizer whose step does not read host state
(`wsmgmap.optim.Adam(capturable=True)` or `torch.optim.Adam(capturable=True)`); one process (no gradient all-reduce inside
the graph)."""
import torch

from . import _abi, ops
from .common.aux_losses import AuxLosses


class GraphedUpdate:
    def __init__(self, policy, optimizer, loss_fn, eager_calls=3):
        """loss_fn(pred, aux_loss, observations, weights) -> scalar loss tensor.  The first `eager_calls` updates run eagerly
        (first-call initialisations of the library and of the allocator must not happen under capture)."""
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            raise _abi.WsmgError("GraphedUpdate: data-parallel updates are not captured (the gradient exchange stays eager)")
        self.policy, self.optimizer, self.loss_fn = policy, optimizer, loss_fn
        self.eager_calls = max(1, int(eager_calls))   # at least one: the optimizer's device-side step counters are created eagerly
        self.calls = 0
        self._graphs = {}
        self._dd_stream = None
        self._stream = None    # the eager updates run on a side stream: autograd's AccumulateGrad nodes remember the stream they
        #                        were created on, and one that lives on the default stream breaks a later capture

    def _release_old_graphs(self):
        """Drop every reference to an autograd graph of an earlier (default-stream) update: the tensors the forward pass leaves
        on the modules (`policy.prog`, `net.att_map_t_m`, ...) and the registered auxiliary losses.  While such a graph lives,
        the parameters' AccumulateGrad nodes — bound to the stream of their first use — are re-used by the next forward pass,
        and one bound to the default stream makes the end of a capture crash."""
        AuxLosses.clear()
        for p in self.policy.parameters():
            p.grad = None
        for m in self.policy.modules():
            for k, v in list(vars(m).items()):
                if torch.is_tensor(v) and v.grad_fn is not None:
                    setattr(m, k, v.detach())
                elif not torch.is_tensor(v) and hasattr(v, "check_none_pending"):   # ops.TokenGradSink of the last forward
                    setattr(m, k, None)
        # anything left (a loss, predictions or observations dict kept by the caller ...) would end in a crash at the end of
        # the capture: say so instead
        import gc
        gc.collect()
        left = [o for o in gc.get_objects() if torch.is_tensor(o) and o.is_cuda and o.grad_fn is not None]
        if left:
            what = ", ".join(f"{tuple(t.shape)} <- {type(t.grad_fn).__name__}" for t in left[:6])
            raise _abi.WsmgError(f"GraphedUpdate: {len(left)} tensor(s) of an earlier autograd graph are still referenced ({what}); "
                                 "drop them (del loss, predictions, ...) before the first graphed update")

    # -- the update itself (what is captured) ------------------------------------------------------------------------------
    def _update(self, obs, h_in, prev_actions, masks, weights):
        self.optimizer.zero_grad(set_to_none=True)
        AuxLosses.clear()
        h = h_in.clone()      # the policy overwrites its hidden-state argument in place (reference contract)
        pred, aux = self.policy(obs, h, prev_actions, masks, weights)
        loss = self.loss_fn(pred, aux, obs, weights)
        loss.backward()
        self.optimizer.step()
        return loss.detach(), h.detach()

    @staticmethod
    def _signature(obs, h, prev_actions, masks, weights, dd):
        def sig(t):
            return (tuple(t.shape), t.dtype, t.device)
        return (tuple(sorted((k, sig(v)) for k, v in obs.items() if torch.is_tensor(v))), sig(h), sig(prev_actions), sig(masks), sig(weights),
                int(dd[0].shape[0]), int(dd[2].max()))

    def __call__(self, observations, rnn_hidden_states, prev_actions, masks, weights):
        """-> loss (a tensor owned by the graph: valid until the next call).  rnn_hidden_states is overwritten with the final
        hidden state, as `BasePolicy.forward` does."""
        self.calls += 1
        enc = self.policy.net.instruction_encoder
        # eager, on its own stream: the dedup's host read-back must not wait for the previous replay (it depends on the
        # instruction tokens only), or the GPU idles while the host prepares the next one
        if self._dd_stream is None:
            self._dd_stream = torch.cuda.Stream()
        cur0 = torch.cuda.current_stream()
        if self.calls == 1:
            self._dd_stream.wait_stream(cur0)            # the caller's tensors are complete there (first call only: later calls
        with torch.cuda.stream(self._dd_stream):         # assume the tokens were written before the previous update returned)
            dd = enc.dedup(observations["instruction"])
        cur0.wait_stream(self._dd_stream)
        for t in dd:
            if t.is_cuda:
                t.record_stream(cur0)    # consumed on the caller's stream (copied into the captured tensors, behind the previous
                #                          replay): the allocator must not hand its memory to the NEXT call's dedup before that
        if self.calls <= self.eager_calls:
            obs = dict(observations)
            obs["instruction_dedup"] = dd
            if self._stream is None:
                self._stream = torch.cuda.Stream()
                self._release_old_graphs()
            cur = torch.cuda.current_stream()
            self._stream.wait_stream(cur)
            with torch.cuda.stream(self._stream):
                loss, h = self._update(obs, rnn_hidden_states, prev_actions, masks, weights)
                rnn_hidden_states.copy_(h)
            cur.wait_stream(self._stream)
            return loss
        key = self._signature(observations, rnn_hidden_states, prev_actions, masks, weights, dd)
        g = self._graphs.get(key)
        if g is None:
            g = self._capture(observations, rnn_hidden_states, prev_actions, masks, weights, dd)
            self._graphs[key] = g
        # inputs: same storage -> nothing to do; new storage -> copy into the captured tensors
        for k, s in g["obs"].items():
            v = observations[k]
            if v.data_ptr() != s.data_ptr():
                s.copy_(v)
        for s, v in ((g["h_in"], rnn_hidden_states), (g["prev"], prev_actions), (g["masks"], masks), (g["weights"], weights),
                     (g["dd"][0], dd[0]), (g["dd"][1], dd[1]), (g["dd"][3], dd[3])):
            if v.data_ptr() != s.data_ptr():
                s.copy_(v)
        g["graph"].replay()
        if hasattr(self.optimizer, "note_replayed_steps"):
            self.optimizer.note_replayed_steps(1)
        rnn_hidden_states.copy_(g["h_out"])
        return g["loss"]

    def _capture(self, observations, rnn_hidden_states, prev_actions, masks, weights, dd):
        obs_s = {k: v for k, v in observations.items() if torch.is_tensor(v)}
        dd_s = (dd[0].clone(), dd[1].clone(), dd[2].clone(), dd[3].clone())   # owned copies: the eager dedup returns fresh tensors
        h_in = rnn_hidden_states.clone()
        obs_c = dict(obs_s)
        obs_c["instruction_dedup"] = dd_s
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            loss, h_out = self._update(obs_c, h_in, prev_actions, masks, weights)
        if hasattr(self.optimizer, "note_replayed_steps"):
            self.optimizer.note_replayed_steps(-1)   # capture ran the optimizer's host bookkeeping once without executing anything
        ops.check_rnn_status()
        return dict(graph=graph, obs=obs_s, dd=dd_s, h_in=h_in, prev=prev_actions, masks=masks, weights=weights, loss=loss, h_out=h_out)
