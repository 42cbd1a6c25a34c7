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


def _eager_dedup(owner, enc, tokens, first_call, reuse=False):
    """The instruction dedup on its own stream (its host read-back must not wait for the previous replay); the result tensors
    are handed to the caller's stream.  reuse: InstructionEncoder.dedup may return the kept result of equal tokens (the very
    same tuple: GraphedAct then skips the copies into its captured tensors)."""
    if owner._dd_stream is None:
        owner._dd_stream = torch.cuda.Stream()
    cur = torch.cuda.current_stream()
    if first_call:
        owner._dd_stream.wait_stream(cur)        # the caller's tensors are complete there (first call only: later calls assume the
    with torch.cuda.stream(owner._dd_stream):    # tokens were written before the previous call returned)
        dd = enc.dedup(tokens, reuse=reuse)
    cur.wait_stream(owner._dd_stream)
    for t in dd:
        if t.is_cuda:
            t.record_stream(cur)    # consumed on the caller's stream (copied into the captured tensors, behind the previous
            #                         replay): the allocator must not hand its memory to the NEXT call's dedup before that
    return dd


class GraphedAct:
    """One rollout step — `BasePolicy.act` (policy.py:34-56 of the reference: network forward from raw RGB-D, progress head,
    action distribution, critic) — as a HIP graph.  At B = 1 the step is ≈300 launches of a few microseconds: 3.75 ms eagerly
    (bf16 mode) of which 1 ms is GPU work.  Inputs are copied into the captured tensors every call (1 MB per environment); the
    map state (`net.rgb_mapping_module.full_global_map`, which trainers slice and re-assign between steps) is adopted: a
    re-assigned map of the same shape is copied into the captured one, another shape gets its own graph.  Outputs are owned by
    the graph (valid until the next call)."""

    def __init__(self, policy, eager_calls=2):
        self.policy = policy
        self.eager_calls = max(1, int(eager_calls))
        self.calls = 0
        self._graphs = {}
        self._dd_stream = None
        self._stream = None

    def __call__(self, observations, rnn_hidden_states, prev_actions, masks, deterministic=False):
        self.calls += 1
        pol = self.policy
        dd = _eager_dedup(self, pol.net.instruction_encoder, observations["instruction"], self.calls == 1, reuse=True)
        if self.calls <= self.eager_calls:
            obs = dict(observations)
            obs["instruction_dedup"] = dd
            if self._stream is None:
                self._stream = torch.cuda.Stream()
            cur = torch.cuda.current_stream()
            self._stream.wait_stream(cur)
            with torch.cuda.stream(self._stream), torch.no_grad():
                out = pol.act(obs, rnn_hidden_states, prev_actions, masks, deterministic=deterministic)
            cur.wait_stream(self._stream)
            return out
        mm = pol.net.rgb_mapping_module

        def sig(t):
            return (tuple(t.shape), t.dtype)
        key = (tuple(sorted((k, sig(v)) for k, v in observations.items() if torch.is_tensor(v))), sig(rnn_hidden_states), sig(prev_actions),
               sig(masks), bool(deterministic), int(dd[0].shape[0]), int(dd[2].max()), sig(mm.full_global_map))
        g = self._graphs.get(key)
        if g is None:
            g = self._capture(observations, rnn_hidden_states, prev_actions, masks, deterministic, dd)
            self._graphs[key] = g
        pairs = [(g["h_in"], rnn_hidden_states), (g["prev"], prev_actions), (g["masks"], masks)] + [(t, observations[k]) for k, t in g["obs"].items()]
        if g.get("dd_src") is not dd:          # (the kept dedup of unchanged instructions is in the captured tensors already)
            pairs += [(g["dd"][i], dd[i]) for i in (0, 1, 3)]
            g["dd_src"] = dd
        ops.copy_multi([d for d, _ in pairs], [s_ for _, s_ in pairs])      # one launch (12 separate copies are 0.1 ms of launches)
        if mm.full_global_map.data_ptr() != g["map"].data_ptr():     # re-assigned by the trainer (episode bookkeeping)
            g["map"].copy_(mm.full_global_map)
            mm.full_global_map = g["map"]
        refresh = getattr(pol.net, "refresh_folded", None)
        if refresh is not None:      # the rollout route's folded convolution operands follow the parameters (in place)
            refresh()
        g["graph"].replay()
        # the reference's act() leaves the new hidden state in the caller's tensor (mg_map_policy.py:220-227,242-249 write
        # rnn_hidden_states in place): so does the replay.  The returned tensors are owned by the graph — valid until the next
        # call ("output lifetime").
        rnn_hidden_states.copy_(g["out"][3])
        return g["out"]

    def _capture(self, observations, rnn_hidden_states, prev_actions, masks, deterministic, dd):
        pol = self.policy
        mm = pol.net.rgb_mapping_module
        obs_s = {k: v.clone() for k, v in observations.items() if torch.is_tensor(v)}
        dd_s = tuple(t.clone() for t in dd)
        h_in, prev_s, masks_s = rnn_hidden_states.clone(), prev_actions.clone(), masks.clone()
        map_s = mm.full_global_map
        obs_c = dict(obs_s)
        obs_c["instruction_dedup"] = dd_s
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(graph):
            value, action, logp, h = pol.act(obs_c, h_in.clone(), prev_s, masks_s, deterministic=deterministic)
        if mm.full_global_map.data_ptr() != map_s.data_ptr():
            raise _abi.WsmgError("GraphedAct: the map state was re-allocated inside act(); the graph cannot adopt it")
        ops.check_rnn_status()
        return dict(graph=graph, obs=obs_s, dd=dd_s, h_in=h_in, prev=prev_s, masks=masks_s, map=map_s, out=(value, action, logp, h))


class GraphedUpdate:
    def __init__(self, policy, optimizer, loss_fn, eager_calls=3):
        """loss_fn(pred, aux_loss, observations, weights) -> scalar loss tensor.  The first `eager_calls` updates run eagerly
        (first-call initialisations of the library and of the allocator must not happen under capture)."""
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            raise _abi.WsmgError("GraphedUpdate: data-parallel updates are not captured (the gradient exchange stays eager)")
        self.policy, self.optimizer, self.loss_fn = policy, optimizer, loss_fn
        self.eager_calls = max(1, int(eager_calls))   # at least one: the optimizer's device-side step counters are created eagerly
        self.calls = 0
        self._graphs = {}      # signature -> captured graph, least recently used first; at most `max_graphs` are kept
        self.max_graphs = 4
        self._static = {}      # id(tensor) -> tensor: inputs the caller registered as static buffers (adopted, never cloned)
        self._written = None   # parameters + buffers a replay rewrites behind autograd's back (version bumps after a replay)
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
        mine = {id(p) for p in self.policy.parameters()}
        seen = set()

        def reaches_policy(fn):      # does this autograd graph end in one of THIS policy's parameters?
            stack = [fn]
            while stack:
                f = stack.pop()
                if f is None or id(f) in seen:
                    continue
                seen.add(id(f))
                v = getattr(f, "variable", None)      # AccumulateGrad
                if v is not None and id(v) in mine:
                    return True
                stack.extend(nf for nf, _ in f.next_functions)
            return False
        left = [o for o in gc.get_objects() if torch.is_tensor(o) and o.is_cuda and o.grad_fn is not None]
        left = [t for t in left if reaches_policy(t.grad_fn)]
        if left:
            what = ", ".join(f"{tuple(t.shape)} <- {type(t.grad_fn).__name__}" for t in left[:6])
            raise _abi.WsmgError(f"GraphedUpdate: {len(left)} tensor(s) of an earlier autograd graph are still referenced ({what}); "
                                 "drop them (del loss, predictions, ...) before the first graphed update")

    def register_static_inputs(self, *tensors):
        """Declare input tensors (observation tensors, masks, ...) as STATIC buffers: the captured graph reads them in place, so a
        caller that refills the same buffers for every batch pays no copy (the cached ego map alone is 1.3 GB per update).  Any
        other input tensor is cloned at capture and copied into that clone on every call — the caller's tensors are never
        written to."""
        for t in tensors:
            if isinstance(t, dict):
                self.register_static_inputs(*[v for v in t.values() if torch.is_tensor(v)])
            elif torch.is_tensor(t):
                self._static[id(t)] = t
        return self

    def _adopt(self, t):
        return t if id(t) in self._static else t.clone()

    def _bump_versions(self):
        """A replay rewrote parameters (the optimizer step), BatchNorm running statistics and counters without any autograd
        version counter noticing; caches of derived operands (FoldCache, packed LSTM weights) compare those counters."""
        if self._written is None:
            self._written = [p for p in self.policy.parameters()] + [b for b in self.policy.buffers()]
        torch.autograd.graph.increment_version(self._written)

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
        # eager, on its own stream: the dedup's host read-back must not wait for the previous replay (it depends on the
        # instruction tokens only), or the GPU idles while the host prepares the next one
        dd = _eager_dedup(self, self.policy.net.instruction_encoder, observations["instruction"], self.calls == 1)
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
        g = self._graphs.pop(key, None)
        if g is None:
            while len(self._graphs) >= self.max_graphs:      # every graph owns a private memory pool: evict the least recently used
                old_key = next(iter(self._graphs))
                self._graphs.pop(old_key)["graph"].reset()
            g = self._capture(observations, rnn_hidden_states, prev_actions, masks, weights, dd)
        self._graphs[key] = g                                # (re-inserted last = most recently used)
        # inputs: same storage (a registered static buffer) -> nothing to do; other storage -> copy into the graph's own tensors
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
        self._bump_versions()
        rnn_hidden_states.copy_(g["h_out"])
        return g["loss"]

    def _capture(self, observations, rnn_hidden_states, prev_actions, masks, weights, dd):
        # the graph's static inputs: registered static buffers are adopted, everything else is CLONED — a later call with tensors
        # at other addresses copies into the clones, never into a batch the caller still holds
        obs_s = {k: self._adopt(v) for k, v in observations.items() if torch.is_tensor(v)}
        prev_actions, masks, weights = self._adopt(prev_actions), self._adopt(masks), self._adopt(weights)
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
