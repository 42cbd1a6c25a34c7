"""Adam for the update path, as ONE multi-tensor HIP launch per 48 tensors (csrc/wsmg_optim.hip).

The reference builds `torch.optim.Adam(self.actor_critic.parameters(), lr=...)` (common_trainer.py:67-69) and steps it after
`loss.backward()` (dagger_trainer.py:540-541).  This class is that optimizer — same constructor arguments, same arithmetic
in the same order, `state_dict()` interchangeable with torch.optim.Adam's (`step`, `exp_avg`, `exp_avg_sq`) — with the step
issued as 3 kernel launches for the policy's 102 live tensors instead of the stock multi-tensor path's 15 (0.25 ms of GPU time
per update -> 0.06 ms).  float32 CUDA parameters only: anything else raises (there is no fallback path)."""
import ctypes

import torch

from . import _abi


class _AdamDesc(ctypes.Structure):
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p), ("exp_avg_sq", ctypes.c_void_p),
                ("n", ctypes.c_longlong)]


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, maximize=False,
                 capturable=False):
        """capturable=True: the step count also lives in a device scalar (one per parameter group, incremented on the device)
        and the kernel computes the bias corrections from it — the form `wsmgmap.graph.GraphedUpdate` captures into a HIP graph
        (kernel arguments are frozen at capture).  All stepped parameters of a group must then share one step count."""
        self._capturable = bool(capturable)
        self._step_dev = {}
        if amsgrad or maximize:
            raise ValueError("wsmgmap.optim.Adam implements amsgrad=False, maximize=False (what the reference trains with)")
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0 or not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError(f"invalid Adam hyper-parameters: lr={lr} betas={betas} eps={eps} weight_decay={weight_decay}")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            by_step = {}
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise _abi.WsmgError("wsmgmap.optim.Adam: parameters must be contiguous float32 CUDA tensors")
                if g.is_sparse or g.dtype != torch.float32 or g.device != p.device:
                    raise _abi.WsmgError("wsmgmap.optim.Adam: gradients must be dense float32 tensors on the parameter's device")
                if not g.is_contiguous():
                    g = g.contiguous()
                    g.record_stream(torch.cuda.current_stream(p.device))   # the copy must outlive the launch queued below
                st = self.state[p]
                if not st:
                    st["step"] = 0    # a Python int while training (102 CPU-tensor increments per step cost 1 ms of host time);
                    #                   state_dict() / load_state_dict() convert from / to torch.optim.Adam's float32 tensor
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                by_step.setdefault(st["step"], []).append((p, g, st["exp_avg"], st["exp_avg_sq"]))
            b1, b2 = group["betas"]
            if self._capturable and len(by_step) > 1:
                raise _abi.WsmgError("wsmgmap.optim.Adam(capturable=True): the stepped parameters of a group must share one step count")
            for step, items in by_step.items():
                descs = (_AdamDesc * len(items))()
                for d, (p, g, m, v) in zip(descs, items):
                    d.param, d.grad, d.exp_avg, d.exp_avg_sq, d.n = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()
                dev = items[0][0].device
                stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                with torch.cuda.device(dev):
                    if self._capturable:
                        key = id(group)
                        if key not in self._step_dev:
                            self._step_dev[key] = torch.full((), float(step - 1), device=dev, dtype=torch.float32)
                        sd = self._step_dev[key]
                        sd.add_(1.0)      # on the device: a replayed graph advances it without the host
                        _abi.call("wsmg_adam_step_multi_dev", descs, len(items), float(group["lr"]), float(b1), float(b2),
                                  float(group["eps"]), float(group["weight_decay"]), ctypes.c_void_p(sd.data_ptr()), stream)
                    else:
                        _abi.call("wsmg_adam_step_multi", descs, len(items), float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                  float(group["weight_decay"]), 1.0 - b1 ** step, 1.0 - b2 ** step, stream)
                # the kernel wrote the parameters through raw pointers: advance their autograd version counters, which is what
                # caches of derived operands (FoldCache, InstructionEncoder.packed_lstm_weights) compare
                torch.autograd.graph.increment_version([it[0] for it in items])
        return loss

    def note_replayed_steps(self, n=1):
        """A captured graph that contains this optimizer's step was replayed n times: advance the host-side step counts (the
        device counters advanced inside the graph)."""
        stepped = []
        for p, st in self.state.items():
            if "step" in st:
                st["step"] += n
                stepped.append(p)
        if n > 0 and stepped:     # the replay rewrote them without any version counter noticing
            torch.autograd.graph.increment_version(stepped)

    def state_dict(self):
        sd = super().state_dict()
        sd["state"] = {k: {**v, "step": torch.tensor(float(v["step"]), dtype=torch.float32)} if "step" in v else v
                       for k, v in sd["state"].items()}
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for st in self.state.values():
            if "step" in st:
                st["step"] = int(round(float(st["step"])))
        self._step_dev = {}       # re-created from the loaded step counts at the next capturable step
