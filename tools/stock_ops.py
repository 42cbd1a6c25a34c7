#!/usr/bin/env python3
"""GPU-box tool: which STOCK torch operators one update still launches, and from where.  torch.profiler with Python stacks over a few
updates of the bench workload; per (aten op, innermost frame inside this repository): calls per update and whether a device kernel or a
runtime copy / fill follows.   python tools/stock_ops.py [updates]"""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ws-mgmap_amd"))
import torch
import bench
from wsmgmap import ops
from wsmgmap.common.aux_losses import AuxLosses
from wsmgmap.config import default_model_config
from wsmgmap.models.policy import BasePolicy
from wsmgmap.optim import Adam

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda")
T, N = 64, 8


class Box:
    shape = (2,)


pol = BasePolicy(None, Box(), default_model_config(num_proc=N, compute_dtype="bf16")).to(dev)
pol.train(); pol.net.depth_encoder.eval(); pol.net.rgb_encoder.eval()
opt = Adam(pol.parameters(), lr=2.5e-4)
obs, prev, masks, weights = bench.synth_batch(T, N, dev, 1000)
ops.mark_inputs_ready(obs["instruction"])
AuxLosses.activate()


def update():
    opt.zero_grad(set_to_none=True)
    AuxLosses.clear()
    h0 = torch.zeros(pol.net.num_recurrent_layers, N, 512, device=dev)
    o = dict(obs)
    pred, aux = pol(o, h0, prev, masks, weights)
    loss = bench.dagger_loss(pred, aux, o["waypoint"], weights)
    loss.backward()
    opt.step()


for _ in range(6):
    update()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(n):
        update()
    torch.cuda.synchronize()
LAUNCHERS = ("hipLaunchKernel", "hipExtModuleLaunchKernel", "hipMemcpyAsync", "hipMemsetAsync", "hipExtLaunchKernel", "hipModuleLaunchKernel",
             "hipMemcpyWithStream", "hipMemcpy")
events = prof.events()
# aten ops (leaves) that contain a runtime launch call
by = collections.Counter()
launch_total = collections.Counter()
for e in events:
    if e.name in LAUNCHERS:
        launch_total[e.name] += 1
        p = e.cpu_parent
        op = None
        while p is not None:
            if p.name.startswith("aten::") or p.name.startswith("wsmg") or "Backward" in p.name:
                op = p
                break
            p = p.cpu_parent
        name = op.name if op is not None else "(no aten parent: ctypes call)"
        where = ""
        q = op
        while q is not None and not where:
            for fr in (q.stack or []):
                if "/repo/" in fr and "torch/" not in fr:
                    where = fr.split("/repo/")[-1]
                    break
            q = q.cpu_parent
        by[(name, e.name, where)] += 1
print("runtime calls per update:", {k: round(v / n, 1) for k, v in launch_total.items()})
print("%-34s %-22s %7s  %s" % ("aten op", "runtime call", "/update", "innermost repository frame"))
for (name, call, where), c in sorted(by.items(), key=lambda kv: -kv[1]):
    if name.startswith("(no aten"):
        continue
    print("%-34s %-22s %7.1f  %s" % (name[:34], call, c / n, where[:110]))
print("(launches made through the C ABI, no aten parent: %.1f per update)" % (sum(c for (nm, _, _), c in by.items() if nm.startswith("(no aten")) / n))
