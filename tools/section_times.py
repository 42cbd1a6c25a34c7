#!/usr/bin/env python3
"""GPU-box diagnostic: where one update's time goes, section by section, WITHOUT a profiler attached (HIP events at the
stage boundaries of wsmgmap.ops.mark, forward and backward).  python tools/section_times.py [bf16|f32] [updates]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ws-mgmap_amd"))
import torch
import bench
from wsmgmap import ops
from wsmgmap.common.aux_losses import AuxLosses
from wsmgmap.config import default_model_config
from wsmgmap.models.policy import BasePolicy

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
T, N = 64, 8
dev = torch.device("cuda:0")
torch.manual_seed(0)
policy = BasePolicy(None, bench._Box(), default_model_config(num_proc=1, gpu_id=0, compute_dtype=mode))
policy.net.instruction_encoder.embedding_layer.weight.requires_grad_(False)
policy = policy.to(dev); policy.train(); policy.net.depth_encoder.eval(); policy.net.rgb_encoder.eval()
from wsmgmap.optim import Adam as WsmgAdam
opt = WsmgAdam(policy.parameters(), lr=2.5e-4)
obs, prev, masks, weights = bench.synth_batch(T, N, dev, 1000)
AuxLosses.activate()

def update(marks):
    if marks: ops.marks_begin(); ops.mark("start")
    opt.zero_grad(set_to_none=True)
    AuxLosses.clear()
    h0 = torch.zeros(policy.net.num_recurrent_layers, N, 512, device=dev)
    o = dict(obs)
    pred, aux = policy(o, h0, prev, masks, weights)
    loss = bench.dagger_loss(pred, aux, o["waypoint"], weights)
    if marks: ops.mark("loss")
    loss.backward()
    if marks: ops.mark("backward_done")
    opt.step()
    if marks:
        ops.mark("opt_step")
        return ops.marks_end()

for _ in range(3): update(False)
torch.cuda.synchronize()
acc, order = {}, []
import time
t0 = time.perf_counter()
# WSMG_SECTIONS_NOSYNC=1: no synchronisation between the updates — the host runs ahead as it does in bench.py, and the sections are
# those of the steady state (with the per-update sync the host starts every update level with the GPU and the first sections are its)
nosync = os.environ.get("WSMG_SECTIONS_NOSYNC", "0") == "1"
runs = []
for r in range(reps):
    m = update(True)
    if not nosync:
        torch.cuda.synchronize()
    runs.append(m)
torch.cuda.synchronize()
if nosync:   # update to update: the previous update's last mark to this one's first
    for r in range(1, len(runs)):
        runs[r] = [("f:prev_opt_step", runs[r - 1][-1][1])] + runs[r]
    runs = runs[1:]
for m in runs:
    names = [n for n, _ in m]
    if not order: order = names
    for (n0, e0), (n1, e1) in zip(m[:-1], m[1:]):
        acc.setdefault((n0, n1), []).append(e0.elapsed_time(e1))
wall = (time.perf_counter() - t0) / reps * 1e3   # (the loop above only sorts events: the wall time is dominated by the updates)
tot = 0.0
for (n0, n1), v in acc.items():
    ms = sum(v) / len(v); tot += ms
    print("%-16s -> %-16s %7.3f ms" % (n0, n1, ms))
print("sum %.3f ms   wall per update (with per-update sync) %.3f ms" % (tot, wall))
