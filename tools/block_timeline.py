#!/usr/bin/env python3
"""GPU-box diagnostic: when the chunks of the pipelined recurrent core (wsmgmap/recurrent.py) finish, WITHOUT a profiler: HIP events
recorded behind every chunk's GRU 1 / attention stage / GRU 2 on their own streams (ops.mark), times relative to the block's entry,
forward and backward.  python tools/block_timeline.py [updates]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ws-mgmap_amd"))
import torch
import bench
from wsmgmap import ops
from wsmgmap.common.aux_losses import AuxLosses
from wsmgmap.config import default_model_config
from wsmgmap.models.policy import BasePolicy
from wsmgmap.optim import Adam as WsmgAdam

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
T, N = 64, 8
dev = torch.device("cuda:0")
torch.manual_seed(0)
policy = BasePolicy(None, bench._Box(), default_model_config(num_proc=1, gpu_id=0, compute_dtype="bf16"))
policy.net.instruction_encoder.embedding_layer.weight.requires_grad_(False)
policy = policy.to(dev); policy.train(); policy.net.depth_encoder.eval(); policy.net.rgb_encoder.eval()
opt = WsmgAdam(policy.parameters(), lr=2.5e-4)
obs, prev, masks, weights = bench.synth_batch(T, N, dev, 1000)
AuxLosses.activate()

def update(marks):
    if marks: ops.marks_begin()
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
    if marks: return ops.marks_end()

for _ in range(4): update(False)
torch.cuda.synchronize()
acc = {}
for r in range(reps):
    # no synchronisation between the updates: the host is as far ahead of the GPU as it is in the bench
    m = update(True)
    acc[r] = m
torch.cuda.synchronize()
import collections
tot = collections.OrderedDict()
for r, m in acc.items():
    d = dict(m)
    base = d.get("f:state_in")
    for n, e in m:
        tot.setdefault(n, []).append(base.elapsed_time(e) * 1e3)
print("event                 us after f:state_in (mean / max of %d updates)" % reps)
for n, v in tot.items():
    print("%-20s %9.1f %9.1f" % (n, sum(v) / len(v), max(v)))
# intervals per update: where do slow updates lose their time?
names = list(tot)
def col(n): return tot[n]
pairs = [("f:entry", "f:map_stack"), ("f:map_stack", "f:state_in"), ("f:state_in", "f:gru2"), ("f:gru2", "b:gru2"), ("b:gru2", "b:state_in"),
         ("b:state_in", "b:map_stack"), ("b:map_stack", "f:backward_done")]
print("\ninterval                         med      p90      max   (us, over %d updates)" % reps)
for a, b in pairs:
    if a in tot and b in tot:
        d = sorted(y - x for x, y in zip(col(a), col(b)))
        print("%-14s -> %-14s %8.1f %8.1f %8.1f" % (a, b, d[len(d) // 2], d[int(len(d) * .9)], d[-1]))
