#!/usr/bin/env python3
"""GPU-box diagnostic: GPU execution time of one update with the launch queue pre-filled (no host starvation).
A ~60 ms dummy workload is queued first, the whole update is enqueued behind it while it runs, and HIP events around the
update give the time the GPU needs when it never waits for the host.  Needs WSMG_DIAG_DEDUP_MEMO=1 (no read-back inside
the update).  The difference to bench.py's ms_per_step is what launch overhead / host-boundness costs."""
import os, sys, time
os.environ.setdefault("WSMG_DIAG_DEDUP_MEMO", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ws-mgmap_amd"))
import torch
import bench
from wsmgmap.common.aux_losses import AuxLosses
from wsmgmap.config import default_model_config
from wsmgmap.models.policy import BasePolicy

T, N = 64, 8
dev = torch.device("cuda:0")
torch.manual_seed(0)
policy = BasePolicy(None, bench._Box(), default_model_config(num_proc=1, gpu_id=0, compute_dtype=sys.argv[1] if len(sys.argv) > 1 else "bf16"))
policy.net.instruction_encoder.embedding_layer.weight.requires_grad_(False)
policy = policy.to(dev); policy.train(); policy.net.depth_encoder.eval(); policy.net.rgb_encoder.eval()
from wsmgmap.optim import Adam as WsmgAdam
opt = WsmgAdam(policy.parameters(), lr=2.5e-4)
obs, prev, masks, weights = bench.synth_batch(T, N, dev, 1000)
AuxLosses.activate()

def update():
    opt.zero_grad(set_to_none=True)
    AuxLosses.clear()
    h0 = torch.zeros(policy.net.num_recurrent_layers, N, 512, device=dev)
    o = dict(obs)
    pred, aux = policy(o, h0, prev, masks, weights)
    loss = bench.dagger_loss(pred, aux, o["waypoint"], weights)
    loss.backward()
    opt.step()

for _ in range(3):
    update()
torch.cuda.synchronize()
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
res = []
for rep in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(40):
        a @ a           # ~1.5 ms each: the GPU is busy for ~60 ms
    e0.record()
    h = time.perf_counter()
    update()
    host = time.perf_counter() - h
    e1.record()
    torch.cuda.synchronize()
    res.append((e0.elapsed_time(e1), host * 1e3))
print("GPU time of one pre-queued update (ms), host enqueue (ms):", [(round(g, 3), round(h, 2)) for g, h in res])
# steady-state wall for comparison
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    update()
torch.cuda.synchronize()
print("steady-state wall %.3f ms/update" % ((time.perf_counter() - t0) * 100))
