#!/usr/bin/env python3
"""GPU-box diagnostic: per update, how long the HOST needed to enqueue it and how long the GPU needed to run it (events, no
synchronisation inside the run), and how far ahead of the GPU the host was when it finished enqueuing.  python tools/host_vs_gpu.py [updates]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ws-mgmap_amd"))
import torch
import bench
from wsmgmap.common.aux_losses import AuxLosses
from wsmgmap.config import default_model_config
from wsmgmap.models.policy import BasePolicy
from wsmgmap.optim import Adam as WsmgAdam
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
T, N = 64, 8
dev = torch.device("cuda:0")
torch.manual_seed(0)
policy = BasePolicy(None, bench._Box(), default_model_config(num_proc=1, gpu_id=0, compute_dtype="bf16"))
policy.net.instruction_encoder.embedding_layer.weight.requires_grad_(False)
policy = policy.to(dev); policy.train(); policy.net.depth_encoder.eval(); policy.net.rgb_encoder.eval()
opt = WsmgAdam(policy.parameters(), lr=2.5e-4)
obs, prev, masks, weights = bench.synth_batch(T, N, dev, 1000)
AuxLosses.activate()
phases = []
def update():
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    AuxLosses.clear()
    h0 = torch.zeros(policy.net.num_recurrent_layers, N, 512, device=dev)
    o = dict(obs)
    pred, aux = policy(o, h0, prev, masks, weights)
    t1 = time.perf_counter()
    loss = bench.dagger_loss(pred, aux, o["waypoint"], weights)
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    opt.step()
    phases.append((t1 - t0, t2 - t1, t3 - t2, time.perf_counter() - t3))
for _ in range(30): update()
torch.cuda.synchronize()
ev, host = [], []
t_ref = time.perf_counter()
e0 = torch.cuda.Event(enable_timing=True); e0.record()
for i in range(reps):
    h0_ = time.perf_counter()
    update()
    e = torch.cuda.Event(enable_timing=True); e.record()
    ev.append(e); host.append((h0_ - t_ref, time.perf_counter() - t_ref))
torch.cuda.synchronize()
def q(v, p): v = sorted(v); return v[min(len(v) - 1, int(p * len(v)))]
gpu_end = [e0.elapsed_time(e) * 1e-3 for e in ev]       # seconds after e0 (~ t_ref)
gd = [gpu_end[0]] + [gpu_end[i] - gpu_end[i - 1] for i in range(1, reps)]
hd = [b - a for a, b in host]
lead = [gpu_end[i] - host[i][1] for i in range(reps)]    # > 0: the host finished enqueuing update i before the GPU finished it
ph = phases[-reps:]
for j, nm in enumerate(("forward (incl. the dedup read-back wait)", "loss", "backward", "optimizer")):
    v = [p_[j] * 1e3 for p_ in ph]
    print("host %-42s med %.2f p90 %.2f max %.2f ms" % (nm, q(v, .5), q(v, .9), max(v)))
print("GPU  ms/update: med %.2f p90 %.2f max %.2f" % (q(gd, .5) * 1e3, q(gd, .9) * 1e3, max(gd) * 1e3))
print("host ms/update: med %.2f p90 %.2f max %.2f" % (q(hd, .5) * 1e3, q(hd, .9) * 1e3, max(hd) * 1e3))
print("host lead at end of enqueue (ms): med %.2f min %.2f" % (q(lead, .5) * 1e3, min(lead) * 1e3))
slow = [i for i in range(reps) if gd[i] > q(gd, .5) * 1.05]
print("slow updates (GPU > 1.05 x median): %d;  their host times: %s" % (len(slow), ["%.1f" % (hd[i] * 1e3) for i in slow[:12]]))
print("                                          their GPU times:  %s" % ["%.1f" % (gd[i] * 1e3) for i in slow[:12]])
