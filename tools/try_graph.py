#!/usr/bin/env python3
"""GPU-box experiment: one whole update (zero_grad + forward + loss + backward + Adam) captured into a HIP graph and replayed,
against the same update launched eagerly.  python tools/try_graph.py [updates]"""
import os, sys, time
os.environ.setdefault("WSMG_DIAG_DEDUP_MEMO", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ws-mgmap_amd"))
import torch
import bench
from wsmgmap import ops
from wsmgmap.common.aux_losses import AuxLosses
from wsmgmap.config import default_model_config
from wsmgmap.models.policy import BasePolicy

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
T, N = 64, 8
dev = torch.device("cuda:0")
torch.manual_seed(0)
policy = BasePolicy(None, bench._Box(), default_model_config(num_proc=1, gpu_id=0, compute_dtype="bf16"))
policy.net.instruction_encoder.embedding_layer.weight.requires_grad_(False)
policy = policy.to(dev); policy.train(); policy.net.depth_encoder.eval(); policy.net.rgb_encoder.eval()
obs, prev, masks, weights = bench.synth_batch(T, N, dev, 1000)
AuxLosses.activate()
loss_out = torch.zeros((), device=dev)


def make_update(opt):
    def update():
        opt.zero_grad(set_to_none=True)
        AuxLosses.clear()
        h0 = torch.zeros(policy.net.num_recurrent_layers, N, 512, device=dev)
        o = dict(obs)
        pred, aux = policy(o, h0, prev, masks, weights)
        loss = bench.dagger_loss(pred, aux, o["waypoint"], weights)
        loss.backward()
        opt.step()
        loss_out.copy_(loss.detach())
    return update


def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


plain = make_update(torch.optim.Adam(policy.parameters(), lr=2.5e-4))
for _ in range(4): plain()
print(f"eager, stock Adam: {timed(plain, reps):.3f} ms/update")
update = make_update(torch.optim.Adam(policy.parameters(), lr=2.5e-4, capturable=True))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(4): update()
torch.cuda.current_stream().wait_stream(s)
print(f"eager, capturable Adam: {timed(update, reps):.3f} ms/update  loss {float(loss_out):.5f}")
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    update()
g.replay(); torch.cuda.synchronize()
print(f"graph: {timed(g.replay, reps):.3f} ms/update  loss {float(loss_out):.5f}")
print(f"eager again: {timed(update, reps):.3f} ms/update  loss {float(loss_out):.5f}")
ops.check_rnn_status()
