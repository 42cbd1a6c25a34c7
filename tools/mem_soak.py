#!/usr/bin/env python3
"""GPU-box tool: device and host memory over a long run of updates (leak check): python tools/mem_soak.py [updates]   (WSMG_HOSTPROF_DP=1: one-rank RCCL group).
Set-up shared with host_profile.py — where the HOST's time goes in one update (both threads: the caller's and autograd's), by torch.profiler's CPU
activity — self CPU time per operator / Function over a few updates of the bench workload.   python tools/host_profile.py [updates]
WSMG_HOSTPROF_DP=1: with a one-rank RCCL process group and the gradient exchange bench.py --gpus N builds (what the exchange adds)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ws-mgmap_amd"))
import torch
import bench
from wsmgmap import ops
from wsmgmap.common.aux_losses import AuxLosses
from wsmgmap.config import default_model_config
from wsmgmap.models.policy import BasePolicy
from wsmgmap.optim import Adam

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda")
T, N = 64, 8


class Box:
    shape = (2,)


pol = BasePolicy(None, Box(), default_model_config(num_proc=N, compute_dtype="bf16")).to(dev)
pol.train(); pol.net.depth_encoder.eval(); pol.net.rgb_encoder.eval()
opt = Adam(pol.parameters(), lr=2.5e-4)
reducer = None
if os.environ.get("WSMG_HOSTPROF_DP") == "1":
    import torch.distributed as dist
    from wsmgmap.parallel import GradAllReducer
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev if dev.index is not None else torch.device("cuda:0"))
    reducer = GradAllReducer(pol.parameters(), bucket_bytes=8 << 20, single_rank_exchange=True, exchange_stream="instruction")
    reducer.broadcast_parameters(pol)
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
    if reducer:
        reducer.finish()
    opt.step()


import gc, resource
for _ in range(10):
    update()
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
print("update  allocated MB  reserved MB  host RSS MB  gc objects")
for i in range(n + 1):
    if i % max(1, n // 8) == 0:
        torch.cuda.synchronize()
        print("%6d  %12.1f  %11.1f  %11.1f  %10d" % (i, torch.cuda.memory_allocated() / 1e6, torch.cuda.memory_reserved() / 1e6,
                                                      resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e3, len(gc.get_objects())), flush=True)
    update()
