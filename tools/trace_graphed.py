#!/usr/bin/env python3
"""GPU-box diagnostic: the update captured as ONE HIP graph and replayed — run under `rocprofv3 --kernel-trace` this gives the
GPU-side timeline of an update without the tracer's host overhead between launches (the eager trace shows 1.2 ms of host-paced
gaps per update that do not exist without the profiler).  python tools/trace_graphed.py [replays]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ws-mgmap_amd"))
import torch
import bench
from wsmgmap import ops
from wsmgmap.common.aux_losses import AuxLosses
from wsmgmap.config import default_model_config
from wsmgmap.models.policy import BasePolicy
from wsmgmap.graph import GraphedUpdate
from wsmgmap.optim import Adam as WsmgAdam

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
T, N = 64, 8
dev = torch.device("cuda:0")
torch.manual_seed(0)
policy = BasePolicy(None, bench._Box(), default_model_config(num_proc=1, gpu_id=0, compute_dtype="bf16"))
policy.net.instruction_encoder.embedding_layer.weight.requires_grad_(False)
policy = policy.to(dev); policy.train(); policy.net.depth_encoder.eval(); policy.net.rgb_encoder.eval()
opt = WsmgAdam(policy.parameters(), lr=2.5e-4, capturable=True)
obs, prev, masks, weights = bench.synth_batch(T, N, dev, 1000)
AuxLosses.activate()
gu = GraphedUpdate(policy, opt, lambda pred, aux, o, w: bench.dagger_loss(pred, aux, o["waypoint"], w), eager_calls=2)
gu.register_static_inputs(obs, prev, masks, weights)
hs = torch.zeros(policy.net.num_recurrent_layers, N, 512, device=dev)
import time
for i in range(4 + reps):
    if i == 4:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    hs.zero_()
    loss = gu(obs, hs, prev, masks, weights)
torch.cuda.synchronize()
print("graph replay: %.3f ms per update, loss %.5f" % ((time.perf_counter() - t0) / reps * 1e3, float(loss)))
ops.check_rnn_status()
