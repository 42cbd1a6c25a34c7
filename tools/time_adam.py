#!/usr/bin/env python3
"""GPU-box tool: GPU time of one optimizer step over the policy's live parameters: torch.optim.Adam multi-tensor
(default) vs fused=True."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ws-mgmap_amd"))
import torch
import bench
from wsmgmap.common.aux_losses import AuxLosses
from wsmgmap.config import default_model_config
from wsmgmap.models.policy import BasePolicy
T, N = 8, 8
dev = torch.device("cuda:0")
torch.manual_seed(0)
policy = BasePolicy(None, bench._Box(), default_model_config(num_proc=1, gpu_id=0, compute_dtype="bf16"))
policy.net.instruction_encoder.embedding_layer.weight.requires_grad_(False)
policy = policy.to(dev); policy.train(); policy.net.depth_encoder.eval(); policy.net.rgb_encoder.eval()
obs, prev, masks, weights = bench.synth_batch(T, N, dev, 1000)
AuxLosses.activate(); AuxLosses.clear()
o = dict(obs)
pred, aux = policy(o, torch.zeros(policy.net.num_recurrent_layers, N, 512, device=dev), prev, masks, weights)
bench.dagger_loss(pred, aux, o["waypoint"], weights).backward()
live = [p for p in policy.parameters() if p.grad is not None]
print("live parameter tensors:", len(live), "floats:", sum(p.numel() for p in live))
for name, kw in (("foreach (default)", {}), ("fused", {"fused": True})):
    opt = torch.optim.Adam(policy.parameters(), lr=1e-6, **kw)
    for _ in range(3): opt.step()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): opt.step()
    b.record(); torch.cuda.synchronize()
    print(f"Adam {name}: {a.elapsed_time(b) / 20 * 1e3:.0f} us per step")
