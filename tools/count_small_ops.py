#!/usr/bin/env python3
"""GPU-box diagnostic: which Python lines issue the small copy / fill / add launches of one update (torch.profiler, CPU side,
with stacks).  python tools/count_small_ops.py"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ws-mgmap_amd"))
import torch
import bench
from wsmgmap.common.aux_losses import AuxLosses
from wsmgmap.config import default_model_config
from wsmgmap.models.policy import BasePolicy
from wsmgmap.optim import Adam

T, N = int(os.environ.get("T", 64)), int(os.environ.get("N", 8))
dev = torch.device("cuda:0")
torch.manual_seed(0)
policy = BasePolicy(None, bench._Box(), default_model_config(num_proc=1, gpu_id=0, compute_dtype="bf16"))
policy.net.instruction_encoder.embedding_layer.weight.requires_grad_(False)
policy = policy.to(dev); policy.train(); policy.net.depth_encoder.eval(); policy.net.rgb_encoder.eval()
opt = Adam(policy.parameters(), lr=2.5e-4)
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


for _ in range(3): update()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    update()
    torch.cuda.synchronize()
want = ("aten::zeros_like", "aten::new_zeros", "aten::full", "aten::full_like", "aten::new_full", "aten::ones_like", "aten::gather", "aten::scatter", "aten::scatter_", "aten::scatter_add_", "aten::take_along_dim", "aten::index_add_", "aten::gather_backward", "aten::value_selecting_reduction_backward", "aten::max", "aten::min", "aten::sort", "aten::topk", "aten::copy_", "aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::cat", "aten::sum", "aten::mul", "aten::clone", "aten::contiguous",
        "aten::zeros", "aten::index_select", "aten::_to_copy")
cnt = collections.Counter()
for e in prof.events():
    if e.name not in want:
        continue
    frame = ""
    for s in (e.stack or []):
        if "wsmgmap" in s or "bench.py" in s or "count_small_ops" in s:
            frame = s.split("ws-mgmap_amd/")[-1][:90]
            break
    if not frame and e.stack:
        frame = "| " + e.stack[0][-80:]
    shapes = str(e.input_shapes)[:60]
    cnt[(e.name, frame, shapes)] += 1
only = os.environ.get("ONLY")
for (name, frame, shapes), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:200]:
    if only and only not in name: continue
    print(f"{c:4d} {name:18s} {shapes:60s} {frame}")
