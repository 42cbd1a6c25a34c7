#!/usr/bin/env python3
"""GPU-box tool: the update path's fused heads, forward and backward, alone (HIP events)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ws-mgmap_amd"))
import torch
from wsmgmap import ops
B, K, A = 512, 512, 2
x = torch.randn(B, K, device="cuda", requires_grad=True)
fc_mean = torch.nn.Linear(K, A).cuda(); prog_pred = torch.nn.Linear(K, 1).cuda()
progress = torch.rand(B, 1, device="cuda")
def fb():
    pred, prog, rows = ops.update_heads(x, fc_mean, prog_pred, progress)
    (pred.sum() + prog.sum() + rows.sum()).backward()
for _ in range(5): fb()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(20): fb()
    torch.cuda.synchronize()
for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:6]:
    print("%-60s %8.1f us x %d" % (e.key[:60], e.device_time_total / max(e.count, 1), e.count))
