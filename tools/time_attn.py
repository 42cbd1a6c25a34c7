#!/usr/bin/env python3
"""GPU-box tool: time the map-stage attention (keys are values, B=512, I=576, C=256, bf16) forward and backward."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ws-mgmap_amd"))
import torch
from wsmgmap import ops, _abi
P, st = ops._p, ops._stream
B, I, C = 512, 576, 256
torch.manual_seed(0)
x = torch.randn(B, I, C, device="cuda").bfloat16()
q = torch.randn(B, C, device="cuda"); dout = torch.randn(B, C, device="cuda"); dattn = torch.randn(B, I, device="cuda") * 0.1
out = torch.empty(B, C, device="cuda"); attn = torch.empty(B, I, device="cuda")
dq = torch.empty(B, C, device="cuda"); dx = torch.empty_like(x)
def fwd(): _abi.call("wsmg_attn_fwd_bf16", P(q), P(x), P(x), None, 1 / 16, B, I, C, P(out), P(attn), st())
def bwd(): _abi.call("wsmg_attn_bwd_bf16", P(q), P(x), P(x), P(attn), P(dout), P(dattn), 1 / 16, B, I, C, P(dq), P(dx), P(dx), st())
def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
mb = B * I * C * 2 / 1e6
tf, tb = timeit(fwd), timeit(bwd)
print(f"map attention B={B} I={I}: fwd {tf:.1f} us ({mb / tf:.2f} TB/s of {mb:.0f} MB)  bwd {tb:.1f} us ({2 * mb / tb:.2f} TB/s of {2 * mb:.0f} MB)")
