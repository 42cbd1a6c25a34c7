#!/usr/bin/env python3
"""GPU-box tool: stress + time the persistent GRU / LSTM kernels.  Runs the same sequence `reps`
times while a second stream keeps the chip busy with conv-engine launches, and checks that every
repeat is bitwise identical to the first (a stale hand-off would show up as a mismatch)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ws-mgmap_amd"))
import torch
from wsmgmap import ops, _abi

torch.manual_seed(0)
T, N, H = 64, 8, 512
gi = torch.randn(T, N, 3 * H, device="cuda")
whh = torch.randn(3 * H, H, device="cuda") * 0.04
bhh = torch.randn(3 * H, device="cuda") * 0.1
h0 = torch.randn(N, H, device="cuda")
masks = torch.ones(T, N, device="cuda"); masks[0] = 0; masks[17, 3] = 0
gy = torch.randn(T, N, H, device="cuda")
U, L = 8, 80
lgi = torch.randn(U, L, 2, 512, device="cuda")
lw = torch.randn(2, 512, 128, device="cuda") * 0.08
lb = torch.randn(2, 512, device="cuda") * 0.1
lens = torch.tensor([80, 37, 1, 64, 80, 12, 55, 79], device="cuda", dtype=torch.int32)
lgy = torch.randn(U, L, 256, device="cuda")

def run():
    g = gi.clone().requires_grad_(True); w = whh.clone().requires_grad_(True)
    y = ops.masked_gru(g, w, bhh, h0, masks)
    (y * gy).sum().backward()
    lg = lgi.clone().requires_grad_(True); lww = lw.clone().requires_grad_(True)
    o = ops.bilstm(lg, lww, lb, lens)
    (o * lgy).sum().backward()
    return [y.detach(), g.grad, w.grad, o.detach(), lg.grad, lww.grad]

NAMES = ["gru.y", "gru.dgi", "gru.dw_hh(torch GEMM)", "lstm.out", "lstm.dgi", "lstm.dw_hh(torch GEMM)"]
ref = run()
torch.cuda.synchronize()
# background load on another stream
side = torch.cuda.Stream()
x = torch.randn(256, 24, 24, 256, device="cuda").to(torch.bfloat16)
wconv = torch.randn(256, 256, 3, 3, device="cuda") * 0.02
bad = 0
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for i in range(reps):
    with torch.cuda.stream(side):
        for _ in range(4):
            ops.conv2d(x, wconv, None, 1, 1)
    out = run()
    for j, (a, b) in enumerate(zip(ref, out)):
        if not torch.equal(a, b):
            bad += 1
            d = (a - b).abs()
            nz = int((d > 0).sum())
            print(f"  repeat {i}: tensor {NAMES[j]} differs in {nz} elements, max abs {float(d.max()):.3e} (ref max {float(a.abs().max()):.3e})")
torch.cuda.synchronize()
print("SC1 =", os.environ.get("WSMG_RNN_SC1", "default"), "| mismatching tensors over", reps, "stressed repeats:", bad)
# timing (idle chip)
def tm(fn, n=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
print("gru fwd+bwd + lstm fwd+bwd per call: %.1f us" % tm(run))
