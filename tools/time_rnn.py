#!/usr/bin/env python3
"""GPU-box tool: per-kernel time of the persistent GRU / LSTM launches (HIP events, no other load)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ws-mgmap_amd"))
import torch
from wsmgmap import ops, _abi
L = _abi.lib(); P = ops._p; st = ops._stream
T, N, H = int(os.environ.get("T", "64")), 8, 512
torch.manual_seed(0)
gi = torch.randn(T, N, 3 * H, device="cuda"); whh = torch.randn(3 * H, H, device="cuda") * 0.04
bhh = torch.randn(3 * H, device="cuda") * 0.1; h0 = torch.randn(N, H, device="cuda")
masks = torch.ones(T, N, device="cuda"); masks[0] = 0
y = torch.empty(T, N, H, device="cuda"); saves = [torch.empty(T, N, H, device="cuda") for _ in range(4)]
gy = torch.randn(T, N, H, device="cuda")
dgi = torch.empty(T, N, 3 * H, device="cuda"); dgh = torch.empty_like(dgi); dh0 = torch.empty(N, H, device="cuda")
ws = ops._rnn_workspace(L.wsmg_gru_workspace_bytes(T), gi.device)
def fwd(): _abi.call("wsmg_gru_fwd", P(gi), P(whh), P(bhh), P(h0), P(masks), T, N, H, P(y), *[P(s) for s in saves], P(ws), st())
def bwd(): _abi.call("wsmg_gru_bwd", P(gy), None, P(whh), P(h0), P(masks), P(y), *[P(s) for s in saves], T, N, H, P(dgi), P(dgh), P(dh0), P(ws), st())
def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
tf, tb = timeit(fwd), timeit(bwd)
print(f"GRU T={T} N={N}: fwd {tf:.1f} us ({tf / T:.2f} us/step)  bwd {tb:.1f} us ({tb / T:.2f} us/step)")
U, Lt = 8, int(os.environ.get("L", "80"))
lgi = torch.randn(U, Lt, 2, 512, device="cuda", requires_grad=True); lw = torch.randn(2, 512, 128, device="cuda") * 0.08
lb = torch.randn(2, 512, device="cuda") * 0.1; lens = torch.full((U,), Lt, device="cuda", dtype=torch.int32)
lgy = torch.randn(U, Lt, 256, device="cuda")
def lf():
    return ops.bilstm(lgi, lw, lb, lens)
tlf = timeit(lf)
def lfb():
    o = ops.bilstm(lgi, lw, lb, lens); (o * lgy).sum().backward()
tlfb = timeit(lfb)
print(f"LSTM L={Lt} U={U}: fwd (op) {tlf:.1f} us  fwd+bwd (op) {tlfb:.1f} us")
