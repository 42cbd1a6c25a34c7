#!/usr/bin/env python3
"""GPU-box tool: do two persistent GRU launches on two streams overlap?  Times one launch, two launches back to back on
one stream, and two launches on two streams (separate buffers)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ws-mgmap_amd"))
import torch
from wsmgmap import ops, _abi
L = _abi.lib(); P = ops._p
T, N, H = int(os.environ.get("T", "64")), 8, 512
torch.manual_seed(0)
def mk():
    d = dict(gi=torch.randn(T, N, 3 * H, device="cuda"), whh=torch.randn(3 * H, H, device="cuda") * 0.04,
             bhh=torch.randn(3 * H, device="cuda") * 0.1, h0=torch.randn(N, H, device="cuda"), masks=torch.ones(T, N, device="cuda"),
             y=torch.empty(T, N, H, device="cuda"), saves=[torch.empty(T, N, H, device="cuda") for _ in range(4)],
             gy=torch.randn(T, N, H, device="cuda"), dgi=torch.empty(T, N, 3 * H, device="cuda"), dgh=torch.empty(T, N, 3 * H, device="cuda"),
             dh0=torch.empty(N, H, device="cuda"), ws=ops._rnn_workspace(L.wsmg_gru_workspace_bytes(T), torch.device("cuda")))
    return d
A, B = mk(), mk()
def fwd(d, st): _abi.call("wsmg_gru_fwd", P(d["gi"]), P(d["whh"]), P(d["bhh"]), P(d["h0"]), P(d["masks"]), T, N, H, P(d["y"]), *[P(s) for s in d["saves"]], P(d["ws"]), st)
def bwd(d, st): _abi.call("wsmg_gru_bwd", P(d["gy"]), None, P(d["whh"]), P(d["h0"]), P(d["masks"]), P(d["y"]), *[P(s) for s in d["saves"]], T, N, H, P(d["dgi"]), P(d["dgh"]), P(d["dh0"]), P(d["ws"]), st)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
import ctypes
h1, h2 = ctypes.c_void_p(s1.cuda_stream), ctypes.c_void_p(s2.cuda_stream)
def wall(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6
for name, k in (("fwd", fwd), ("bwd", bwd)):
    one = wall(lambda: k(A, h1))
    seq = wall(lambda: (k(A, h1), k(B, h1)))
    par = wall(lambda: (k(A, h1), k(B, h2)))
    print(f"GRU {name} T={T}: one launch {one:.0f} us | two on one stream {seq:.0f} us | two on two streams {par:.0f} us")
