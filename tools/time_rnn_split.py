#!/usr/bin/env python3
"""GPU-box tool: the N = 8 batch slots of a GRU launch are independent sequences.  Is one launch of 8 slower than 2 launches of 4
(or 4 of 2) on as many streams, each a persistent 32-workgroup kernel whose hand-off latency hides under the others' steps?"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ws-mgmap_amd"))
import torch
from wsmgmap import ops, _abi
L = _abi.lib(); P = ops._p
T, H = int(os.environ.get("T", "64")), 512
torch.manual_seed(0)
whh = torch.randn(3 * H, H, device="cuda") * 0.04; bhh = torch.randn(3 * H, device="cuda") * 0.1
def mk(N):
    return dict(N=N, gi=torch.randn(T, N, 3 * H, device="cuda"), h0=torch.randn(N, H, device="cuda"), masks=torch.ones(T, N, device="cuda"),
                y=torch.empty(T, N, H, device="cuda"), saves=[torch.empty(T, N, H, device="cuda") for _ in range(4)],
                gy=torch.randn(T, N, H, device="cuda"), dgi=torch.empty(T, N, 3 * H, device="cuda"), dgh=torch.empty(T, N, 3 * H, device="cuda"),
                dh0=torch.empty(N, H, device="cuda"), ws=ops._rnn_workspace(L.wsmg_gru_workspace_bytes(T), torch.device("cuda")))
def fwd(d, st): _abi.call("wsmg_gru_fwd", P(d["gi"]), P(whh), P(bhh), P(d["h0"]), P(d["masks"]), T, d["N"], H, P(d["y"]), *[P(s) for s in d["saves"]], P(d["ws"]), st)
def bwd(d, st): _abi.call("wsmg_gru_bwd", P(d["gy"]), None, P(whh), P(d["h0"]), P(d["masks"]), P(d["y"]), *[P(s) for s in d["saves"]], T, d["N"], H, P(d["dgi"]), P(d["dgh"]), P(d["dh0"]), P(d["ws"]), st)
streams = [torch.cuda.Stream() for _ in range(4)]
hs = [ctypes.c_void_p(s.cuda_stream) for s in streams]
def wall(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6
for parts in (1, 2, 4):
    ds = [mk(8 // parts) for _ in range(parts)]
    for name, k in (("fwd", fwd), ("bwd", bwd)):
        if name == "bwd":
            for i, d in enumerate(ds): fwd(d, hs[i])
            torch.cuda.synchronize()
        t = wall(lambda: [k(d, hs[i]) for i, d in enumerate(ds)])
        print(f"GRU {name} T={T}: {parts} launch(es) of N={8 // parts} on {parts} stream(s): {t:.0f} us ({t / T:.2f} us per step)")
