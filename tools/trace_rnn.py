import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ws-mgmap_amd"))
from wsmgmap import _abi
_abi.LIB_PATH = os.path.join(ROOT, "tools", "bin", "libwsmgmap_trace.so")
import torch
from wsmgmap import ops
L = _abi.lib(); P = ops._p; st = ops._stream
T, N, H = 64, 8, 512
torch.manual_seed(0)
gi = torch.randn(T, N, 3 * H, device="cuda"); whh = torch.randn(3 * H, H, device="cuda") * 0.04
bhh = torch.randn(3 * H, device="cuda") * 0.1; h0 = torch.randn(N, H, device="cuda")
masks = torch.ones(T, N, device="cuda"); masks[0] = 0
y = torch.empty(T, N, H, device="cuda"); saves = [torch.empty(T, N, H, device="cuda") for _ in range(4)]
ws = ops._rnn_workspace(L.wsmg_gru_workspace_bytes(T), gi.device)
tr = torch.zeros(T * 8, dtype=torch.int64, device="cuda")
L.wsmg_debug_set_trace.argtypes = [ctypes.c_void_p]
def fwd(): _abi.call("wsmg_gru_fwd", P(gi), P(whh), P(bhh), P(h0), P(masks), T, N, H, P(y), *[P(s) for s in saves], P(ws), st())
for _ in range(3): fwd()
torch.cuda.synchronize()
assert L.wsmg_debug_set_trace(ctypes.c_void_p(tr.data_ptr())) == 0
fwd(); torch.cuda.synchronize()
def report(title, names):
    t = tr.view(T, 8).cpu().double() * 10.0  # ns (100 MHz)
    print(title)
    for i, n in enumerate(names):
        d = (t[5:60, i + 1] - t[5:60, i])
        print(f"  {n:28s} mean {d.mean():8.0f} ns  min {d.min():6.0f} max {d.max():6.0f}")
    d = (t[6:60, 0] - t[5:59, 0]).abs()
    print(f"  step period                  mean {d.mean():8.0f} ns")
report("GRU fwd", ["poll+stage+sync", "LDS read + FMA", "butterfly", "gates + stores"])
gy = torch.randn(T, N, H, device="cuda")
dgi = torch.empty(T, N, 3 * H, device="cuda"); dgh = torch.empty_like(dgi); dh0 = torch.empty(N, H, device="cuda")
tr.zero_()
_abi.call("wsmg_gru_bwd", P(gy), None, P(whh), P(h0), P(masks), P(y), *[P(s) for s in saves], T, N, H, P(dgi), P(dgh), P(dh0), P(ws), st())
torch.cuda.synchronize()
report("GRU bwd", ["elementwise + stores", "grid barrier", "global reads + FMA", "butterfly + carry"])
