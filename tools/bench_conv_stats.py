"""Forward / backward-data of one 3x3 layer WITH the fused BatchNorm sums, window kernel against implicit-GEMM kernel in one process.
usage: python tools/bench_conv_stats.py [Cin Cout [nslab]]"""
import ctypes
import sys
import time

import torch

sys.path.insert(0, "ws-mgmap_amd")
from wsmgmap import _abi  # noqa: E402

Cin, Cout = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 256)
nslab = int(sys.argv[3]) if len(sys.argv) > 3 else 64
B, H = 512, 24
x = torch.relu(torch.randn(B, H, H, Cin, device="cuda")).bfloat16()
w = (torch.randn(Cout, 3, 3, Cin, device="cuda") * 0.05).bfloat16()
y = torch.empty(B, H, H, Cout, device="cuda", dtype=torch.bfloat16)
stats = torch.zeros(nslab, 2, Cout, device="cuda", dtype=torch.float64)
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(with_stats):
    _abi.call("wsmg_conv2d_fwd_bf16_stats", P(x), P(w), None, P(y), 0, P(stats) if with_stats else None, nslab if with_stats else 0,
              B, H, H, Cin, Cout, 3, 3, 1, 1, H, H, st)


def timeit(fn, reps=30):
    t0 = time.time()
    while time.time() - t0 < 0.25:
        for _ in range(10): fn()
        torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for tile in (0, 1, 0, 1):
    _abi.lib().wsmg_conv_debug_win3_tile(tile)
    print(f"tile={tile}  plain {timeit(lambda: run(False)):.4f} ms   with BN sums {timeit(lambda: run(True)):.4f} ms")
