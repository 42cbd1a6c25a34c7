"""Forward of the k8 s2 stem (64 -> 64, 100 x 100, B = 512) with and without the fused BatchNorm sums; input randn or relu(randn)."""
import ctypes, sys, time
import torch
sys.path.insert(0, "ws-mgmap_amd")
from wsmgmap import _abi
B, H, C = 512, 100, 64
OH = 50
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
w = (torch.randn(C, 8, 8, C, device="cuda") * 0.02).bfloat16()
y = torch.empty(B, OH, OH, C, device="cuda", dtype=torch.bfloat16)
stats = torch.zeros(64, 2, C, device="cuda", dtype=torch.float64)


def timeit(fn, reps=20):
    t0 = time.time()
    while time.time() - t0 < 0.3:
        for _ in range(5): fn()
        torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for data in ("randn", "relu"):
    x = torch.randn(B, H, H, C, device="cuda")
    if data == "relu": x = torch.relu(x)
    x = x.bfloat16()
    for with_stats in (False, True):
        f = lambda: _abi.call("wsmg_conv2d_fwd_bf16_stats", P(x), P(w), None, P(y), 0, P(stats) if with_stats else None, 64 if with_stats else 0,
                              B, H, H, C, C, 8, 8, 2, 3, OH, OH, st)
        print(f"{data:6s} stats={with_stats}: {timeit(f):.4f} ms")
