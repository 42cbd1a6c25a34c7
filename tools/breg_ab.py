#!/usr/bin/env python3
"""GPU-box tool, round 6 experiment: the 3 x 3 window convolution with its weights loaded global -> registers (BREG) against the LDS-staged
form, per layer, forward and backward-data: bit-identity first, then interleaved timings (plain launches, no mixed tiles in either arm).
    python tools/breg_ab.py [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ws-mgmap_amd"))
os.environ["WSMG_CONV_WIN3_MIXED"] = "0"
import torch
from wsmgmap import _abi, ops
from wsmgmap.ops.core import _p, _stream

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
L = _abi.lib()
B = 512
LAYERS = [("enc6_k3", 128, 256), ("encoded_lin_k3", 256, 128), ("cated_k3", 256, 256)]


def run(bwd, x, w, y, Cin, Cout):
    dims = (B, 24, 24, Cin, Cout, 3, 3, 1, 1, 24, 24)
    if bwd:
        _abi.call("wsmg_conv2d_bwd_data_bf16", _p(x), _p(w), _p(y), 0, *dims, _stream())
    else:
        _abi.call("wsmg_conv2d_fwd_bf16", _p(x), _p(w), None, _p(y), 0, *dims, _stream())


def timeit(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


g = torch.Generator(device="cuda"); g.manual_seed(1)
for name, Cin, Cout in LAYERS:
    for bwd in (0, 1):
        cs, cd = (Cout, Cin) if bwd else (Cin, Cout)         # channels of the source / destination tensor of this launch
        x = torch.randn(B, 24, 24, cs, device="cuda", generator=g).to(torch.bfloat16)
        w = (torch.randn(cd, 3, 3, cs, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
        y0 = torch.empty(B, 24, 24, cd, device="cuda", dtype=torch.bfloat16)
        y1 = torch.empty_like(y0)
        L.wsmg_conv_debug_win3_breg(0); run(bwd, x, w, y0, Cin, Cout)
        L.wsmg_conv_debug_win3_breg(1); run(bwd, x, w, y1, Cin, Cout)
        torch.cuda.synchronize()
        same = bool(torch.equal(y0, y1))
        ts = {0: [], 1: []}
        for _ in range(3):
            for arm in (0, 1):
                L.wsmg_conv_debug_win3_breg(arm)
                ts[arm].append(timeit(lambda: run(bwd, x, w, y0, Cin, Cout)))
        gf = 2.0 * B * 576 * Cin * Cout * 9 / 1e9
        print("%-16s %s  bit-identical %s   LDS-staged %s ms (%.0f TF)   BREG %s ms (%.0f TF)" % (
            name, "bwdD" if bwd else "fwd ", same, ["%.4f" % t for t in ts[0]], gf / min(ts[0]), ["%.4f" % t for t in ts[1]], gf / min(ts[1])))
L.wsmg_conv_debug_win3_breg(-1)
