#!/usr/bin/env python3
"""GPU-box tool: time every conv-engine layer shape of the update path (B=512) separately —
forward, backward-data, backward-weight — with HIP events, and print algorithmic TFLOP/s.
    python tools/bench_conv.py [--reps 5] [--only NAME]
"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ws-mgmap_amd"))
import torch
from wsmgmap import _abi, ops

# name, Cin, Cout, k, stride, pad, H (input), needs_dx
LAYERS = [
    ("enc0_k8s2", 64, 64, 8, 2, 3, 100, False),
    ("enc3_k5s2", 64, 128, 5, 2, 1, 50, True),
    ("enc6_k3", 128, 256, 3, 1, 1, 24, True),
    ("encoded_lin_k3", 256, 128, 3, 1, 1, 24, True),
    ("orig0_k3", 256, 64, 3, 1, 1, 24, True),
    ("orig1_k3", 64, 64, 3, 1, 1, 24, True),
    ("stem_k7s2", 256, 64, 7, 2, 3, 24, True),
    ("block_k3_6", 64, 64, 3, 1, 1, 6, True),
    ("up0_k3_12", 128, 128, 3, 1, 1, 12, True),
    ("orig2_k3", 192, 64, 3, 1, 1, 24, True),
    ("convT_adj_k4s2", 32, 64, 4, 2, 1, 48, True),
    ("cls_k3_48", 32, 32, 3, 1, 1, 48, True),
    ("classified_k3", 32, 128, 3, 1, 1, 24, True),
    ("cated_k3", 256, 256, 3, 1, 1, 24, True),
    ("l0_1x1_12", 64, 64, 1, 1, 0, 12, True),
    ("l1_1x1_6", 64, 64, 1, 1, 0, 6, True),
]
# Not in the update any more, kept for --only: the classifier's 1 x 1 head runs inside the fused tail (csrc/wsmg_cls_tail.hip) and
# text_map_k_layer is folded into the attention's query (ops.attention_folded); the tables of rounds 1-3 still listed them.
EXTRA = [
    ("cls_1x1_48", 32, 32, 1, 1, 0, 48, True),
    ("mapk_1x1", 256, 256, 1, 1, 0, 24, True),
]

def timeit(fn, reps, warm_s=0.25):
    # a cold GPU runs its first kernels of a process at a lower clock (cated_k3 forward: 0.44 ms cold, 0.36 ms warm): spin first
    import time
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    while time.time() - t0 < warm_s:
        for _ in range(10): fn()
        torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default=None)
    ap.add_argument("--B", type=int, default=512)
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--wgrad", default="slabs", choices=["slabs", "atomics"],
                    help="slabs (default): the product's deterministic form, ops._weight_grad = slab launch + ordered reduce launch; "
                         "atomics: the A/B form (float atomics into a zeroed dW; the tables of rounds 1-3 timed this one)")
    ap.add_argument("--data", default="randn", choices=["randn", "relu"],
                    help="relu: x and dy half zeros, as the update's post-ReLU activations and masked gradients are (the kernels' clock, "
                         "and with it their ranking, depends on the data)")
    a = ap.parse_args()
    B = a.B
    st = ops._stream
    tot = {"fwd": 0.0, "bwdD": 0.0, "wgrad": 0.0}
    print(f"{'layer':18s} {'GF':>8s} | {'fwd ms':>8s} {'TF':>6s} | {'bwdD ms':>8s} {'TF':>6s} | {'wgrad ms':>8s} {'TF':>6s}")
    for name, Cin, Cout, k, s, p, H, need_dx in LAYERS + (EXTRA if a.only else []):
        if a.only and a.only not in name: continue
        OH = (H + 2 * p - k) // s + 1
        dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
        x = torch.randn(B, H, H, Cin, device="cuda")
        if a.data == "relu": x = torch.relu(x)
        x = x.to(dt)
        w = (torch.randn(Cout, k, k, Cin, device="cuda") * 0.05).to(dt)
        wi = w.permute(3, 1, 2, 0).contiguous()
        y = torch.empty(B, OH, OH, Cout, device="cuda", dtype=dt)
        dy = torch.randn(B, OH, OH, Cout, device="cuda")
        if a.data == "relu": dy = dy * (torch.rand_like(dy) < 0.5)
        dy = dy.to(dt)
        dx = torch.empty_like(x)
        dw = torch.zeros(Cout, k, k, Cin, device="cuda")
        P = ops._p
        gf = 2.0 * B * OH * OH * Cout * Cin * k * k / 1e9
        args = (B, H, H, Cin, Cout, k, k, s, p, OH, OH)
        if a.dtype == "bf16":
            t_f = timeit(lambda: _abi.call("wsmg_conv2d_fwd_bf16", P(x), P(w), None, P(y), 0, *args, st()), a.reps)
            t_d = timeit(lambda: _abi.call("wsmg_conv2d_bwd_data_bf16", P(dy), P(wi), P(dx), 0, *args, st()), a.reps) if need_dx else 0.0
            if a.wgrad == "slabs":
                t_w = timeit(lambda: ops._weight_grad("_bf16", x, dy, args, 0.0, Cin), a.reps)
            else:
                t_w = timeit(lambda: _abi.call("wsmg_conv2d_bwd_weight_bf16", P(x), P(dy), P(dw), *args, st()), a.reps)
        else:
            t_f = timeit(lambda: _abi.call("wsmg_conv2d_fwd", P(x), P(w), None, P(y), *args, st()), a.reps)
            t_d = timeit(lambda: _abi.call("wsmg_conv2d_bwd_data", P(dy), P(wi), P(dx), *args, st()), a.reps) if need_dx else 0.0
            t_w = timeit(lambda: _abi.call("wsmg_conv2d_bwd_weight", P(x), P(dy), P(dw), *args, st()), a.reps)
        tot["fwd"] += t_f; tot["bwdD"] += t_d; tot["wgrad"] += t_w
        tf = lambda t: gf / t if t > 0 else 0.0
        print(f"{name:18s} {gf:8.1f} | {t_f:8.3f} {tf(t_f):6.1f} | {t_d:8.3f} {tf(t_d):6.1f} | {t_w:8.3f} {tf(t_w):6.1f}")
    print("total ms: fwd %.2f bwdD %.2f wgrad %.2f  sum %.2f" % (tot["fwd"], tot["bwdD"], tot["wgrad"], sum(tot.values())))

if __name__ == "__main__":
    main()
