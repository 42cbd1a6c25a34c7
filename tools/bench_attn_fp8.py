#!/usr/bin/env python3
"""GPU-box tool: BASELINE configs[4] — text attention, B=64, L=160, e4m3 tokens — time and bandwidth of the fused
kernel (algorithmic bytes per row = 256*L token bytes + 256*4 query + (256+L)*4 outputs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ws-mgmap_amd"))
import torch
from wsmgmap import ops, _abi
P = ops._p; st = ops._stream
def _floor():
    t = torch.zeros(256, device="cuda")
    f = lambda: _abi.call("wsmg_relu_fwd", P(t), P(t), 256, st())
    for _ in range(20): f()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200): f()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / 200 * 1e3
print(f"launch floor of this loop (a 256-element ReLU launched back to back through the same ctypes path): {_floor():.1f} us per launch")
for B, L in ((64, 160), (512, 80), (4096, 160), (16384, 160)):
    C = 256
    torch.manual_seed(0)
    q = torch.randn(B, C, device="cuda"); w = torch.randn(C, C, device="cuda") / 16; b = torch.randn(C, device="cuda") * 0.1
    x = torch.randn(B, L, C, device="cuda")
    xs = float(x.abs().max() / 448.0)
    codes = ops.quantize_e4m3(x, xs)
    lengths = torch.full((B,), L, dtype=torch.int32, device="cuda")
    qf = (q @ w).contiguous()
    out = torch.empty(B, C, device="cuda"); attn = torch.empty(B, L, device="cuda")
    xs_t = torch.full((1,), xs, device="cuda")
    ws, ticket = ops._fp8_scratch(B, L, q.device)
    dout = torch.randn(B, C, device="cuda"); dqf = torch.empty(B, C, device="cuda"); dx = torch.empty(B, L, C, device="cuda")
    def kern(): _abi.call("wsmg_attn_fp8_fwd", P(qf), P(codes), P(xs_t), P(lengths), 1 / 16, B, L, C, P(out), P(attn), P(ws), P(ticket), st())
    def whole(): ops.attn_fp8_fused(q, w, b, codes, xs_t, lengths, 1 / 16)
    def bwd(): _abi.call("wsmg_attn_fp8_bwd", P(qf), P(codes), P(xs_t), P(attn), P(dout), None, 1 / 16, B, L, C, P(dqf), P(dx), st())
    def fold(): _abi.call("wsmg_attn_fp8_fold", P(q), P(w), B, C, 0, P(qf), st())
    res = []
    for f in (kern, whole, bwd, fold):
        for _ in range(5): f()
        torch.cuda.synchronize()
        a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50): f()
        e.record(); torch.cuda.synchronize()
        res.append(a.elapsed_time(e) / 50 * 1e3)
    by = B * (C * L + C * 4 + (C + L) * 4)
    byb = B * (C * L + 2 * C * 4 + L * 4 + C * 4 + L * C * 4)   # backward: bytes in, q_f, dout, attn -> d q_f, dx (float32)
    print(f"B={B} L={L} ({_abi.lib().wsmg_attn_fp8_splits(B, L)} workgroups per row): forward {res[0]:.1f} us = {by / res[0] / 1e3:.0f} GB/s "
          f"({by / res[0] / 1e3 / 8000 * 100:.1f} % of 8 TB/s); with the MFMA query fold {res[1]:.1f} us (fold alone {res[3]:.1f}); "
          f"backward {res[2]:.1f} us = {byb / res[2] / 1e3:.0f} GB/s ({byb / res[2] / 1e3 / 8000 * 100:.1f} %)")

# ---- the shared-set form on the fp8 matrix pipe (csrc/wsmg_attn_fp8_mfma.hip): ONE launch for S = Q K^T (v_mfma ... fp8_fp8), the
# softmax and O = P V (bf16 pipe); the keys are the projected keys of each unique instruction (no fold launch).  Algorithmic bytes:
# U sets x L x 256 bytes of keys + of values, B queries (256 B) + outputs.
print("shared instruction sets, fp8 matrix pipe (one launch):")
for B, U, L in ((64, 8, 160), (512, 8, 80), (4096, 64, 160)):
    C = 256
    torch.manual_seed(1)
    q = torch.randn(B, C, device="cuda"); k = torch.randn(U, L, C, device="cuda"); v = torch.randn(U, L, C, device="cuda")
    inv = (torch.arange(B, device="cuda") % U)
    lens = torch.full((U,), L, dtype=torch.int32, device="cuda")
    sc = [float(t.abs().max() / 448.0) for t in (q, k, v)]
    qc, kc, vc = (ops.quantize_e4m3(t, s) for t, s in zip((q, k, v), sc))
    st3 = [torch.full((1,), s, device="cuda") for s in sc]
    order = torch.argsort(inv, stable=True).to(torch.int32)
    start = torch.zeros(U + 1, device="cuda", dtype=torch.int32); start[1:] = torch.cumsum(torch.bincount(inv, minlength=U), 0).to(torch.int32)
    out = torch.empty(B, C, device="cuda"); attn = torch.empty(B, L, device="cuda")
    def kern(): _abi.call("wsmg_attn_fp8_mfma_fwd", P(qc), P(st3[0]), P(kc), P(st3[1]), P(vc), P(st3[2]), P(lens), P(order), P(start), 1 / 16, B, U, L, C, P(out), P(attn), st())
    def whole(): ops.attention_fp8_shared(q, k, v, lens, inv, 1 / 16)
    res = []
    for f in (kern, whole):
        for _ in range(5): f()
        torch.cuda.synchronize()
        a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50): f()
        e.record(); torch.cuda.synchronize()
        res.append(a.elapsed_time(e) / 50 * 1e3)
    by = 2 * U * L * C + B * (C + C * 4 + L * 4)
    fl = 2.0 * B * L * C * 2
    print(f"  B={B} U={U} L={L}: kernel {res[0]:.1f} us ({by / res[0] / 1e3:.0f} GB/s algorithmic, {fl / res[0] / 1e6:.2f} TFLOP/s); "
          f"with the three quantisations and the row grouping (ops.attention_fp8_shared) {res[1]:.1f} us")
