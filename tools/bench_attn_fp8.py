#!/usr/bin/env python3
"""GPU-box tool: BASELINE configs[4] — text attention, B=64, L=160, e4m3 tokens — time and bandwidth of the fused
kernel (algorithmic bytes per row = 256*L token bytes + 256*4 query + (256+L)*4 outputs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ws-mgmap_amd"))
import torch
from wsmgmap import ops, _abi
P = ops._p; st = ops._stream
for B, L in ((64, 160), (512, 80), (4096, 160)):
    C = 256
    torch.manual_seed(0)
    q = torch.randn(B, C, device="cuda"); w = torch.randn(C, C, device="cuda") / 16; b = torch.randn(C, device="cuda") * 0.1
    x = torch.randn(B, L, C, device="cuda")
    xs = float(x.abs().max() / 448.0)
    codes = ops.quantize_e4m3(x, xs)
    lengths = torch.full((B,), L, dtype=torch.int32, device="cuda")
    qf = (q @ w).contiguous(); qb = (q @ b).contiguous()
    out = torch.empty(B, C, device="cuda"); attn = torch.empty(B, L, device="cuda")
    def kern(): _abi.call("wsmg_attn_fp8_fused_fwd", P(qf), P(qb), P(codes), xs, P(lengths), 1 / 16, B, L, C, P(out), P(attn), st())
    def whole(): ops.attn_fp8_fused(q, w, b, codes, xs, lengths, 1 / 16)
    res = []
    for f in (kern, whole):
        for _ in range(5): f()
        torch.cuda.synchronize()
        a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50): f()
        e.record(); torch.cuda.synchronize()
        res.append(a.elapsed_time(e) / 50 * 1e3)
    by = B * (C * L + C * 4 + (C + L) * 4)
    print(f"B={B} L={L}: fused kernel {res[0]:.1f} us = {by / res[0] / 1e3:.0f} GB/s ({by / res[0] / 1e3 / 8000 * 100:.1f} % of 8 TB/s); with the query fold GEMMs {res[1]:.1f} us")
