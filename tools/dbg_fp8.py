import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ws-mgmap_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from wsmgmap import ops, _abi
from oracle import attn_fp8_ref as ar
from test_gpu_kernels import _cfg5_inputs
T = torch.from_numpy
for B, L in ((64, 160), (3, 37)):
    q, w, b, x, lengths = _cfg5_inputs(B=B, L=L, seed=B + L)
    x_scale = float(np.float32(np.abs(x).max()) / np.float32(448.0))
    codes = ar.quantize_e4m3(x, x_scale)
    xt = T(x).cuda()
    xs_t = (xt.abs().amax() / 448.0).clamp_min(1e-30).reshape(1).float()
    print("scale", x_scale, float(xs_t), x_scale == float(xs_t))
    x_q = torch.empty(B, L, 256, device="cuda", dtype=torch.uint8)
    _abi.call("wsmg_quantize_e4m3_dev", ops._p(xt), xt.numel(), ops._p(xs_t), ops._p(x_q), ops._stream())
    print("codes differ:", int((x_q.cpu().numpy() != codes).sum()))
    out_ref, attn_ref = ar.attn_fp8(q, w, b, codes, x_scale, lengths, 1.0 / 16)
    out, attn = ops.attn_fp8_fused(T(q).cuda(), T(w).cuda(), T(b).cuda(), T(codes).cuda(), x_scale, T(lengths).cuda(), 1 / 16)
    print("fused: attn err", np.abs(attn.cpu().numpy() - attn_ref).max(), "out err", np.abs(out.cpu().numpy() - out_ref).max())
    out, attn = ops.attention_fp8(T(q).cuda(), T(w.reshape(256, 256, 1)).cuda(), T(b).cuda(), xt, T(lengths).cuda(), 1 / 16)
    e = np.abs(attn.cpu().numpy() - attn_ref)
    print("op: attn err", e.max(), "rows with err>2e-5:", np.where(e.max(1) > 2e-5)[0][:10], "lengths", lengths[np.where(e.max(1) > 2e-5)[0][:10]])
    # second call
    out2, attn2 = ops.attention_fp8(T(q).cuda(), T(w.reshape(256, 256, 1)).cuda(), T(b).cuda(), xt, T(lengths).cuda(), 1 / 16)
    print("second call err", np.abs(attn2.cpu().numpy() - attn_ref).max())
