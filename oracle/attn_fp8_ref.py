"""TEST INFRASTRUCTURE (oracle) — CPU restatement of the text attention in fp8 storage (BASELINE configs[4]).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product path
(ws-mgmap_amd/) never does.

Reference arithmetic (vlnce_baselines/models/mg_map_policy.py):
  :126-127  state_text_k_layer = Conv1d(256, 256, 1)           k_l = W_k x_l + b_k
  :173-178  _attn: logits = einsum(q, k); logits -= 1e8 * mask; softmax(logits / 16); out = einsum(attn, v)
with v = the instruction embedding itself (:229-232).  The reference is float32-only: the fp8 variant
of configs[4] is defined HERE as "x stored as OCP e4m3 with one per-tensor scale, everything else float32":
the embedding is quantised (round to nearest even, saturating at +-448, as torch.float8_e4m3fn does below 480)
and the reference formula is evaluated on the de-quantised values in float64.  Parity unpinned against the
reference itself for this config (it has no fp8 path); g5_attn.npz pins the float32 formula.
"""
import numpy as np
import torch

E4M3_MAX = 448.0


def quantize_e4m3(x, scale):
    """float32 array -> uint8 e4m3 codes of clamp(x / scale, +-448) (torch's CPU cast is the encoder)."""
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)) * np.float32(1.0 / scale)
    t = t.clamp(-E4M3_MAX, E4M3_MAX)
    return t.to(torch.float8_e4m3fn).view(torch.uint8).numpy()


def dequantize_e4m3(codes, scale):
    return torch.from_numpy(np.ascontiguousarray(codes)).view(torch.float8_e4m3fn).float().numpy().astype(np.float64) * scale


def attn_fp8(q, w_k, b_k, x_codes, x_scale, lengths, scale):
    """q [B,C], w_k [C,C] (out, in), b_k [C], x_codes [B,L,C] uint8, lengths [B] -> (out [B,C], attn [B,L]) float64."""
    x = dequantize_e4m3(x_codes, x_scale)                       # [B, L, C]
    q = q.astype(np.float64)
    k = x @ w_k.astype(np.float64).T + b_k.astype(np.float64)   # k_l = W_k x_l + b_k
    logits = np.einsum("bc,blc->bl", q, k)
    L = x.shape[1]
    mask = (np.arange(L)[None, :] >= np.asarray(lengths)[:, None]).astype(np.float64)
    logits = (logits - 1e8 * mask) * scale
    logits = logits - logits.max(axis=1, keepdims=True)
    e = np.exp(logits)
    attn = e / e.sum(axis=1, keepdims=True)
    out = np.einsum("bl,blc->bc", attn, x)
    return out, attn


def attn_fp8_grads(q, w_k, b_k, x_codes, x_scale, lengths, scale, dout, dattn=None):
    """Gradients of <out, dout> (+ <attn, dattn>) w.r.t. q, W_k, b_k and the DE-QUANTISED tokens x, by float64 autograd over the
    reference formula (the product path's straight-through gradient of x is this dx).  Returns (dq, dW_k, db_k, dx)."""
    x = torch.from_numpy(dequantize_e4m3(x_codes, x_scale)).requires_grad_(True)
    qt = torch.from_numpy(q.astype(np.float64)).requires_grad_(True)
    wt = torch.from_numpy(w_k.astype(np.float64)).requires_grad_(True)
    bt = torch.from_numpy(b_k.astype(np.float64)).requires_grad_(True)
    k = x @ wt.T + bt
    logits = torch.einsum("bc,blc->bl", qt, k)
    L = x.shape[1]
    mask = torch.from_numpy((np.arange(L)[None, :] >= np.asarray(lengths)[:, None]).astype(np.float64))
    a = torch.softmax((logits - 1e8 * mask) * scale, dim=1)
    out = torch.einsum("bl,blc->bc", a, x)
    obj = (out * torch.from_numpy(dout.astype(np.float64))).sum()
    if dattn is not None:
        obj = obj + (a * torch.from_numpy(dattn.astype(np.float64))).sum()
    obj.backward()
    return qt.grad.numpy(), wt.grad.numpy(), bt.grad.numpy(), x.grad.numpy()
