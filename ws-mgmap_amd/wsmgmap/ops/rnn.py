"""wsmgmap.ops.rnn — the persistent masked-GRU and packed bi-LSTM sequence kernels.
"""
import ctypes

import torch

from .. import _abi
from ..debug import sw
from .core import _p, _raw_stream, _stream, _req, _f32, _sfx, _workspace, _rows_of, _conv_out, _launch, _zeros_f32, _prelaid, _join_side_at_end, TokenGradSink


def _rnn_workspace(nbytes, device):
    """Barrier words + exchange image of the persistent RNN kernels.  debug.sw.rnn_poison (stress tool) fills
    it with NaN first so that any stale or missed hand-off read poisons the results visibly."""
    ws = torch.empty((int(nbytes) + 3) // 4, device=device, dtype=torch.float32)
    if sw.rnn_poison:
        ws.fill_(float("nan"))
    return ws


check_rnn_status = _abi.check_rnn_status


def _rnn_launched():
    pass


class _MaskedGRU(torch.autograd.Function):
    """gi [T,N,3H] (input projections), w_hh [3H,H], b_hh [3H], h0 [N,H], masks [T,N] -> y [T,N,H]."""

    @staticmethod
    def forward(ctx, gi, w_hh, b_hh, h0, masks):
        _req(gi, w_hh, b_hh, h0, masks)
        _f32(gi, w_hh, b_hh, h0, masks)
        T, N, H3 = gi.shape
        H = H3 // 3
        dev = gi.device
        y = torch.empty(T, N, H, device=dev, dtype=torch.float32)
        saves = [torch.empty(T, N, H, device=dev, dtype=torch.float32) for _ in range(4)]
        sync = _rnn_workspace(_abi.lib().wsmg_gru_workspace_bytes(T), dev)
        _abi.call("wsmg_gru_fwd", _p(gi), _p(w_hh), _p(b_hh), _p(h0), _p(masks), T, N, H, _p(y),
                  *[_p(s) for s in saves], _p(sync), _stream())
        _rnn_launched()
        ctx.save_for_backward(w_hh, h0, masks, y, *saves)
        return y

    @staticmethod
    def backward(ctx, dy):
        w_hh, h0, masks, y, sr, sz, sn, sghn = ctx.saved_tensors
        T, N, H = y.shape
        dev = y.device
        dy = dy.contiguous()
        dgi = torch.empty(T, N, 3 * H, device=dev, dtype=torch.float32)
        dgh = torch.empty(T, N, 3 * H, device=dev, dtype=torch.float32)
        dh0 = torch.empty(N, H, device=dev, dtype=torch.float32)
        sync = _rnn_workspace(_abi.lib().wsmg_gru_workspace_bytes(T), dev)
        _abi.call("wsmg_gru_bwd", _p(dy), None, _p(w_hh), _p(h0), _p(masks), _p(y), _p(sr), _p(sz), _p(sn), _p(sghn),
                  T, N, H, _p(dgi), _p(dgh), _p(dh0), _p(sync), _stream())
        _rnn_launched()
        hprev = torch.cat([h0.unsqueeze(0), y[:-1]], dim=0) * masks.unsqueeze(-1)
        g2 = dgh.view(T * N, 3 * H)
        dw_hh = g2.t() @ hprev.view(T * N, H)
        db_hh = g2.sum(dim=0)
        return dgi, dw_hh, db_hh, dh0, None


def masked_gru(gi, w_hh, b_hh, h0, masks):
    """Whole-sequence masked GRU in one persistent launch.  Returns y [T,N,H]; final state = y[-1]."""
    return _MaskedGRU.apply(gi.contiguous(), w_hh.contiguous(), b_hh.contiguous(), h0.contiguous(), masks.contiguous())


# ----------------------------------------------------------------------------- persistent packed bi-LSTM
class _BiLSTM(torch.autograd.Function):
    """gi [U,L,2,4H], w_hh [2,4H,H], b_hh [2,4H], lengths int32 [U] -> out [U,L,2H]."""

    @staticmethod
    def forward(ctx, gi, w_hh, b_hh, lengths):
        _req(gi, w_hh, b_hh, lengths)
        _f32(gi, w_hh, b_hh)
        U, L, _, H4 = gi.shape
        H = H4 // 4
        dev = gi.device
        out = torch.empty(U, L, 2 * H, device=dev, dtype=torch.float32)
        sg = torch.zeros(2, U, L, 4, H, device=dev, dtype=torch.float32)
        sc = torch.zeros(2, U, L, H, device=dev, dtype=torch.float32)
        ws = _rnn_workspace(_abi.lib().wsmg_lstm_workspace_bytes(L), dev)
        _abi.call("wsmg_lstm_fwd", _p(gi), _p(w_hh), _p(b_hh), _p(lengths), U, L, H, _p(out), _p(sg), _p(sc), _p(ws), _stream())
        _rnn_launched()
        ctx.save_for_backward(w_hh, lengths, out, sg, sc)
        return out

    @staticmethod
    def backward(ctx, dout):
        w_hh, lengths, out, sg, sc = ctx.saved_tensors
        U, L, H2 = out.shape
        H = H2 // 2
        dev = out.device
        dout = dout.contiguous()
        dg = torch.empty(U, L, 2, 4 * H, device=dev, dtype=torch.float32)
        ws = _rnn_workspace(_abi.lib().wsmg_lstm_workspace_bytes(L), dev)
        _abi.call("wsmg_lstm_bwd", _p(dout), _p(w_hh), _p(lengths), _p(sg), _p(sc), U, L, H, _p(dg), _p(ws), _stream())
        _rnn_launched()
        zero = torch.zeros(U, 1, H, device=dev, dtype=torch.float32)
        hprev_f = torch.cat([zero, out[:, :-1, :H]], dim=1)       # state before step t (forward direction)
        hprev_r = torch.cat([out[:, 1:, H:], zero], dim=1)        # state before step t (reverse direction)
        # one direction's gate gradients as a contiguous [U L, 4H] matrix first: on the strided view dg[:, :, d] (row pitch 8H)
        # the GEMM library picked a 32 x 16 tile kernel that took 340 us for this 0.7 GFLOP product (beside the map stack's
        # backward, on the instruction stream)
        dgd = dg.permute(2, 0, 1, 3).contiguous().view(2, U * L, 4 * H)
        dw = torch.stack([dgd[0].t() @ hprev_f.reshape(U * L, H), dgd[1].t() @ hprev_r.reshape(U * L, H)])
        from .heads import colsum_multi    # stock sum over (0, 1) of the strided view: 324 us on the instruction stream
        db = torch.stack(colsum_multi([dgd[0], dgd[1]]))
        return dg, dw, db, None


def bilstm(gi, w_hh, b_hh, lengths):
    """Packed bidirectional LSTM over <= 8 sequences in one persistent launch."""
    return _BiLSTM.apply(gi.contiguous(), w_hh.contiguous(), b_hh.contiguous(), lengths.contiguous())
