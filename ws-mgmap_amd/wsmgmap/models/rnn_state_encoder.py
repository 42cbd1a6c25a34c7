"""Masked GRU state encoder with the interface and state_dict keys (`rnn.*`) of habitat-lab
v0.1.5 `RNNStateEncoder`, which the reference imports (mg_map_policy.py:9,118,147).
Semantics: the hidden state is multiplied by `masks` before a step; for a flattened
[T*N, .] sequence that happens wherever an episode restarts.

`forward` runs the persistent HIP kernel pair of csrc/wsmg_rnn.hip: the input
projection of all T*N rows is one GEMM, the recurrence is ONE launch per direction instead of
~30 MIOpen launches per time step, and restarts are applied in-kernel (no host sync).
`forward_stock` keeps the stock PyTorch-ROCm (MIOpen) formulation for comparison in tests.
The `nn.GRU` child is the parameter container (checkpoint keys rnn.weight_ih_l0, ...).
"""
import torch
import torch.nn as nn

from .. import ops

MAX_BATCH = 8  # batch slots of the kernel; wider batches are processed in independent column chunks


class RNNStateEncoder(nn.Module):
    def __init__(self, input_size, hidden_size, num_layers=1, rnn_type="GRU"):
        super().__init__()
        if rnn_type != "GRU" or num_layers != 1:
            raise ValueError("the WS-MGMap policy uses single-layer GRU state encoders")
        self._num_recurrent_layers = num_layers
        self.rnn = nn.GRU(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers)
        for name, p in self.rnn.named_parameters():
            if "weight" in name:
                nn.init.orthogonal_(p)
            elif "bias" in name:
                nn.init.constant_(p, 0)

    @property
    def num_recurrent_layers(self):
        return self._num_recurrent_layers

    def forward(self, x, hidden_states, masks):
        """x [T*N, in] (time-major rows) or [N, in]; hidden_states [1, N, H]; masks [T*N, 1]."""
        from ..debug import sw
        if sw.rnn_stock:      # no persistent kernel (bench.py's last fallback under a process group; MIOpen, one host read-back)
            return self.forward_stock(x, hidden_states, masks)
        r = self.rnn
        n = hidden_states.size(1)
        t = x.size(0) // n
        if ops.rows_route(x):      # rollout: the input projection of a few rows in one launch
            gi = ops.linear_rows(x, r.weight_ih_l0, r.bias_ih_l0).view(t, n, -1)
        else:
            gi = torch.addmm(r.bias_ih_l0, x, r.weight_ih_l0.t()).view(t, n, -1)
        m = masks.reshape(t, n).float()
        h0 = hidden_states[0].clone()  # the caller overwrites hidden_states in place (reference contract)
        if n <= MAX_BATCH:
            y = ops.masked_gru(gi, r.weight_hh_l0, r.bias_hh_l0, h0, m)
        else:
            y = torch.cat([ops.masked_gru(gi[:, c:c + MAX_BATCH], r.weight_hh_l0, r.bias_hh_l0, h0[c:c + MAX_BATCH],
                                          m[:, c:c + MAX_BATCH]) for c in range(0, n, MAX_BATCH)], dim=1)
        return y.reshape(t * n, -1), y[-1:]

    # -- stock formulation (MIOpen GRU, split at restarts; one host sync) -------------------------
    @staticmethod
    def restart_steps(masks, n):
        t = masks.numel() // n
        if t <= 1:
            return []
        flags = (masks.view(t, n)[1:] == 0.0).any(dim=-1)
        return (flags.nonzero().flatten() + 1).tolist()

    def forward_stock(self, x, hidden_states, masks):
        n = hidden_states.size(1)
        if x.size(0) == n:
            y, h = self.rnn(x.unsqueeze(0), hidden_states * masks.unsqueeze(0))
            return y.squeeze(0), h
        t = x.size(0) // n
        x = x.view(t, n, x.size(1))
        m = masks.view(t, n, 1)
        bounds = [0] + self.restart_steps(masks, n) + [t]
        h = hidden_states
        outs = []
        for s, e in zip(bounds[:-1], bounds[1:]):
            y, h = self.rnn(x[s:e], h * m[s].unsqueeze(0))
            outs.append(y)
        y = outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)
        return y.reshape(t * n, -1), h
