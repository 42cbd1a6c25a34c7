"""Masked GRU state encoder with the interface and state_dict keys (`rnn.*`) of habitat-lab
v0.1.5 `RNNStateEncoder`, which the reference imports (mg_map_policy.py:9,118,147).
Semantics: the hidden state is multiplied by `masks` before a step; for a flattened
[T*N, .] sequence that is done wherever an episode restarts.  The cell is the stock
PyTorch-ROCm (MIOpen) GRU — not one of the three hand-written operators.
"""
import torch
import torch.nn as nn


class RNNStateEncoder(nn.Module):
    def __init__(self, input_size, hidden_size, num_layers=1, rnn_type="GRU"):
        super().__init__()
        if rnn_type != "GRU":
            raise ValueError("the WS-MGMap policy uses GRU state encoders")
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

    @staticmethod
    def restart_steps(masks, n):
        """Host-side list of time steps t >= 1 at which some episode restarts (one device sync;
        compute once per forward and share between the two encoders)."""
        t = masks.numel() // n
        if t <= 1:
            return []
        flags = (masks.view(t, n)[1:] == 0.0).any(dim=-1)
        return (flags.nonzero().flatten() + 1).tolist()

    def forward(self, x, hidden_states, masks, restarts=None):
        n = hidden_states.size(1)
        if x.size(0) == n:  # single step
            y, h = self.rnn(x.unsqueeze(0), hidden_states * masks.unsqueeze(0))
            return y.squeeze(0), h
        t = x.size(0) // n
        x = x.view(t, n, x.size(1))
        m = masks.view(t, n, 1)
        if restarts is None:
            restarts = self.restart_steps(masks, n)
        bounds = [0] + list(restarts) + [t]
        h = hidden_states
        outs = []
        for s, e in zip(bounds[:-1], bounds[1:]):
            y, h = self.rnn(x[s:e], h * m[s].unsqueeze(0))
            outs.append(y)
        y = outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)
        return y.reshape(t * n, -1), h
