"""Action head: diagonal Gaussian with a learned, state-independent log-std.
Interface and state_dict keys (`fc_mean.{weight,bias}`, `logstd._bias`) follow the reference's
vlnce_baselines/common/distributions.py:42-71; `mode()`, `log_probs()` follow :21-29."""
import torch
import torch.nn as nn


class ActionNormal(torch.distributions.Normal):
    def mode(self):
        return self.mean

    def log_probs(self, actions):
        return super().log_prob(actions).sum(-1, keepdim=False)

    def entropy(self):
        return super().entropy().sum(-1)


class AddBias(nn.Module):
    def __init__(self, bias):
        super().__init__()
        self._bias = nn.Parameter(bias.unsqueeze(1))

    def forward(self, x):
        shape = (1, -1) if x.dim() == 2 else (1, -1, 1, 1)
        return x + self._bias.t().view(*shape)


class DiagGaussian(nn.Module):
    def __init__(self, num_inputs, num_outputs):
        super().__init__()
        self.fc_mean = nn.Linear(num_inputs, num_outputs)
        self.logstd = AddBias(torch.zeros(num_outputs))

    def forward(self, x):
        mean = self.fc_mean(x)
        logstd = self.logstd(torch.zeros_like(mean))
        # validate_args=False: torch.distributions' default argument check ends in `.all()` on the host, i.e. a
        # device synchronisation in every forward pass (the host then cannot queue the backward pass, or the next
        # update, while the GPU is still busy); the values are unchanged
        return ActionNormal(mean, logstd.exp(), validate_args=False)
