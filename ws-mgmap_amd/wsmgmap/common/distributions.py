"""Action head: diagonal Gaussian with a learned, state-independent log-std.
Interface and state_dict keys (`fc_mean.{weight,bias}`, `logstd._bias`) follow the reference's
vlnce_baselines/common/distributions.py:42-71; `mode()`, `log_probs()` follow :21-29."""
import torch
import torch.nn as nn


class ActionNormal(torch.distributions.Normal):
    def mode(self):
        return self.mean

    def sample(self, sample_shape=torch.Size()):
        """The same draw as torch.distributions.Normal.sample() — standard normals from the current generator, times scale,
        plus loc, in that order — without torch.normal(mean, std)'s `std.min() >= 0` check, which reads a value back to the
        host in every rollout step and cannot be captured into a HIP graph (wsmgmap.graph.GraphedAct)."""
        shape = self._extended_shape(sample_shape)
        with torch.no_grad():
            return torch.empty(shape, dtype=self.loc.dtype, device=self.loc.device).normal_(0.0, 1.0).mul_(self.scale.expand(shape)).add_(
                self.loc.expand(shape))

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
