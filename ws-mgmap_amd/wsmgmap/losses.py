"""The trainer's loss of one teacher-forcing / DAgger update (vlnce_baselines/dagger_trainer.py:526-534):

    logits = torch.tanh(pred).view(T, N, -1)
    action_loss = F.mse_loss(logits, observations['waypoint'][:, :2].view(T, N, -1), reduction="none").sum(dim=2)
    action_loss = ((weights * action_loss).sum(0) / weights.sum(0)).mean()
    loss = action_loss + aux_loss

Eleven element-wise / reduction launches forward and as many backward on a [T*N, 2] tensor — dependent 3-5 us launches on the
critical path between the policy's forward and backward pass.  `dagger_loss` is the same arithmetic as ONE launch per direction
(csrc/wsmg_heads.hip).  Like `wsmgmap.optim.Adam` it is optional — the reference's lines work on the policy's outputs
unchanged — and like every operator of this package it has no CPU path: float32 CUDA tensors, or WsmgError."""
from . import ops


def dagger_loss(pred, aux_loss, waypoint, weights):
    """-> (loss, action_loss), 0-dim tensors: pred [T*N, A] (the policy's first output), aux_loss (its second: a 0-dim tensor or
    a number), waypoint [T*N, >= A] (`observations['waypoint']`: the first A columns are the target), weights [T, N]."""
    return ops.dagger_loss(pred, aux_loss, waypoint, weights)
