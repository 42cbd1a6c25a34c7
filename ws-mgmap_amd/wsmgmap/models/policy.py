"""Drop-in boundary: the policy object the reference's trainers construct and call
(vlnce_baselines/models/policy.py:15-103 — constructed at common_trainer.py:55-59, called at
dagger_trainer.py:430-439,522 and common_trainer.py:326-339).

Same constructor, `forward` / `act` / `update_map` signatures, attribute tree
(`.net`, `.action_distribution`, `.critic`, `.prog_pred`, `.prog`) and state_dict keys; the
compute behind `self.net` runs in the gfx950 kernels of libwsmgmap.so.  `CMAPolicy` is an alias
(BASELINE.json names the surface that way; the reference class is `BasePolicy`).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..common.aux_losses import AuxLosses
from ..common.distributions import DiagGaussian
from .. import debug, ops
from .mg_map_policy import MGMapNet, SEM_CLASSES


class CriticHead(nn.Module):
    """habitat-lab v0.1.5 CriticHead: Linear(input, 1), orthogonal init (state_dict: critic.fc.*)."""

    def __init__(self, input_size):
        super().__init__()
        self.fc = nn.Linear(input_size, 1)
        nn.init.orthogonal_(self.fc.weight)
        nn.init.constant_(self.fc.bias, 0)

    def forward(self, x):
        return self.fc(x)


class BasePolicy(nn.Module):
    def __init__(self, observation_space, action_space, model_config):
        super().__init__()
        self.model_config = model_config
        self.net = MGMapNet(observation_space, model_config)
        self.action_distribution = DiagGaussian(self.net.output_size, action_space.shape[0])
        self.critic = CriticHead(self.net.output_size)
        self.prog_pred = nn.Linear(model_config.STATE_ENCODER.hidden_size, 1)
        self.prog = None

    def check_status(self, sync=True):
        """Not a reference method.  Raises WsmgError if a persistent GRU / LSTM kernel of this process timed out (its outputs are
        NaN-filled then).  sync=True waits for the device first, so that a trainer calling it between `loss.backward()` and
        `optimizer.step()` can never apply an update computed from a timed-out kernel (the checks inside forward() see what had
        finished by then and can be one update late).  Costs the host its run-ahead: opt-in."""
        ops.check_rnn_status(sync=sync)

    # -- rollout -------------------------------------------------------------------
    def update_map(self, observations, masks):
        _, proj = self.net.rgb_encoder(observations)
        self.net.rgb_mapping_module(proj, observations, masks)

    def act(self, observations, rnn_hidden_states, prev_actions, masks, deterministic=False):
        features, rnn_hidden_states, pred_map = self.net(observations, rnn_hidden_states, prev_actions, masks)
        if ops.rows_route(features):
            # rollout: progress, critic, action mean, mode / sample and log-probability in one launch (≈20 otherwise); the
            # noise of a sampled action is Normal.sample()'s own draw (same generator, same shape)
            ad = self.action_distribution
            noise = None if deterministic else torch.empty(features.shape[0], ad.fc_mean.out_features, device=features.device).normal_()
            prog, value, action, logp = ops.act_heads(features, self.prog_pred, ad.fc_mean, ad.logstd._bias, self.critic.fc, noise)
            self.aux_prediction(features, observations, pred_map, prog=prog)
            return value, action, logp, rnn_hidden_states
        self.aux_prediction(features, observations, pred_map)
        distribution = self.action_distribution(features)
        value = self.critic(features)
        action = distribution.mode() if deterministic else distribution.sample()
        return value, action, distribution.log_probs(action), rnn_hidden_states

    # -- auxiliary heads -------------------------------------------------------------
    def aux_prediction(self, features, observations, pred_map, prog=None, prog_rows=None):
        """prog: the progress head's output when the caller already has it (the fused heads of act / forward); prog_rows: the
        progress monitor's per-row loss, likewise."""
        cfg = self.model_config
        self.prog = torch.tanh(self.prog_pred(features)) if prog is None else prog
        if not AuxLosses.is_active():
            return
        if cfg.PREDICTION_MONITOR.use:
            if pred_map is None and getattr(self.net, "sem_ce_rows", None) is not None:
                loss = self.net.sem_ce_rows      # computed by the fused classifier tail (ops.cls_tail), nearest resize included
            else:
                side = self.net.sem_logits_nhwc.shape[1] if pred_map is None else pred_map.shape[-1]
                target = F.interpolate(observations["gt_semantic_map"].unsqueeze(1), size=(side, side)).squeeze(1).long()
                if pred_map is None:   # per-pixel CE straight from the NHWC logits of the conv engine
                    loss = ops.cross_entropy_nhwc(self.net.sem_logits_nhwc, target, SEM_CLASSES).mean([1, 2])
                else:
                    loss = F.cross_entropy(pred_map, target, reduction="none").mean([1, 2])
            AuxLosses.register_loss("prediction_monitor", loss, cfg.PREDICTION_MONITOR.alpha)
        if cfg.CONTRASTIVE_MONITOR.use:
            size = self.net.map_encoder.output_shape[-1]
            dis = observations["gt_path"] if "gt_path" in observations.keys() else observations["waypoint_distribution"]
            att = self.net.att_map_t_m
            if dis.is_cuda and dis.dtype == torch.float32 and att.dtype == torch.float32:
                kl = ops.path_kl(dis, att, size, cfg.CONTRASTIVE_MONITOR.target_tau)      # one launch per direction
            else:
                lo, hi = torch.aminmax(dis)  # batch-global normalisation, as the reference does (dis.max(), dis.min()) in one pass
                target = F.interpolate(((hi - dis) / (hi - lo)).unsqueeze(1), size=[size, size], mode="area").squeeze(1)
                target = F.softmax(target.reshape(target.shape[0], -1) / cfg.CONTRASTIVE_MONITOR.target_tau, dim=1)
                kl = F.kl_div(torch.log(att), target, reduction="none").mean(-1)
            AuxLosses.register_loss("contrastive_monitor", kl, cfg.CONTRASTIVE_MONITOR.alpha)
        if cfg.PROGRESS_MONITOR.use:
            loss = F.mse_loss(self.prog, observations["progress"], reduction="none").mean(-1) if prog_rows is None else prog_rows
            AuxLosses.register_loss("progress_monitor", loss, cfg.PROGRESS_MONITOR.alpha)

    # -- teacher forcing / DAgger update ---------------------------------------------
    def forward(self, observations, rnn_hidden_states, prev_actions, masks, weights):
        self.net.skip_pred_map_nchw = True   # the only consumer of pred_sem_map is the loss below
        gt = observations.get("gt_semantic_map") if (AuxLosses.is_active() and self.model_config.PREDICTION_MONITOR.use) else None
        self.net._gt_semantic_map = gt if (gt is not None and gt.is_cuda and gt.dtype == torch.float32 and gt.dim() == 3) else None
        # the fused classifier tail returns logits that carry no gradient: it may only run when the loss on them is its own
        self.net._cls_tail_allowed = gt is None or self.net._gt_semantic_map is not None
        try:
            features, rnn_hidden_states, pred_map = self.net(observations, rnn_hidden_states, prev_actions, masks)
        finally:
            self.net.skip_pred_map_nchw = False
            self.net._gt_semantic_map = None
            self.net._cls_tail_allowed = False
        # = self.action_distribution(features).mean (policy.py:96-97) without building the Normal: its log-std / exp / expand
        # kernels produce nothing the update path reads (logstd gets no gradient in the reference either)
        fused = features.is_cuda and features.dtype == torch.float32 and features.shape[1] % 4 == 0
        if fused:
            # action mean, tanh progress head and the progress monitor's per-row loss in one launch per direction (≈10 each way)
            progress = observations.get("progress") if (AuxLosses.is_active() and self.model_config.PROGRESS_MONITOR.use) else None
            if progress is not None and not (progress.is_cuda and progress.dtype == torch.float32 and progress.numel() == features.shape[0]):
                progress = None
            pred, prog, prog_rows = ops.update_heads(features, self.action_distribution.fc_mean, self.prog_pred, progress)
            self.aux_prediction(features, observations, pred_map, prog=prog, prog_rows=prog_rows)
        else:
            pred = self.action_distribution.fc_mean(features)
            self.aux_prediction(features, observations, pred_map)
        aux_loss = AuxLosses.reduce((weights > 0).view(-1))
        ops.mark("heads_aux")
        return pred, aux_loss


CMAPolicy = BasePolicy
