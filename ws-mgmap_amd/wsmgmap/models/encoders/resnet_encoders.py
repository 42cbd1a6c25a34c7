"""Depth branch: the frozen DD-PPO ResNet50 (GroupNorm, half width) + a learned 16 x 64 spatial embedding
(reference: vlnce_baselines/models/encoders/resnet_encoders.py:12-102).  When the trajectory cache already holds
`depth_features` the backbone is bypassed (resnet_encoders.py:79-80) — the update path never runs it.

The backbone is habitat-lab's (third-party, not vendored by the reference); `ddppo_resnet.py` restates it with the
state_dict keys of habitat-lab v0.1.5, so the reference's checkpoint loading (:37-50: the `visual_encoder.*` entries of
a DD-PPO checkpoint, `strict=True`) works unchanged.
"""
import torch
import torch.nn as nn

from .ddppo_resnet import ResNetEncoder


class VlnResnetDepthEncoder(nn.Module):
    def __init__(self, observation_space=None, output_size=128, checkpoint="NONE", backbone="resnet50",
                 resnet_baseplanes=32, normalize_visual_inputs=False, trainable=False, spatial_output=True,
                 visual_encoder=None):
        super().__init__()
        if visual_encoder is None:
            if backbone != "resnet50" or normalize_visual_inputs:
                raise ValueError("the WS-MGMap depth branch is the DD-PPO resnet50 without input normalisation "
                                 "(config/default.py:104-108)")
            hw, ch = 256, 1
            spaces = getattr(observation_space, "spaces", None)
            if spaces is not None and "depth" in spaces:
                hw, ch = int(spaces["depth"].shape[0]), int(spaces["depth"].shape[2])
            visual_encoder = ResNetEncoder(hw, ch, baseplanes=resnet_baseplanes, ngroups=resnet_baseplanes // 2)
        self.visual_encoder = visual_encoder
        for p in self.visual_encoder.parameters():
            p.requires_grad_(trainable)
        if checkpoint != "NONE":
            ddppo = torch.load(checkpoint, map_location="cpu")
            weights = {}
            for k, v in ddppo["state_dict"].items():
                parts = k.split(".")[2:]           # "actor_critic.net.visual_encoder.backbone..." -> from "visual_encoder"
                if parts and parts[0] == "visual_encoder":
                    weights[".".join(parts[1:])] = v
            del ddppo
            self.visual_encoder.load_state_dict(weights, strict=True)
        self.spatial_output = spatial_output
        c, h, w = self.visual_encoder.output_shape
        if not spatial_output:
            self.output_shape = (output_size,)
            self.visual_fc = nn.Sequential(nn.Flatten(), nn.Linear(c * h * w, output_size), nn.ReLU(True))
        else:
            self.spatial_embeddings = nn.Embedding(h * w, 64)
            self.output_shape = (c + self.spatial_embeddings.embedding_dim, h, w)

    def forward(self, observations):
        x = observations["depth_features"] if "depth_features" in observations else self.visual_encoder(observations)
        if not self.spatial_output:
            return self.visual_fc(x)
        b, c, h, w = x.size()
        # the reference looks up ALL rows in order (embedding(arange(h * w)), resnet_encoders.py:86-98): that is the weight itself
        # — same values, same gradient (every row is hit exactly once), no arange / gather launches and no sort-based
        # embedding backward in the update
        emb = self.spatial_embeddings
        if emb.num_embeddings != h * w:
            raise ValueError(f"spatial embedding table has {emb.num_embeddings} rows for a {h} x {w} feature map")
        spatial = emb.weight.view(1, -1, h, w).expand(b, emb.embedding_dim, h, w)
        return torch.cat([x, spatial], dim=1)
