"""Depth branch: learned 16 x 64 spatial embedding concatenated to the DD-PPO ResNet50
features (reference: resnet_encoders.py:12-102).  The ResNet50 itself is third-party
(habitat-lab `ResNetEncoder`, not vendored in the reference): callers supply
`observations['depth_features']` (what the trajectory cache stores and what the reference's
forward short-circuits on, resnet_encoders.py:79-80) or attach their own `visual_encoder`.
"""
import torch
import torch.nn as nn


class _ExternalDepthBackbone(nn.Module):
    """Placeholder for habitat-lab's DD-PPO ResNet50 (hookable, parameter-free)."""

    output_shape = (128, 4, 4)

    def forward(self, observations):
        raise RuntimeError(
            "the DD-PPO depth ResNet50 is a third-party habitat-lab module: pass observations['depth_features'] "
            "[B,128,4,4] or assign policy.net.depth_encoder.visual_encoder")


class VlnResnetDepthEncoder(nn.Module):
    def __init__(self, observation_space=None, output_size=128, checkpoint="NONE", backbone="resnet50",
                 resnet_baseplanes=32, normalize_visual_inputs=False, trainable=False, spatial_output=True,
                 visual_encoder=None):
        super().__init__()
        self.visual_encoder = visual_encoder if visual_encoder is not None else _ExternalDepthBackbone()
        for p in self.visual_encoder.parameters():
            p.requires_grad_(trainable)
        self.spatial_output = spatial_output
        c, h, w = self.visual_encoder.output_shape
        self.spatial_embeddings = nn.Embedding(h * w, 64)
        self.output_shape = (c + self.spatial_embeddings.embedding_dim, h, w)

    def forward(self, observations):
        x = observations["depth_features"] if "depth_features" in observations else self.visual_encoder(observations)
        b, c, h, w = x.size()
        idx = torch.arange(0, self.spatial_embeddings.num_embeddings, device=x.device, dtype=torch.long)
        spatial = self.spatial_embeddings(idx).view(1, -1, h, w).expand(b, self.spatial_embeddings.embedding_dim, h, w)
        return torch.cat([x, spatial], dim=1)
