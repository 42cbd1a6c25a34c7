"""Frozen RGB encoder: 5-level UNet on a resnet18 trunk, pretrained for semantic
segmentation (reference: unet_encoder.py:14-111).  It is on the rollout path only and is not
one of the three hand-written operators: stock PyTorch-ROCm convolutions, eval-mode BN.
When the trajectory cache already holds `rgb_features` it is bypassed (unet_encoder.py:65-66).
"""
import torch
import torch.nn as nn

from .map_encoder import convrelu
from .resnet18 import ResNet18


class ResNetUNet(nn.Module):
    def __init__(self, n_channel_in, n_class_out):
        super().__init__()
        self.base_model = ResNet18(n_channel_in)
        children = list(self.base_model.children())
        self.layer0 = nn.Sequential(*children[:3])
        self.layer0_1x1 = convrelu(64, 64, 1, 0)
        self.layer1 = nn.Sequential(*children[3:5])
        self.layer1_1x1 = convrelu(64, 64, 1, 0)
        self.layer2 = children[5]
        self.layer2_1x1 = convrelu(128, 128, 1, 0)
        self.layer3 = children[6]
        self.layer3_1x1 = convrelu(256, 256, 1, 0)
        self.layer4 = children[7]
        self.layer4_1x1 = convrelu(512, 512, 1, 0)
        self.upsample = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
        self.conv_up3 = convrelu(256 + 512, 512, 3, 1)
        self.conv_up2 = convrelu(128 + 512, 256, 3, 1)
        self.conv_up1 = convrelu(64 + 256, 256, 3, 1)
        self.conv_up0 = convrelu(64 + 256, 128, 3, 1)
        self.conv_original_size0 = convrelu(n_channel_in, 64, 3, 1)
        self.conv_original_size1 = convrelu(64, 64, 3, 1)
        self.conv_original_size2 = convrelu(64 + 128, 64, 3, 1)
        self.conv_last = nn.Conv2d(64, n_class_out, 1)
        self.output_shape = [512, 7, 7]

    def forward(self, observations):
        if "rgb_features" in observations:
            return observations["rgb_features"], None
        x = observations["rgb"].permute(0, 3, 1, 2)
        full = self.conv_original_size1(self.conv_original_size0(x))
        skips = []
        y = x
        for stage in (self.layer0, self.layer1, self.layer2, self.layer3, self.layer4):
            y = stage(y)
            skips.append(y)
        layer4 = self.layer4_1x1(skips[4])
        y = layer4
        for skip, lateral, fuse in ((skips[3], self.layer3_1x1, self.conv_up3), (skips[2], self.layer2_1x1, self.conv_up2),
                                    (skips[1], self.layer1_1x1, self.conv_up1), (skips[0], self.layer0_1x1, self.conv_up0)):
            y = fuse(torch.cat([self.upsample(y), lateral(skip)], dim=1))
        proj_feat = self.conv_original_size2(torch.cat([self.upsample(y), full], dim=1))
        return layer4, proj_feat


class UNet(nn.Module):
    def __init__(self, model_config):
        super().__init__()
        self.base_model = ResNetUNet(3, 27)
        ckpt = getattr(model_config.RGB_ENCODER, "pretrain_model", None)
        if ckpt:
            state = torch.load(ckpt, map_location="cpu")["models"]["img_segm_model"]
            self.base_model.load_state_dict({".".join(k.split(".")[2:]): v for k, v in state.items()})
        self.output_shape = self.base_model.output_shape

    def forward(self, observations):
        return self.base_model(observations)
