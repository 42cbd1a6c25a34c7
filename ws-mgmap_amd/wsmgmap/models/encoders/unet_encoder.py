"""Frozen RGB encoder: 5-level UNet on a resnet18 trunk, pretrained for semantic
segmentation (reference: unet_encoder.py:14-111).  It is on the rollout path only.
When the trajectory cache already holds `rgb_features` it is bypassed (unet_encoder.py:65-66).

Two execution paths: the float32 parity mode runs stock PyTorch-ROCm convolutions with eval-mode
BN (what the goldens were checked against); with `compute_dtype = bf16` the whole UNet runs on this repo's NHWC
implicit-GEMM engine (bf16 storage, f32 accumulate, eval-mode BN + ReLU kernels), which is 92 % of the rollout
FLOPs.  `layer4_1x1` stays a module call in both, so the trainers' forward hook on it (dagger_trainer.py:311)
sees the same [B,512,H/32,W/32] float32 tensor.
"""
import torch
import torch.nn as nn

from ... import ops
from .map_encoder import FoldCache, convrelu
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

    engine_dtype = None   # set to torch.bfloat16 by MGMapNet when compute_dtype == "bf16"
    _fold_cache = None

    def _folded(self, conv, bn, cin_pad=None):
        """(OHWI bf16 weight, float32 bias) of conv followed by eval-mode bn, folded; cached while the parameters and
        running statistics keep their versions (the encoder is frozen, so this is computed once) and re-folded in place
        otherwise (map_encoder.FoldCache)."""
        return self._fold_cache.get(None, None, bn, cin_pad or 0, owner=conv)

    def _cbr(self, x, conv, bn, relu=True, add_to=None):
        w, b = self._folded(conv, bn, x.shape[-1])
        return ops.conv2d_infer_bf16(x, w, b, conv.stride[0], conv.padding[0], relu, add_to=add_to)

    def _block(self, x, blk):
        """BasicBlock; relu(bn2(conv2) + identity) leaves the second convolution's epilogue, written over the identity tensor
        (the block's input, or the downsampled copy of it: neither is read again — the skips are stage OUTPUTS)."""
        identity = x if blk.downsample is None else self._cbr(x, blk.downsample[0], blk.downsample[1], relu=False)
        y = self._cbr(x, blk.conv1, blk.bn1)
        return self._cbr(y, blk.conv2, blk.bn2, add_to=identity)

    def _forward_engine(self, rgb_nhwc):
        """rgb [B,H,W,3] float -> (layer4 [B,512,H/32,W/32] f32 NCHW, proj_feat [B,64,H,W] f32 NCHW).  Inside: NHWC bf16,
        every conv + eval-mode BN (+ ReLU) is ONE launch (BN folded into weight / bias, ReLU in the conv epilogue)."""
        dt = self.engine_dtype
        x = torch.nn.functional.pad(rgb_nhwc.float(), (0, 29)).to(dt).contiguous()       # 3 -> 32 channels
        cr = lambda t, seq: self._cbr(t, seq[0], seq[1])  # noqa: E731
        full = cr(cr(x, self.conv_original_size0), self.conv_original_size1)
        stem = self.base_model
        l0 = self._cbr(x, stem.conv1, stem.bn1)
        skips = [l0]
        y = ops.maxpool3x3s2(l0)
        for stage in (stem.layer1, stem.layer2, stem.layer3, stem.layer4):
            for blk in stage:
                y = self._block(y, blk)
            skips.append(y)
        layer4 = self.layer4_1x1(ops.to_nchw(skips[4], 512))                             # stock module call: hookable
        y = ops.to_nhwc(layer4.contiguous(), 512, dtype=dt)
        for skip, lateral, fuse in ((skips[3], self.layer3_1x1, self.conv_up3), (skips[2], self.layer2_1x1, self.conv_up2),
                                    (skips[1], self.layer1_1x1, self.conv_up1), (skips[0], self.layer0_1x1, self.conv_up0)):
            y = cr(ops.upsample2x_cat(y, cr(skip, lateral)), fuse)
        proj = cr(ops.upsample2x_cat(y, full), self.conv_original_size2)
        return layer4, ops.to_nchw(proj, 64)

    def forward(self, observations):
        if "rgb_features" in observations:
            return observations["rgb_features"], None
        if self.engine_dtype is not None and observations["rgb"].is_cuda and not torch.is_grad_enabled():
            if self._fold_cache is None:
                self._fold_cache = FoldCache()
            return self._forward_engine(observations["rgb"])
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
