"""DD-PPO depth backbone: the half-width GroupNorm ResNet50 + compression head that the reference instantiates from
habitat-lab v0.1.5 (`habitat_baselines.rl.ddppo.policy.resnet.resnet50` inside `resnet_policy.ResNetEncoder`; reference call
site vlnce_baselines/models/encoders/resnet_encoders.py:25-32, weights loaded at :37-50 from the DD-PPO checkpoint's
`actor_critic.net.visual_encoder.*` entries).  habitat-lab is third-party and not under /root/reference: the structure and
the state_dict keys below restate its published v0.1.5 sources (parity UNPINNED):

    input  depth [B,H,W,1] -> NCHW -> avg_pool2d(2)                                     (resnet_policy.ResNetEncoder.forward)
    backbone.conv1      Conv(1->32, k7, s2, p3, no bias) + GroupNorm(16, 32) + ReLU      keys backbone.conv1.{0,1}.*
    backbone.maxpool    MaxPool(3, s2, p1)
    backbone.layer1..4  Bottleneck x [3, 4, 6, 3], planes 32/64/128/256, expansion 4;   keys backbone.layerL.B.convs.{0,1,3,4,6,7}.*
                        convs = 1x1, GN, ReLU, 3x3(stride), GN, ReLU, 1x1, GN;                backbone.layerL.0.downsample.{0,1}.*
                        downsample = 1x1(stride) + GN on the first block of a layer
    compression         Conv(1024->128, k3, p1, no bias) + GroupNorm(1, 128) + ReLU      keys compression.{0,1}.*
    output [B,128,4,4] for 256 x 256 depth (spatial 256/2/32 = 4, channels round(2048 / 4^2) = 128)

It is frozen and on the rollout path only (0.7 GFLOP per frame against the RGB UNet's 36-48), so it runs stock PyTorch-ROCm
float32 convolutions and F.group_norm in BOTH numeric modes: bf16 storage compounds over its 53 conv + GroupNorm layers
(measured on MI355X, tools/dbg_depth.py: relative L2 error 1 % after layer1, 3.5 % after layer2, 5 % at the output with
default initialisation, 26 % with the ill-conditioned hash-filled test weights — against 2 % for the 20-layer RGB UNet),
which is a poor trade for a frozen pretrained encoder that costs 2 % of the rollout FLOPs.  The NHWC bf16 engine path
(convolutions on the implicit-GEMM engine with float32 output, csrc/wsmg_norm.hip's group-norm kernel reading it
unrounded) exists and is tested, opt-in: `engine_dtype = torch.bfloat16` (WSMG_DEPTH_ENGINE=1 makes MGMapNet set it in
bf16 mode).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import ops

_BLOCKS = (3, 4, 6, 3)
_EXPANSION = 4


def _gn(groups, ch):
    return nn.GroupNorm(groups, ch)


class Bottleneck(nn.Module):
    def __init__(self, inplanes, planes, ngroups, stride=1, with_downsample=False):
        super().__init__()
        out = planes * _EXPANSION
        self.convs = nn.Sequential(
            nn.Conv2d(inplanes, planes, 1, bias=False), _gn(ngroups, planes), nn.ReLU(True),
            nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False), _gn(ngroups, planes), nn.ReLU(True),
            nn.Conv2d(planes, out, 1, bias=False), _gn(ngroups, out),
        )
        self.downsample = None
        if with_downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, out, 1, stride=stride, bias=False), _gn(ngroups, out))
        self.relu = nn.ReLU(True)

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        return self.relu(self.convs(x) + identity)


class ResNet50GN(nn.Module):
    def __init__(self, in_channels, base_planes, ngroups):
        super().__init__()
        self.conv1 = nn.Sequential(nn.Conv2d(in_channels, base_planes, 7, stride=2, padding=3, bias=False),
                                   _gn(ngroups, base_planes), nn.ReLU(True))
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        inplanes = base_planes
        for li, (nblk, mult) in enumerate(zip(_BLOCKS, (1, 2, 4, 8)), start=1):
            planes, stride = base_planes * mult, (1 if li == 1 else 2)
            blocks = []
            for b in range(nblk):
                first = b == 0
                blocks.append(Bottleneck(inplanes, planes, ngroups, stride if first else 1,
                                         with_downsample=first and (stride != 1 or inplanes != planes * _EXPANSION)))
                inplanes = planes * _EXPANSION
            setattr(self, f"layer{li}", nn.Sequential(*blocks))
        self.final_channels = inplanes
        self.final_spatial_compress = 1.0 / 32

    def forward(self, x):
        x = self.maxpool(self.conv1(x))
        for li in range(1, 5):
            x = getattr(self, f"layer{li}")(x)
        return x


class ResNetEncoder(nn.Module):
    """habitat-lab's ResNetEncoder for a depth-only observation space (n_input_rgb = 0, no input normalisation)."""

    engine_dtype = None   # torch.bfloat16: NHWC conv engine + group-norm kernel (set by MGMapNet in bf16 mode)

    def __init__(self, depth_hw=256, depth_channels=1, baseplanes=32, ngroups=16):
        super().__init__()
        self.backbone = ResNet50GN(depth_channels, baseplanes, ngroups)
        final_spatial = int((depth_hw // 2) * self.backbone.final_spatial_compress)
        comp = int(round(2048 / (final_spatial ** 2)))
        self.compression = nn.Sequential(nn.Conv2d(self.backbone.final_channels, comp, 3, padding=1, bias=False),
                                         nn.GroupNorm(1, comp), nn.ReLU(True))
        self.output_shape = (comp, final_spatial, final_spatial)

    # -- NHWC bf16 engine path ---------------------------------------------------------------------------------------
    _wcache = None

    def _w(self, conv, cin_pad):
        """OHWI bf16 weight of a frozen conv, input channels zero-padded to the activation's; cached by parameter version,
        re-laid out in place when it changes (map_encoder.FoldCache)."""
        if self._wcache is None:
            from .map_encoder import FoldCache
            self._wcache = FoldCache()
        return self._wcache.get(conv.weight, None, None, cin_pad)[0]

    def _conv_gn(self, x, conv, gn, relu, residual=None):
        # the convolution's float32 accumulators go to the group norm unrounded: one bf16 rounding per layer, not two
        y = ops.conv2d_infer_bf16(x, self._w(conv, x.shape[-1]), None, conv.stride[0], conv.padding[0], False, out_f32=True)
        return ops.group_norm_nhwc(y, gn.weight, gn.bias, gn.num_groups, gn.eps, relu, residual)

    def _forward_engine(self, x_nchw):
        bb = self.backbone
        b, c, h, w = x_nchw.shape
        x = ops.to_nhwc(x_nchw.contiguous(), 32, dtype=torch.bfloat16)           # 1 -> 32 channels (zero padded)
        x = ops.maxpool3x3s2(self._conv_gn(x, bb.conv1[0], bb.conv1[1], True))
        for li in range(1, 5):
            for blk in getattr(bb, f"layer{li}"):
                idt = x if blk.downsample is None else self._conv_gn(x, blk.downsample[0], blk.downsample[1], False)
                y = self._conv_gn(x, blk.convs[0], blk.convs[1], True)
                y = self._conv_gn(y, blk.convs[3], blk.convs[4], True)
                x = self._conv_gn(y, blk.convs[6], blk.convs[7], True, residual=idt)     # relu(gn(conv) + identity)
        x = self._conv_gn(x, self.compression[0], self.compression[1], True)
        return ops.to_nchw(x, self.compression[0].out_channels)

    def forward(self, observations):
        x = observations["depth"].permute(0, 3, 1, 2).float()
        x = F.avg_pool2d(x, 2)
        if self.engine_dtype is not None and x.is_cuda and not torch.is_grad_enabled():
            return self._forward_engine(x)
        return self.compression(self.backbone(x))
