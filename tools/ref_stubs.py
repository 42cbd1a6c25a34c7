"""sys.modules stubs that let the UNMODIFIED reference (/root/reference) import on a
CPU-only box without its un-vendored dependencies.  Used ONLY by tools/make_goldens.py
in the build container; never shipped on a product path and never imported by tests.

What is stubbed:
  * type-only: gym.Space / gym.spaces.Dict / gym.spaces.Box, habitat.Config
  * arithmetic restated from the pinned third-party versions (parity UNPINNED at these
    boundaries — the reference has no tests that pin them):
      - torch_scatter 2.0.6 `scatter_max`   (call site rgb_mapping.py:220-225)
      - torchvision `resnet18`              (call sites map_encoder.py:75, unet_encoder.py:34)
      - habitat-lab v0.1.5 `RNNStateEncoder`, `Net`, `CriticHead`, `Flatten`,
        `ResNetEncoder` + `resnet.resnet50` (the GroupNorm DD-PPO depth backbone)
"""
import sys
import types

import torch
import torch.nn as nn

CAPTURE = {}  # filled by the scatter_max stub: last index / src seen


def _mod(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


# ----------------------------------------------------------------------------- gym / habitat types
class _Space:
    pass


class _Box(_Space):
    def __init__(self, low=0, high=1, shape=(2,), dtype=None):
        self.shape = tuple(shape)


class _DictSpace(_Space):
    def __init__(self, spaces=None):
        self.spaces = dict(spaces or {})


class Config(dict):
    """attribute-style dict standing in for yacs/habitat Config."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


# ----------------------------------------------------------------------------- torch_scatter
def scatter_max(src, index, dim=-1, out=None, dim_size=None):
    """torch_scatter 2.0.6 semantics relied on by the reference: per-index max along
    `dim`; positions that receive no source are 0; returns (values, argmax)."""
    CAPTURE["index"] = index.detach().clone()
    CAPTURE["src"] = src.detach().clone()
    dim = dim % src.dim()
    shape = list(src.shape)
    shape[dim] = int(dim_size)
    res = torch.zeros(shape, dtype=src.dtype)
    res.scatter_reduce_(dim, index.contiguous(), src, reduce="amax", include_self=False)
    return res, None


# ----------------------------------------------------------------------------- torchvision resnet18
class _BasicBlock(nn.Module):
    def __init__(self, inp, out, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inp, out, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(out)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(out, out, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(out)
        self.downsample = None
        if stride != 1 or inp != out:
            self.downsample = nn.Sequential(nn.Conv2d(inp, out, 1, stride, bias=False), nn.BatchNorm2d(out))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + idt)


class _ResNet18(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = nn.Sequential(_BasicBlock(64, 64), _BasicBlock(64, 64))
        self.layer2 = nn.Sequential(_BasicBlock(64, 128, 2), _BasicBlock(128, 128))
        self.layer3 = nn.Sequential(_BasicBlock(128, 256, 2), _BasicBlock(256, 256))
        self.layer4 = nn.Sequential(_BasicBlock(256, 512, 2), _BasicBlock(512, 512))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, 1000)


def resnet18(pretrained=False, **kw):
    return _ResNet18()


# ----------------------------------------------------------------------------- habitat_baselines
class Net(nn.Module):
    pass


class CriticHead(nn.Module):
    def __init__(self, input_size):
        super().__init__()
        self.fc = nn.Linear(input_size, 1)

    def forward(self, x):
        return self.fc(x)


class Flatten(nn.Module):
    def forward(self, x):
        return x.reshape(x.size(0), -1)


class RNNStateEncoder(nn.Module):
    """habitat-lab v0.1.5 rl/models/rnn_state_encoder.py restated: hidden state is
    multiplied by `masks` before the step (single step) or at each segment start
    (sequence split wherever any mask is 0)."""

    def __init__(self, input_size, hidden_size, num_layers=1, rnn_type="GRU"):
        super().__init__()
        assert rnn_type == "GRU"
        self._num_recurrent_layers = num_layers
        self.rnn = nn.GRU(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers)

    @property
    def num_recurrent_layers(self):
        return self._num_recurrent_layers

    def forward(self, x, hidden_states, masks):
        n = hidden_states.size(1)
        if x.size(0) == n:
            y, h = self.rnn(x.unsqueeze(0), hidden_states * masks.unsqueeze(0))
            return y.squeeze(0), h
        t = x.size(0) // n
        x = x.view(t, n, x.size(1))
        masks = masks.view(t, n)
        zeros = (masks[1:] == 0.0).any(dim=-1).nonzero().flatten().tolist()
        bounds = [0] + [z + 1 for z in zeros] + [t]
        outs = []
        h = hidden_states
        for s, e in zip(bounds[:-1], bounds[1:]):
            y, h = self.rnn(x[s:e], h * masks[s].view(1, -1, 1))
            outs.append(y)
        return torch.cat(outs, 0).view(t * n, -1), h


# habitat-lab v0.1.5 rl/ddppo/policy/resnet.py + resnet_policy.py restated (third-party, not under /root/reference:
# parity UNPINNED): GroupNorm ResNet, Bottleneck expansion 4, conv1 = Sequential(conv7x7 s2, GN, ReLU), layers [3,4,6,3],
# ResNetEncoder = avg_pool2d(2) -> backbone -> compression(conv3x3 -> GroupNorm(1, C) -> ReLU).
def _conv3x3(i, o, stride=1):
    return nn.Conv2d(i, o, kernel_size=3, stride=stride, padding=1, bias=False)


def _conv1x1(i, o, stride=1):
    return nn.Conv2d(i, o, kernel_size=1, stride=stride, bias=False)


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, ngroups, stride=1, downsample=None):
        super().__init__()
        self.convs = nn.Sequential(
            _conv1x1(inplanes, planes), nn.GroupNorm(ngroups, planes), nn.ReLU(True),
            _conv3x3(planes, planes, stride), nn.GroupNorm(ngroups, planes), nn.ReLU(True),
            _conv1x1(planes, planes * self.expansion), nn.GroupNorm(ngroups, planes * self.expansion))
        self.relu = nn.ReLU(True)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.convs(x)
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class _GNResNet(nn.Module):
    def __init__(self, in_channels, base_planes, ngroups, block, layers):
        super().__init__()
        self.conv1 = nn.Sequential(nn.Conv2d(in_channels, base_planes, kernel_size=7, stride=2, padding=3, bias=False),
                                   nn.GroupNorm(ngroups, base_planes), nn.ReLU(True))
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.inplanes = base_planes
        self.layer1 = self._make_layer(block, ngroups, base_planes, layers[0])
        self.layer2 = self._make_layer(block, ngroups, base_planes * 2, layers[1], stride=2)
        self.layer3 = self._make_layer(block, ngroups, base_planes * 4, layers[2], stride=2)
        self.layer4 = self._make_layer(block, ngroups, base_planes * 8, layers[3], stride=2)
        self.final_channels = self.inplanes
        self.final_spatial_compress = 1.0 / (2 ** 5)

    def _make_layer(self, block, ngroups, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(_conv1x1(self.inplanes, planes * block.expansion, stride),
                                       nn.GroupNorm(ngroups, planes * block.expansion))
        layers = [block(self.inplanes, planes, ngroups, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, ngroups))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.conv1(x))
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))


def resnet50(in_channels, base_planes, ngroups):
    return _GNResNet(in_channels, base_planes, ngroups, _Bottleneck, [3, 4, 6, 3])


class ResNetEncoder(nn.Module):
    def __init__(self, observation_space, baseplanes=32, ngroups=32, spatial_size=128, make_backbone=None,
                 normalize_visual_inputs=False, obs_transform=None):
        super().__init__()
        assert not normalize_visual_inputs
        self._n_input_rgb = 0
        self._n_input_depth = observation_space.spaces["depth"].shape[2]
        spatial_size = observation_space.spaces["depth"].shape[0] // 2
        self.running_mean_and_var = nn.Sequential()
        self.backbone = make_backbone(self._n_input_depth, baseplanes, ngroups)
        final_spatial = int(spatial_size * self.backbone.final_spatial_compress)
        after_compression_flat_size = 2048
        num_compression_channels = int(round(after_compression_flat_size / (final_spatial ** 2)))
        self.compression = nn.Sequential(
            nn.Conv2d(self.backbone.final_channels, num_compression_channels, kernel_size=3, padding=1, bias=False),
            nn.GroupNorm(1, num_compression_channels), nn.ReLU(True))
        self.output_shape = (num_compression_channels, final_spatial, final_spatial)

    def forward(self, observations):
        depth_observations = observations["depth"].permute(0, 3, 1, 2)
        x = torch.nn.functional.avg_pool2d(depth_observations, 2)
        x = self.running_mean_and_var(x)
        return self.compression(self.backbone(x))


def install():
    gym = _mod("gym")
    gym.Space = _Space
    spaces = _mod("gym.spaces")
    spaces.Dict = _DictSpace
    spaces.Box = _Box
    gym.spaces = spaces

    hab = _mod("habitat")
    hab.Config = Config

    ts = _mod("torch_scatter")
    ts.scatter_max = scatter_max

    tv = _mod("torchvision")
    tvm = _mod("torchvision.models")
    tvm.resnet18 = resnet18
    tv.models = tvm

    hb = _mod("habitat_baselines")
    for name in ["rl", "rl.models", "rl.ppo", "rl.ddppo", "rl.ddppo.policy", "common"]:
        _mod("habitat_baselines." + name)
    m = _mod("habitat_baselines.rl.models.rnn_state_encoder")
    m.RNNStateEncoder = RNNStateEncoder
    m = _mod("habitat_baselines.rl.ppo.policy")
    m.Net = Net
    m.CriticHead = CriticHead
    m = _mod("habitat_baselines.rl.ddppo.policy.resnet")
    m.resnet50 = resnet50
    sys.modules["habitat_baselines.rl.ddppo.policy"].resnet = m
    m = _mod("habitat_baselines.rl.ddppo.policy.resnet_policy")
    m.ResNetEncoder = ResNetEncoder
    m = _mod("habitat_baselines.common.utils")
    m.Flatten = Flatten

    # empty package shells so vlnce_baselines/__init__.py (imports the trainers -> lmdb,
    # habitat sim) is bypassed; sub-modules then load from /root/reference unmodified.
    import os
    ref = "/root/reference"
    for pkg in ["vlnce_baselines", "vlnce_baselines.common", "vlnce_baselines.models",
                "vlnce_baselines.models.encoders"]:
        p = _mod(pkg)
        p.__path__ = [os.path.join(ref, *pkg.split("."))]
    return Config
