"""Parameter containers with torchvision-resnet18-compatible attribute / state_dict names
(conv1, bn1, relu, maxpool, layer1..4, avgpool, fc; BasicBlock conv1,bn1,relu,conv2,bn2,
downsample).  torchvision is not a dependency: the reference only needs the layout
(map_encoder.py:75-81, unet_encoder.py:34-47), weights come from checkpoints."""
import torch.nn as nn


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        self.stride = stride
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))

    def forward(self, x):  # stock-op path (frozen RGB encoder only)
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + identity)


def _stage(inplanes, planes, stride):
    return nn.Sequential(BasicBlock(inplanes, planes, stride), BasicBlock(planes, planes, 1))


class ResNet18(nn.Module):
    def __init__(self, in_channels=3):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = _stage(64, 64, 1)
        self.layer2 = _stage(64, 128, 2)
        self.layer3 = _stage(128, 256, 2)
        self.layer4 = _stage(256, 512, 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, 1000)
