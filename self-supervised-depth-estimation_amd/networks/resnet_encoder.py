"""ResnetEncoder with the reference's constructor / forward / state_dict layout
(reference networks/resnet_encoder.py:17-98), without torchvision.

The reference delegates the trunk to `torchvision.models.resnet*`; torchvision is not a
dependency here, so the ResNet v1.5 trunk is stated directly (same module names, hence the same
state_dict keys: `encoder.conv1.weight`, `encoder.layer1.0.bn1.running_mean`, ..., `encoder.fc.*`).
Convolutions go through `conv_impl` so that the MFMA implicit-GEMM kernels can be swapped in
without touching the module tree.
"""
import numpy as np
import torch
import torch.nn as nn


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + idt)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)   # v1.5: stride on the 3x3
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + idt)


_CFG = {18: (BasicBlock, (2, 2, 2, 2)), 34: (BasicBlock, (3, 4, 6, 3)), 50: (Bottleneck, (3, 4, 6, 3)),
        101: (Bottleneck, (3, 4, 23, 3)), 152: (Bottleneck, (3, 8, 36, 3))}


class ResNetTrunk(nn.Module):
    """Module names follow torchvision.models.ResNet (conv1, bn1, relu, maxpool, layer1-4, fc)."""

    def __init__(self, num_layers, num_input_images=1):
        super().__init__()
        block, layers = _CFG[num_layers]
        self.inplanes = 64
        self.conv1 = nn.Conv2d(num_input_images * 3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], 2)
        self.layer3 = self._make_layer(block, 256, layers[2], 2)
        self.layer4 = self._make_layer(block, 512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, 1000)      # kept for checkpoint compatibility; never used
        for m in self.modules():                               # networks/resnet_encoder.py:34-39
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        down = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                 nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, down)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)


class ResnetEncoder(nn.Module):
    """networks/resnet_encoder.py:62-98.  `pretrained=True` needs a network download and is refused."""

    def __init__(self, num_layers, pretrained, num_input_images=1):
        super().__init__()
        if num_layers not in _CFG:
            raise ValueError("{} is not a valid number of resnet layers".format(num_layers))
        if pretrained:
            raise RuntimeError("pretrained ImageNet weights need a download; load a checkpoint with load_state_dict")
        self.num_ch_enc = np.array([64, 64, 128, 256, 512])
        self.encoder = ResNetTrunk(num_layers, num_input_images)
        if num_layers > 34:
            self.num_ch_enc[1:] *= 4

    def forward(self, input_image):
        e = self.encoder
        self.features = []
        x = (input_image - 0.45) / 0.225
        x = e.relu(e.bn1(e.conv1(x)))
        self.features.append(x)
        self.features.append(e.layer1(e.maxpool(x)))
        self.features.append(e.layer2(self.features[-1]))
        self.features.append(e.layer3(self.features[-1]))
        self.features.append(e.layer4(self.features[-1]))
        return self.features
