"""MobileNetV3-small feature trunk (the ``vision_model.features`` AudioEmbedding calls).

The reference takes it from torchvision (/root/reference/vicreg_audio_params.py:52-54,
``mobilenet_v3_small(pretrained=...)``); torchvision is not in this image and there is no network, so
the architecture (Howard et al. 2019, "small" table) is defined here with torchvision's module layout
-- ``features.N``, ``.block``, ``fc1/fc2`` -- so torchvision state_dicts load by key.  Weights are
random-initialised; ``pretrained=True`` only warns.  Convolutions run on MIOpen through PyTorch-ROCm
(the trunk is a caller of the hot path, not part of it).
"""
import warnings

import torch.nn as nn


def _divisible(v, d=8):
    new = max(d, int(v + d / 2) // d * d)
    return new + d if new < 0.9 * v else new


class ConvBNAct(nn.Sequential):
    def __init__(self, cin, cout, k=3, stride=1, groups=1, act=None):
        layers = [nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, groups=groups, bias=False),
                  nn.BatchNorm2d(cout, eps=0.001, momentum=0.01)]
        if act is not None:
            layers.append(act(inplace=True))
        super().__init__(*layers)


class SqueezeExcitation(nn.Module):
    def __init__(self, channels, squeeze):
        super().__init__()
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc1 = nn.Conv2d(channels, squeeze, 1)
        self.fc2 = nn.Conv2d(squeeze, channels, 1)
        self.activation = nn.ReLU()
        self.scale_activation = nn.Hardsigmoid()

    def forward(self, x):
        s = self.scale_activation(self.fc2(self.activation(self.fc1(self.avgpool(x)))))
        return s * x


class InvertedResidual(nn.Module):
    def __init__(self, cin, k, cexp, cout, use_se, hs, stride):
        super().__init__()
        act = nn.Hardswish if hs else nn.ReLU
        self.use_res = stride == 1 and cin == cout
        layers = []
        if cexp != cin:
            layers.append(ConvBNAct(cin, cexp, 1, act=act))
        layers.append(ConvBNAct(cexp, cexp, k, stride, groups=cexp, act=act))
        if use_se:
            layers.append(SqueezeExcitation(cexp, _divisible(cexp // 4)))
        layers.append(ConvBNAct(cexp, cout, 1, act=None))
        self.block = nn.Sequential(*layers)

    def forward(self, x):
        y = self.block(x)
        return x + y if self.use_res else y


# (in, kernel, expanded, out, squeeze-excite, hardswish, stride)
_SMALL = [
    (16, 3, 16, 16, True, False, 2), (16, 3, 72, 24, False, False, 2), (24, 3, 88, 24, False, False, 1),
    (24, 5, 96, 40, True, True, 2), (40, 5, 240, 40, True, True, 1), (40, 5, 240, 40, True, True, 1),
    (40, 5, 120, 48, True, True, 1), (48, 5, 144, 48, True, True, 1), (48, 5, 288, 96, True, True, 2),
    (96, 5, 576, 96, True, True, 1), (96, 5, 576, 96, True, True, 1),
]


class MobileNetV3SmallFeatures(nn.Module):
    """Only ``.features`` is used by AudioEmbedding (audioembed.py:61): [B,3,240,245] -> [B,576,8,8]."""

    def __init__(self):
        super().__init__()
        layers = [ConvBNAct(3, 16, 3, 2, act=nn.Hardswish)]
        layers += [InvertedResidual(*c) for c in _SMALL]
        layers.append(ConvBNAct(96, 576, 1, act=nn.Hardswish))
        self.features = nn.Sequential(*layers)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        return self.features(x)


def mobilenet_v3_small(pretrained=False):
    if pretrained:
        warnings.warn("pretrained MobileNetV3 weights are not available offline; using random init")
    return MobileNetV3SmallFeatures()
