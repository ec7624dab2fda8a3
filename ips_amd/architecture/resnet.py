"""ResNet trunk used as the IPS patch encoder.

The reference builds its patch encoder out of a torchvision ResNet
(/root/reference/architecture/ips_net.py:17-52: conv1, bn1, relu, maxpool,
layer1, layer2[, layer3, layer4], avgpool).  torchvision is not part of the
ROCm image this framework targets, so the trunk is defined here.  Attribute
names and parameter shapes follow torchvision 0.14's ``ResNet`` so that
state-dicts interchange with the reference (``encoder.4.0.conv1.weight`` ...).

Only what the IPS path needs is here: residual stages and the stem.  The HIP
encoder (ips_amd/csrc) reads the parameters of these modules directly; the
modules themselves are the autograd path (training ``forward``) and the
description of the network the kernels walk.
"""

import torch
from torch import nn


def _conv(c_in, c_out, k, stride):
    return nn.Conv2d(c_in, c_out, kernel_size=k, stride=stride, padding=k // 2, bias=False)


class BasicBlock(nn.Module):
    """Two 3x3 convolutions with an identity (or 1x1-projected) shortcut."""

    expansion = 1

    def __init__(self, c_in, width, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv(c_in, width, 3, stride)
        self.bn1 = nn.BatchNorm2d(width)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _conv(width, width, 3, 1)
        self.bn2 = nn.BatchNorm2d(width)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        shortcut = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        y += shortcut
        return self.relu(y)


class Bottleneck(nn.Module):
    """1x1 -> 3x3 (strided) -> 1x1 (x4 channels) residual block (ResNet-50)."""

    expansion = 4

    def __init__(self, c_in, width, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv(c_in, width, 1, 1)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = _conv(width, width, 3, stride)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = _conv(width, width * self.expansion, 1, 1)
        self.bn3 = nn.BatchNorm2d(width * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        shortcut = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        y += shortcut
        return self.relu(y)


class ResNetTrunk(nn.Module):
    """Stem + four residual stages + global average pool (no classifier)."""

    def __init__(self, block, depths, n_chan_in=3):
        super().__init__()
        self._c = 64
        self.conv1 = nn.Conv2d(n_chan_in, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._stage(block, 64, depths[0], 1)
        self.layer2 = self._stage(block, 128, depths[1], 2)
        self.layer3 = self._stage(block, 256, depths[2], 2)
        self.layer4 = self._stage(block, 512, depths[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        # torchvision's ResNet builds its 1000-way classifier here; the reference drops it (ips_net.py:35-50) but its
        # initialisation has consumed torch's global RNG by then, and the transformer / heads built afterwards draw
        # from the stream that follows.  Same draws, then discard - so a seeded construction gives the reference's
        # initial transf.* / output_layers.* weights.
        nn.Linear(512 * block.expansion, 1000)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def _stage(self, block, width, n_block, stride):
        c_out = width * block.expansion
        proj = None
        if stride != 1 or self._c != c_out:
            proj = nn.Sequential(
                nn.Conv2d(self._c, c_out, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(c_out),
            )
        blocks = [block(self._c, width, stride, proj)]
        self._c = c_out
        blocks += [block(c_out, width) for _ in range(n_block - 1)]
        return nn.Sequential(*blocks)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return torch.flatten(self.avgpool(x), 1)


def resnet18_trunk(n_chan_in=3):
    return ResNetTrunk(BasicBlock, (2, 2, 2, 2), n_chan_in)


def resnet50_trunk(n_chan_in=3):
    return ResNetTrunk(Bottleneck, (3, 4, 6, 3), n_chan_in)


def load_torchvision_checkpoint(trunk, path):
    """Load a torchvision ``resnet18`` / ``resnet50`` state-dict file (the ``IMAGENET1K_V1`` ``.pth`` the reference
    gets from ``weights=...``, /root/reference/architecture/ips_net.py:19-27) into ``trunk``.  The file's keys are
    torchvision's (``conv1.weight``, ``layer1.0.bn1.running_mean``, ..., ``fc.weight``); the classifier (``fc.*``)
    is dropped as the reference drops it, everything else must match key for key and shape for shape."""
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(sd, dict) and "state_dict" in sd and not any(k.startswith("conv1") for k in sd):
        sd = sd["state_dict"]
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
    sd = {k: v for k, v in sd.items() if not k.startswith("fc.")}
    own = trunk.state_dict()
    # the official IMAGENET1K_V1 files (resnet18-f37072fd.pth: 102 entries, resnet50-0676ba61.pth) predate BatchNorm's
    # ``num_batches_tracked`` buffer; BatchNorm._load_from_state_dict fills it in for such files (metadata version < 2),
    # strict=True included - so those keys may be absent
    missing = sorted(k for k in set(own) - set(sd) if not k.endswith("num_batches_tracked"))
    extra = sorted(set(sd) - set(own))
    wrong = sorted(k for k in set(own) & set(sd) if tuple(own[k].shape) != tuple(sd[k].shape))
    if missing or extra or wrong:
        raise RuntimeError("{}: not a torchvision checkpoint of this trunk (missing {}, unexpected {}, shape mismatch {})"
                           .format(path, missing[:4], extra[:4], wrong[:4]))
    trunk.load_state_dict(sd, strict=True)
    return trunk
