"""The ResNet trunks under the torchvision boundary against facts torchvision publishes - independent
of this repo's own fixtures, which were generated WITH these trunks standing in for torchvision on both sides.

torchvision 0.14 (the reference's pin, requirements.txt:181): resnet18 has 11,689,512 parameters and a 122-entry
state dict; resnet50 25,557,032 parameters and 320 entries (model cards of ``ResNet18_Weights.IMAGENET1K_V1`` /
``ResNet50_Weights.IMAGENET1K_V1``).  The classifier the reference drops is Linear(512 | 2048, 1000):
513,000 | 2,049,000 parameters and 2 entries."""

import pytest
import torch

from ips_amd.architecture.resnet import resnet18_trunk, resnet50_trunk

PUBLISHED = {
    "resnet18": (resnet18_trunk, 11_689_512, 512, 122),
    "resnet50": (resnet50_trunk, 25_557_032, 2048, 320),
}


@pytest.mark.parametrize("name", sorted(PUBLISHED))
def test_parameter_and_entry_counts_equal_torchvision(name):
    ctor, n_param_tv, width, n_entry_tv = PUBLISHED[name]
    trunk = ctor()
    fc = width * 1000 + 1000
    assert sum(p.numel() for p in trunk.parameters()) == n_param_tv - fc
    assert {"resnet18": 11_176_512, "resnet50": 23_508_032}[name] == n_param_tv - fc
    assert len(trunk.state_dict()) == n_entry_tv - 2


def test_resnet18_shapes_stage_by_stage():
    """torchvision's resnet18 on a 224-px image: 64x112x112 after the stem, 64x56x56 after the pool, then
    64x56x56, 128x28x28, 256x14x14, 512x7x7 (the table of He et al. 2015, Table 1)."""
    t = resnet18_trunk().eval()
    x = torch.zeros(1, 3, 224, 224)
    with torch.no_grad():
        x = t.relu(t.bn1(t.conv1(x)))
        assert tuple(x.shape) == (1, 64, 112, 112)
        x = t.maxpool(x)
        assert tuple(x.shape) == (1, 64, 56, 56)
        for layer, shape in ((t.layer1, (64, 56, 56)), (t.layer2, (128, 28, 28)), (t.layer3, (256, 14, 14)),
                             (t.layer4, (512, 7, 7))):
            x = layer(x)
            assert tuple(x.shape[1:]) == shape
    # shortcut projections only where the shape changes; stride on the first 3x3 of a stage (BasicBlock)
    assert t.layer1[0].downsample is None and t.layer2[0].downsample is not None
    assert t.layer2[0].conv1.stride == (2, 2) and t.layer2[0].conv2.stride == (1, 1)
    assert t.layer2[0].downsample[0].kernel_size == (1, 1) and t.layer2[0].downsample[0].stride == (2, 2)


def test_resnet50_puts_the_stride_on_the_3x3():
    """torchvision's Bottleneck is 'ResNet v1.5': the stride sits on conv2 (3x3), not on conv1."""
    t = resnet50_trunk()
    b = t.layer2[0]
    assert b.conv1.stride == (1, 1) and b.conv2.stride == (2, 2) and b.conv3.stride == (1, 1)
    assert b.conv3.out_channels == 512 and t.layer4[2].conv3.out_channels == 2048
    assert t.layer1[0].downsample is not None            # 64 -> 256 channels


def test_initialisation_follows_torchvision():
    """kaiming_normal_(fan_out, relu) on convolutions, BN weight 1 / bias 0, no zero-init of the last BN."""
    torch.manual_seed(0)
    t = resnet18_trunk()
    w = t.layer3[1].conv2.weight
    assert float(w.detach().std()) == pytest.approx((2.0 / (256 * 9)) ** 0.5, rel=0.02)
    assert float(t.layer3[1].bn2.weight.min()) == 1.0 and float(t.layer3[1].bn2.bias.abs().max()) == 0.0
