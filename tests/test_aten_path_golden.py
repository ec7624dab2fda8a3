"""The product's ATen path (CPU device: host plumbing, BASELINE configs[0]) against the
reference's recorded outputs.  Same ATen ops on the same chunks as the reference
(/root/reference/architecture/ips_net.py:206-241), so indices must be identical."""

import numpy as np
import pytest
import torch

from tests.util import Golden, GOLDEN_CASES

FAST = [c for c in GOLDEN_CASES if c not in ("traffic_full", "mnist_native50", "mnist_full", "cam_small")]


@pytest.mark.parametrize("case", FAST)
def test_aten_ips_matches_reference(case):
    g = Golden(case)
    net = g.net("cpu")
    x = g.patches()
    torch.manual_seed(g.torch_seed)
    mem_patch, mem_pos = net.ips(x)
    assert np.array_equal(net.last_mem_idx.numpy(), g.mem_idx)
    with torch.no_grad():
        preds = net(mem_patch, mem_pos)
    for k, v in g.preds.items():
        assert np.abs(preds[k].numpy() - v).max() < 1e-5
    s = mem_patch.double().sum(dim=tuple(range(2, mem_patch.dim()))).numpy()
    assert np.allclose(s, g.mem_patch_sum, rtol=1e-12, atol=1e-9)


def test_traffic_cpu_plumbing():
    """BASELINE configs[0]: traffic-signs shape, CPU device, batch 1, M = 16."""
    g = Golden("traffic_full")
    net = g.net("cpu")
    mem_patch, mem_pos = net.ips(g.patches())
    assert mem_pos is None and tuple(mem_patch.shape) == (1, 16, 3, 100, 100)
    assert np.array_equal(net.last_mem_idx.numpy(), g.mem_idx)


def test_shortcut_when_memory_covers_all_patches():
    g = Golden("mnist_onechunk")
    conf = g.conf.clone(M=64)            # M >= N = 40 -> reference ips_net.py:185-188
    from ips_amd.architecture import IPSNet
    net = IPSNet(torch.device("cpu"), conf).eval()
    x = g.patches()
    mem_patch, mem_pos = net.ips(x)
    assert mem_patch.data_ptr() == x.data_ptr() or torch.equal(mem_patch, x)
    assert tuple(mem_pos.shape) == (g.B, conf.N, conf.D)
    assert net.last_mem_idx is None


def test_training_mode_is_restored_and_eval_stats_used():
    g = Golden("mnist_mini")
    net = g.net("cpu")
    net.train()
    mem_patch, _ = net.ips(g.patches())
    assert net.training and net.encoder.training and net.transf.training
    # BN running stats untouched by ips(), selection equals the eval-mode one
    assert np.array_equal(net.last_mem_idx.numpy(), g.mem_idx)


def _layout():
    import json, os
    from tests.util import GOLDEN_DIR
    return json.load(open(os.path.join(GOLDEN_DIR, "state_dict_layout.json")))


@pytest.mark.parametrize("name", ["mnist", "traffic", "camelyon"])
def test_state_dict_equals_the_reference_key_for_key(name):
    """tests/golden/state_dict_layout.json was recorded from the REFERENCE's IPSNet (tools/gen_golden_statedict.py):
    same keys in the same order, same shapes and dtypes - state-dicts interchange - and, constructed
    under the same torch seed, the same initial values: the reference builds transf.* / output_layers.* from torch's
    global RNG after the encoder, so this also pins how much of the stream the encoder construction consumes."""
    from ips_amd import synth
    from ips_amd.architecture import IPSNet
    rec = _layout()
    cfg = rec["configs"][name]
    conf = synth.Conf(**cfg["conf"])
    torch.manual_seed(rec["seed"])
    net = IPSNet(torch.device("cpu"), conf)
    sd = net.state_dict()
    assert list(sd.keys()) == [e["key"] for e in cfg["entries"]]
    assert sum(p.numel() for p in net.parameters()) == cfg["n_param"]
    for e in cfg["entries"]:
        v = sd[e["key"]]
        assert list(v.shape) == e["shape"] and str(v.dtype) == "torch." + e["dtype"], e["key"]
        if "sum" in e:
            assert float(v.double().sum()) == pytest.approx(e["sum"], rel=1e-12, abs=1e-12), e["key"]
            assert float(v.double().abs().sum()) == pytest.approx(e["abs_sum"], rel=1e-12, abs=1e-12), e["key"]


def test_pretrained_reads_a_local_torchvision_checkpoint(tmp_path, monkeypatch):
    """pretrained: True (config/traffic_config.yml:26, reference ips_net.py:19-27) reads a torchvision-format state
    dict (incl. the fc.* entries the reference discards) from IPSX_PRETRAINED_RESNET18; without it: a clear error."""
    from ips_amd import synth
    from ips_amd.architecture import IPSNet
    from ips_amd.architecture.resnet import resnet18_trunk
    conf = synth.traffic_conf(N=20, M=4, I=6, patch=64, pretrained=True)
    monkeypatch.delenv("IPSX_PRETRAINED_RESNET18", raising=False)
    monkeypatch.delenv("IPSX_PRETRAINED", raising=False)
    with pytest.raises(RuntimeError, match="IPSX_PRETRAINED_RESNET18"):
        IPSNet(torch.device("cpu"), conf)
    torch.manual_seed(3)
    tv = resnet18_trunk()
    sd = {k: v + 0.25 for k, v in tv.state_dict().items() if v.is_floating_point()}
    sd.update({k: v for k, v in tv.state_dict().items() if not v.is_floating_point()})
    sd["fc.weight"], sd["fc.bias"] = torch.zeros(1000, 512), torch.zeros(1000)
    path = str(tmp_path / "resnet18.pth")
    torch.save(sd, path)
    monkeypatch.setenv("IPSX_PRETRAINED_RESNET18", path)
    net = IPSNet(torch.device("cpu"), conf)
    assert torch.equal(net.encoder[0].weight, sd["conv1.weight"])
    assert torch.equal(net.encoder[7][1].bn2.running_var, sd["layer4.1.bn2.running_var"])
    # 1-channel input: the stem is replaced AFTER loading, like the reference (:29-31)
    net1 = IPSNet(torch.device("cpu"), synth.mnist_conf(N=64, M=8, I=8, pretrained=True))
    assert tuple(net1.encoder[0].weight.shape) == (64, 1, 7, 7)
    assert torch.equal(net1.encoder[4][0].conv1.weight, sd["layer1.0.conv1.weight"])
    # the official IMAGENET1K_V1 files predate BatchNorm's num_batches_tracked (resnet18-f37072fd.pth: 102 entries incl.
    # fc.*): such a file loads like the full one
    old = {k: v for k, v in sd.items() if not k.endswith("num_batches_tracked")}
    assert len(old) == 102
    torch.save(old, path)
    net2 = IPSNet(torch.device("cpu"), conf)
    assert torch.equal(net2.encoder[0].weight, sd["conv1.weight"])
    assert torch.equal(net2.encoder[7][1].bn2.running_var, sd["layer4.1.bn2.running_var"])
    bad = dict(sd)
    del bad["layer2.0.downsample.0.weight"]
    torch.save(bad, path)
    with pytest.raises(RuntimeError, match="not a torchvision checkpoint"):
        IPSNet(torch.device("cpu"), conf)


def test_optimizer_hook_is_installed_by_building_a_net_not_by_importing(tmp_path):
    """torch's optimizer step hook is process-global: importing ips_amd must not install it (a host process that only
    imports the package keeps its optimizers untouched); building an IPSNet does, and a step then invalidates the packed
    weights (fused optimizers do not bump tensor versions)."""
    import subprocess
    import sys
    code = ("import torch, ips_amd.hip as hip\n"
            "from torch.optim import optimizer as O\n"
            "assert len(O._global_optimizer_post_hooks) == 0 and not hip._HOOK_INSTALLED\n"
            "from ips_amd import synth\n"
            "from ips_amd.architecture import IPSNet\n"
            "net = IPSNet(torch.device('cpu'), synth.mnist_conf(N=64, M=8, I=8))\n"
            "assert len(O._global_optimizer_post_hooks) == 1 and hip._HOOK_INSTALLED\n"
            "IPSNet(torch.device('cpu'), synth.mnist_conf(N=64, M=8, I=8))\n"
            "assert len(O._global_optimizer_post_hooks) == 1\n"
            "g = hip.weights_generation()\n"
            "opt = torch.optim.SGD(net.parameters(), lr=0.1)\n"
            "opt.step()\n"
            "assert hip.weights_generation() == g + 1\n"
            "print('ok')\n")
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]
