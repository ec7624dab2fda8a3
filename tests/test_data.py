"""On-disk -> patch tensor (SURVEY.md section 8 f, N-c): the dataset mirror against what the reference's dataset
class returns for the same files (tests/golden/mnist_disk, tools/gen_golden_data.py), the sparse route against
the dense one, and - on the GPU - the patchify kernels against torch's unfold, bit for bit."""

import os

import numpy as np
import pytest
import torch

from ips_amd import hip, synth
from ips_amd.data import megapixel_mnist as mm

DISK = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mnist_disk")
GEOMS = {"p32s32": ([32, 32], [32, 32]), "p50s25": ([50, 50], [25, 25])}
DEV = "cuda:0"


@pytest.mark.parametrize("tag", list(GEOMS))
def test_dataset_items_equal_the_reference(tag):
    want = np.load(os.path.join(DISK, "expected.npz"))
    ps, st = GEOMS[tag]
    conf = synth.mnist_conf(data_dir=DISK, patch_size=ps, patch_stride=st)
    for split, train in (("train", True), ("test", False)):
        dense = mm.MegapixelMNIST(conf, train=train)
        sparse = mm.MegapixelMNIST(conf, train=train, sparse=True)
        assert len(dense) == (3 if train else 2)
        batch = mm.collate_sparse([sparse[i] for i in range(len(sparse))])
        via_sparse, flags = batch['sparse'].patches(ps, st, flags=True)
        for i in range(len(dense)):
            item, key = dense[i], "%s_%s_%d" % (tag, split, i)
            assert item['input'].dtype == torch.float32 and np.array_equal(item['input'].numpy(), want[key + "_input"])
            assert torch.equal(via_sparse[i], item['input'])
            for t in conf.tasks.values():
                assert np.array_equal(np.asarray(item[t['name']]), want[key + "_" + t['name']])
                assert np.array_equal(batch[t['name']][i].numpy(), want[key + "_" + t['name']])
        n = via_sparse.shape[1]
        assert torch.equal(flags.view(-1, n).bool(), (via_sparse.flatten(2) != 0).any(-1))
    with pytest.raises(IndexError):
        dense[len(dense)]


def test_synthetic_writer_label_rules(tmp_path):
    mm.write_synthetic(str(tmp_path), n_train=5, n_test=1, width=300, height=260, n_noise=8, seed=9)
    conf = synth.mnist_conf(data_dir=str(tmp_path))
    ds = mm.MegapixelMNIST(conf, train=True)
    assert ds.parameters["width"] == 300 and len(ds) == 5
    for i in range(len(ds)):
        item = ds[i]
        multi = np.asarray(item['multi'])
        present = set(np.nonzero(multi)[0].tolist())
        assert item['input'].shape == (8 * 9, 1, 32, 32) and multi.shape == (10,)
        assert int(item['majority']) in present and int(item['max']) == max(present) and int(item['top']) in present
        assert 2 <= len(present) <= 3
        assert float(item['input'].max()) <= 1.0 and float(item['input'].min()) >= 0.0


def test_out_of_canvas_index_is_an_error(tmp_path):
    mm.write_synthetic(str(tmp_path), n_train=1, n_test=1, width=128, height=128, n_noise=1, seed=1)
    arr = np.load(os.path.join(str(tmp_path), "train.npy"), allow_pickle=True)
    where, values = arr[0]['input']
    arr[0]['input'] = ((np.append(where[0], 128 * 128),), np.append(values, np.float32(1)))
    np.save(os.path.join(str(tmp_path), "train.npy"), arr, allow_pickle=True)
    conf = synth.mnist_conf(data_dir=str(tmp_path))
    for sparse in (False, True):
        with pytest.raises(IndexError):
            mm.MegapixelMNIST(conf, train=True, sparse=sparse)[0]


@pytest.mark.gpu
@pytest.mark.parametrize("shape,ps,st", [((2, 1, 150, 200), (32, 32), (32, 32)), ((3, 3, 100, 140), (50, 50), (25, 25)),
                                          ((1, 3, 67, 45), (10, 7), (3, 5)), ((2, 1, 64, 64), (64, 64), (64, 64))])
def test_patchify_kernel_equals_unfold(shape, ps, st):
    g = torch.Generator().manual_seed(1)
    img = torch.randn(shape, generator=g)
    want = torch.stack([mm._unfold(im, ps, st) for im in img])
    got = hip.patchify(img.to(DEV), ps, st)
    assert got.shape == want.shape and torch.equal(got.cpu(), want)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(GEOMS))
def test_sparse_patchify_kernel_equals_the_reference_items(tag):
    want = np.load(os.path.join(DISK, "expected.npz"))
    ps, st = GEOMS[tag]
    conf = synth.mnist_conf(data_dir=DISK, patch_size=ps, patch_stride=st)
    ds = mm.MegapixelMNIST(conf, train=True, sparse=True)
    batch = mm.collate_sparse([ds[i] for i in range(len(ds))])
    patches, flags = batch['sparse'].to(DEV).patches(ps, st, flags=True)
    for i in range(len(ds)):
        assert np.array_equal(patches[i].cpu().numpy(), want["%s_train_%d_input" % (tag, i)])
    assert torch.equal(flags.view(len(ds), -1).bool(), (patches.flatten(2) != 0).any(-1))


@pytest.mark.gpu
def test_flags_from_patchify_drive_the_exact_dedup():
    """1500x1500 synthetic images -> sparse -> device patches + flags -> encoder with the flags == plain encoder."""
    import tempfile
    from ips_amd.architecture.ips_net import IPSNet
    d = tempfile.mkdtemp()
    mm.write_synthetic(d, n_train=2, n_test=1, width=1500, height=1500, seed=2)
    conf = synth.mnist_conf(N=46 * 46, data_dir=d)
    ds = mm.MegapixelMNIST(conf, train=True, sparse=True)
    batch = mm.collate_sparse([ds[0], ds[1]])
    patches, flags = batch['sparse'].to(DEV).patches(conf.patch_size, conf.patch_stride, flags=True)
    assert patches.shape == (2, 46 * 46, 1, 32, 32) and 0.02 < flags.float().mean().item() < 0.2
    net = IPSNet(torch.device(DEV), conf).to(DEV).eval()
    synth.fill_weights(net, 3)
    plan = hip.EncoderPlan(net.encoder, True)
    flat = patches.view(-1, 1, 32, 32)
    plain = plan.encode(flat)
    dedup = plan.encode(flat, nonblank=flags)
    assert torch.equal(plain, dedup)
    assert int(plan.n_encoded.item()) == int(flags.sum().item()) + 1
