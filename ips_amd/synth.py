"""Deterministic synthetic configurations, weights and inputs (no datasets, no downloads).

Configurations are the reference's shipped YAMLs (/root/reference/config/*.yml)
with the overrides BASELINE.json / SURVEY.md section 8(d-2) name.  Weights and
inputs are drawn from numpy PCG64 streams keyed by (seed, tensor name), so they
are identical on every machine and independent of module construction order.
"""

import copy
import zlib

import numpy as np
import torch


class Conf:
    """Attribute bag, like the reference's utils.Struct (utils/utils.py:10-12)."""

    def __init__(self, **entries):
        self.__dict__.update(entries)

    def clone(self, **over):
        c = Conf(**copy.deepcopy(self.__dict__))
        c.__dict__.update(over)
        return c


_MNIST_TASKS = {
    'task0': {'id': 0, 'name': 'majority', 'act_fn': 'softmax', 'metric': 'accuracy'},
    'task1': {'id': 1, 'name': 'max', 'act_fn': 'softmax', 'metric': 'accuracy'},
    'task2': {'id': 2, 'name': 'top', 'act_fn': 'softmax', 'metric': 'accuracy'},
    'task3': {'id': 3, 'name': 'multi', 'act_fn': 'sigmoid', 'metric': 'multilabel_accuracy'},
}


def mnist_conf(N=2500, M=64, I=64, patch=32, **over):
    """config/mnist_config.yml with the benchmark's N / M / I / patch size."""
    c = Conf(n_class=10, B=16, B_seq=16, eager=True, eps=1e-6, seed=0,
             is_image=True, enc_type='resnet18', pretrained=False, n_chan_in=1, n_res_blocks=2,
             shuffle=False, shuffle_style='batch', n_token=4, N=N, M=M, I=I,
             patch_size=[patch, patch], patch_stride=[patch, patch],
             use_pos=True, H=8, D=128, D_k=16, D_v=16, D_inner=512, attn_dropout=0.1, dropout=0.1,
             tasks=copy.deepcopy(_MNIST_TASKS))
    return c.clone(**over)


def traffic_conf(N=192, M=16, I=32, patch=100, **over):
    """config/traffic_config.yml, pretrained=False (no network), M=16 per BASELINE configs[0]."""
    c = Conf(n_class=4, B=16, B_seq=16, eager=True, eps=1e-6, seed=0,
             is_image=True, enc_type='resnet18', pretrained=False, n_chan_in=3, n_res_blocks=4,
             shuffle=False, shuffle_style='batch', n_token=1, N=N, M=M, I=I,
             patch_size=[patch, patch], patch_stride=[patch, patch],
             use_pos=False, H=8, D=512, D_k=64, D_v=64, D_inner=2048, attn_dropout=0.1, dropout=0.1,
             tasks={'task0': {'id': 0, 'name': 'sign', 'act_fn': 'softmax', 'metric': 'accuracy'}})
    return c.clone(**over)


def camelyon_conf(N=65536, M=256, I=256, **over):
    """config/camelyon_config.yml with M = I = 256 per BASELINE configs[3]."""
    c = Conf(n_class=1, B=16, B_seq=1, eager=True, eps=1e-6, seed=0,
             is_image=False, enc_type='resnet50', pretrained=False, n_chan_in=2048,
             shuffle=False, shuffle_style='batch', n_token=1, N=N, M=M, I=I,
             use_pos=False, H=8, D=512, D_k=64, D_v=64, D_inner=2048, attn_dropout=0.1, dropout=0.1,
             tasks={'task0': {'id': 0, 'name': 'metastases', 'act_fn': 'sigmoid', 'metric': 'auc'}})
    return c.clone(**over)


def _rng(seed, name):
    return np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))


@torch.no_grad()
def fill_weights(net, seed=0, q_gain=8.0):
    """Overwrite every parameter / buffer of ``net`` from PCG64(seed, name).

    Convolution / Linear weights: N(0, 2/fan_in) (activations keep O(1) scale);
    BatchNorm: gamma ~ U(0.5,1.5), beta ~ 0.1 N, running_mean ~ 0.1 N, running_var ~
    U(0.5,1.5) (non-trivial statistics); the learned queries are scaled by ``q_gain`` so
    that attention is peaked and top-M boundary gaps sit far above fp32 noise.
    """
    sd = net.state_dict()
    for name, t in sd.items():
        g = _rng(seed, name)
        if name.endswith("num_batches_tracked"):
            continue
        shape = tuple(t.shape)
        if name.endswith("running_var"):
            v = g.uniform(0.5, 1.5, shape)
        elif name.endswith("running_mean"):
            v = 0.1 * g.standard_normal(shape)
        elif t.dim() == 1 and name.endswith("weight"):          # BN / LN gamma
            v = g.uniform(0.5, 1.5, shape)
        elif t.dim() == 1:                                       # biases, BN / LN beta
            v = 0.1 * g.standard_normal(shape)
        elif name.endswith("crs_attn.q"):
            bound = q_gain / np.sqrt(net.transf.crs_attn.D_k)
            v = g.uniform(-bound, bound, shape)
        else:
            fan_in = int(np.prod(shape[1:]))
            v = g.standard_normal(shape) * np.sqrt(2.0 / fan_in)
        t.copy_(torch.from_numpy(v.astype(np.float32)))
    return net


def make_patches(conf, B, seed=0, blank_frac=None, N=None):
    """Synthetic ``(B, N, ...)`` patch tensor of the shape ``ips()`` sees (CPU, float32).

    Images: Megapixel-MNIST-like sparsity - a patch is all-zero with probability
    ``blank_frac`` (about 93 % of 32-px patches of the real data are blank,
    data/megapixel_mnist/make_mnist.py), otherwise U[0,1) noise; traffic-like
    (n_chan_in = 3) patches are dense N(0,1).  Features: relu(N(0,1)) like
    post-ReLU ResNet-50 features (data/camelyon/camelyon_dataset.py:137-140).
    """
    N = conf.N if N is None else N
    if blank_frac is None:
        blank_frac = getattr(conf, "blank_frac", 0.93)          # a fixture may pin its own sparsity in the conf
    g = _rng(seed, "patches")
    if not conf.is_image:
        x = np.maximum(g.standard_normal((B, N, conf.n_chan_in), dtype=np.float32), 0)
        return torch.from_numpy(x)
    h, w = conf.patch_size
    if conf.n_chan_in == 1:
        x = g.random((B, N, 1, h, w), dtype=np.float32)
        keep = g.random((B, N, 1, 1, 1)) >= blank_frac
        x = x * keep.astype(np.float32)
    else:
        x = g.standard_normal((B, N, conf.n_chan_in, h, w), dtype=np.float32)
    return torch.from_numpy(np.ascontiguousarray(x))


class ListLoader:
    """A loader the loops of training/iterative.py can drive: ``len()`` and iteration over dict items."""

    def __init__(self, items):
        self.items = items

    def __len__(self):
        return len(self.items)

    def __iter__(self):
        return iter(self.items)


def make_loader(conf, n_item, seed=0):
    """``n_item`` seeded loader items of ``B_seq`` images each: {'input': (B_seq, N, ...), <task>: labels}
    with labels of the dtype the reference's datasets deliver (int64 class ids, float rows for multi-label)."""
    items = []
    for k in range(n_item):
        g = _rng(seed, "labels%d" % k)
        item = {'input': make_patches(conf, conf.B_seq, seed=seed * 1000 + k, blank_frac=0.5)}
        for task in conf.tasks.values():
            if task['metric'] == 'multilabel_accuracy':
                lab = torch.from_numpy((g.random((conf.B_seq, conf.n_class)) < 0.3).astype(np.float32))
            elif task['act_fn'] == 'sigmoid':
                lab = torch.from_numpy(g.integers(0, 2, (conf.B_seq,)).astype(np.int64))
            else:
                lab = torch.from_numpy(g.integers(0, conf.n_class, (conf.B_seq,)).astype(np.int64))
            item[task['name']] = lab
        items.append(item)
    return ListLoader(items)
