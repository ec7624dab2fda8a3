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


# The workloads bench.py times (name -> (configuration, images per step)); tools/gen_golden_bench.py runs the
# reference on exactly these (weights seed 7, patches seed 21) and tests/golden/bench_<name>.npz holds its selections.
BENCH_WORKLOADS = {
    "mnist": lambda: (mnist_conf(N=2500, M=64, I=64), 16),              # BASELINE configs[1] at the reference's B_seq
    "mnist3000": lambda: (mnist_conf(N=10000, M=64, I=64), 16),         # configs[2]: 3000x3000, 10,000 patches per image
    "native50": lambda: (mnist_conf(N=900, M=100, I=100, patch=50), 16),
    "traffic": lambda: (traffic_conf(N=192, M=16, I=32, patch=100), 16),
    "cam": lambda: (camelyon_conf(N=65536, M=256, I=256), 1),           # configs[3]
    # the reference's SHIPPED CAMELYON memory / chunk sizes (config/camelyon_config.yml:35-36: M = I = 5000, 10,000
    # candidates per iteration) on a slide of 38,000 tiles: 7 iterations, the last chunk ragged (3,000 rows)
    "cam_native": lambda: (camelyon_conf(N=38000, M=5000, I=5000), 1),
}


# Seed sweeps (tools/gen_golden_seeds.py -> tests/golden/seeds_<family>.npz): per family a small configuration run by
# the reference under >= 20 (weights, inputs) seeds.  Case k: weights seed 100 + k, input seed 200 + k; the learned
# queries at their DEFAULT scale (q_gain = 1: flat attention, small top-M gaps) for even k and sharpened (q_gain = 8)
# for odd k; Megapixel-MNIST families alternate stroke-like sparse images (k % 4 < 2) with U[0,1) noise patches.
SEED_FAMILIES = {
    "mnist32": lambda: (mnist_conf(N=400, M=16, I=16), 2, 24),
    "mnist50": lambda: (mnist_conf(N=144, M=12, I=12, patch=50), 1, 20),
    "traffic": lambda: (traffic_conf(N=24, M=4, I=8, patch=64), 1, 20),
    "cam": lambda: (camelyon_conf(N=2048, M=64, I=64), 1, 20),
}


def seed_case(family, k):
    """(conf, B, weight seed, q_gain, patches) of case k of a family."""
    conf, B, n = SEED_FAMILIES[family]()
    q_gain = 1.0 if k % 2 == 0 else 8.0
    if conf.is_image and conf.n_chan_in == 1 and k % 4 < 2:
        x = make_stroke_patches(conf, B, seed=200 + k)
    else:
        x = make_patches(conf, B, seed=200 + k)
    return conf, B, 100 + k, q_gain, x


def bench_workload(name):
    return BENCH_WORKLOADS[name]()


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


def _stroke_object(g, size=28):
    """One digit-like object: a thick, soft-edged polyline through 3-5 random control points of a size x size box,
    intensities in [0, 1] like an anti-aliased MNIST digit."""
    n = int(g.integers(3, 6))
    pts = g.uniform(3, size - 4, (n, 2))
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float64)
    d = np.full((size, size), 1e9)
    for a, b in zip(pts[:-1], pts[1:]):
        ab = b - a
        t = np.clip(((yy - a[0]) * ab[0] + (xx - a[1]) * ab[1]) / max(float(ab @ ab), 1e-9), 0, 1)
        d = np.minimum(d, np.hypot(yy - (a[0] + t * ab[0]), xx - (a[1] + t * ab[1])))
    return np.clip(1.7 - d / 1.1, 0, 1).astype(np.float32)


def _scribble_object(g, size=28):
    """One noise object of Megapixel-MNIST (data/megapixel_mnist/make_mnist.py:81-106 describes them: two straight
    lines through a 28 x 28 box, intensities in [0.8, 1])."""
    img = np.zeros((size, size), dtype=np.float32)
    for _ in range(2):
        ang = np.tan(g.random() * np.pi / 2.5)
        m = min(size - 0.51, (size - 0.51) / max(ang, 1e-6))
        x = np.linspace(0, m, 2 * size)
        y = ang * x
        r, c = np.round(x).astype(int), np.round(y).astype(int)
        if g.random() < 0.33:
            c = size - 1 - c
        img[r, c] = 1.0
    return img * (g.random((size, size)).astype(np.float32) * 0.2 + 0.8)


def make_stroke_patches(conf, B, seed=0, N=None):
    """Megapixel-MNIST-like images cut into patches: per image a zero canvas of (rows*h) x (cols*w) pixels with 5
    digit-like stroke objects and noise scribbles (50 per 1500 x 1500 pixels, at least 3) of 28 x 28 pixels at random
    positions - so objects straddle patch borders, most patches are blank and the others are SPARSE (thin strokes on
    zero background), unlike ``make_patches``' dense U[0,1) noise.  (B, N, 1, h, w) float32, row-major patch order
    like the reference's unfold (data/megapixel_mnist/mnist_dataset.py:44-51)."""
    N = conf.N if N is None else N
    h, w = conf.patch_size
    cols = int(np.ceil(np.sqrt(N)))
    rows = -(-N // cols)
    H, W = rows * h, cols * w
    size = min(28, H, W)
    n_noise = max(3, int(round(50 * H * W / 1500.0 ** 2)))
    g = _rng(seed, "strokes")
    out = np.zeros((B, N, 1, h, w), dtype=np.float32)
    for b in range(B):
        canvas = np.zeros((H, W), dtype=np.float32)
        for k in range(n_noise + 5):
            obj = _scribble_object(g, size) if k < n_noise else _stroke_object(g, size)
            y, x = int(g.integers(0, H - size + 1)), int(g.integers(0, W - size + 1))
            if k < n_noise:
                canvas[y:y + size, x:x + size] = obj                # the reference overwrites, digits last
            else:
                canvas[y:y + size, x:x + size] = np.where(obj > 0, obj, canvas[y:y + size, x:x + size])
        patches = canvas.reshape(rows, h, cols, w).transpose(0, 2, 1, 3).reshape(rows * cols, h, w)
        out[b, :, 0] = patches[:N]
    return torch.from_numpy(out)


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


def pos_table_record(conf):
    """What a fixture keeps of the positional table of the machine that generated it.  ``pos_enc_1d`` (reference
    transformer.py:6-18) runs on the host CPU, and its frequency vector ``exp(-2j ln(1e4) / D)`` differs in the last
    ulp between CPU models; multiplied by positions up to N that moves table entries by up to ~N * 6e-8 - enough to
    reorder near-identical blank patches (which differ by their positional encoding ONLY) at N = 10,000.  "Identical
    inputs" includes this table, so fixtures carry the frequency vector (D/2 floats) and a checksum of the table."""
    import math
    freq = torch.exp(torch.arange(0, conf.D, 2, dtype=torch.float) * -(math.log(10000.0) / conf.D))
    return {"pos_freq": freq.numpy(), "pos_abs_sum": np.float64(pos_table_from(freq, conf.N).double().abs().sum().item())}


def pos_table_from(freq, N):
    """The table of ``pos_enc_1d`` from a given frequency vector: (N, D) float32."""
    freq = torch.as_tensor(freq, dtype=torch.float32)
    phase = torch.arange(0, N).unsqueeze(1).float() * freq
    table = torch.zeros(N, 2 * freq.numel())
    table[:, 0::2] = torch.sin(phase)
    table[:, 1::2] = torch.cos(phase)
    return table


def use_fixture_pos_table(net, z):
    """Give ``net`` the positional table of the machine that recorded fixture ``z`` (see pos_table_record); returns
    True when this host's own table was already identical to it."""
    if not net.use_pos or "pos_freq" not in z.files:
        return True
    own = net.pos_enc
    table = pos_table_from(z["pos_freq"], own.shape[1]).unsqueeze(0).to(own.device)
    same = bool(torch.equal(table, own))
    net.pos_enc = table
    return same


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
