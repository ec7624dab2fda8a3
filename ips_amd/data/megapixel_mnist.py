"""Megapixel MNIST: the reference's on-disk format, read two ways, and a synthetic writer for it.

On disk (written by /root/reference/data/megapixel_mnist/make_mnist.py:242-356): ``parameters.json``
(``width``, ``height``, ...) and ``train.npy`` / ``test.npy`` - pickled object arrays with one dict per
image: ``'input'`` = ``((flat_indices,), values)`` of the non-zero pixels of the (H, W, 1) canvas and the
four labels ``'majority'``, ``'max'``, ``'top'`` (ints) and ``'multi'`` (10 floats).

``MegapixelMNIST`` has the reference's constructor and item layout (mnist_dataset.py:6-58): by default an
item is ``{'input': (N, 1, ph, pw) patches, <task>: label}`` densified and unfolded on the host, exactly
as the reference does.  With ``sparse=True`` an item carries the non-zeros instead and ``collate_sparse``
batches them into a ``SparseImages``; its ``patches()`` scatters them into the patch tensor on the GPU
(``ipsx_patchify_sparse``): ~0.4 MB per 1500x1500 image cross PCIe instead of 9 MB, the loader workers do
no densify / unfold work, and the per-patch blank flags fall out for the exact dedup of the encoder.

``write_synthetic`` produces the same files without the MNIST download (no network here): ten procedural
28x28 glyph classes instead of handwritten digits, same canvas, noise strokes, sparsity and label rules
(make_mnist.py:56-230).
"""

import json
import os

import numpy as np
import torch
from torch.utils.data import Dataset
from torch.utils.data._utils.collate import default_collate

from .. import hip


class SparseImages:
    """A batch of sparse (H, W, C) canvases: concatenated non-zeros plus per-image offsets."""

    def __init__(self, index, value, offsets, canvas):
        self.index, self.value, self.offsets, self.canvas = index, value, offsets, tuple(canvas)

    def __len__(self):
        return self.offsets.numel() - 1

    def to(self, device, non_blocking=False):
        return SparseImages(self.index.to(device, non_blocking=non_blocking), self.value.to(device, non_blocking=non_blocking),
                            self.offsets.to(device, non_blocking=non_blocking), self.canvas)

    def pin_memory(self):
        return SparseImages(self.index.pin_memory(), self.value.pin_memory(), self.offsets.pin_memory(), self.canvas)

    def patches(self, patch_size, patch_stride, flags=False):
        """(B, N, C, ph, pw) patch tensor.  On the GPU one scatter kernel; on the CPU densify + unfold (ATen)."""
        if hip.on_device(self.index):
            return hip.patchify_sparse(self.index, self.value, self.offsets, self.canvas, patch_size, patch_stride, flags)
        H, W, Cc = self.canvas
        out = []
        for i in range(len(self)):
            lo, hi = int(self.offsets[i]), int(self.offsets[i + 1])
            img = torch.zeros(H * W * Cc, dtype=torch.float32)
            img[self.index[lo:hi]] = self.value[lo:hi]
            out.append(_unfold(img.view(H, W, Cc).permute(2, 0, 1), patch_size, patch_stride))
        out = torch.stack(out)
        if flags:
            return out, (out.flatten(2) != 0).any(-1).reshape(-1).to(torch.int32)
        return out


def _unfold(img, patch_size, patch_stride):
    """(C, H, W) -> (N, C, ph, pw), patches in row-major (py, px) order (mnist_dataset.py:44-51)."""
    p = img.unfold(1, patch_size[0], patch_stride[0]).unfold(2, patch_size[1], patch_stride[1]).permute(1, 2, 0, 3, 4)
    return p.reshape(-1, *p.shape[2:])


class MegapixelMNIST(Dataset):
    """Loads the Megapixel MNIST dataset (same constructor and items as mnist_dataset.py:6-58)."""

    def __init__(self, conf, train=True, sparse=False):
        with open(os.path.join(conf.data_dir, "parameters.json")) as f:
            self.parameters = json.load(f)
        self.patch_size = conf.patch_size
        self.patch_stride = conf.patch_stride
        self.tasks = conf.tasks
        self.sparse = sparse
        self._img_shape = (self.parameters["height"], self.parameters["width"], 1)
        self._data = np.load(os.path.join(conf.data_dir, "train.npy" if train else "test.npy"), allow_pickle=True)

    def __len__(self):
        return len(self._data)

    def __getitem__(self, i):
        if i >= len(self):
            raise IndexError()
        data = self._data[i]
        where, values = data['input']
        flat = np.asarray(where[0] if isinstance(where, tuple) else where, dtype=np.int64).reshape(-1)
        values = np.asarray(values, dtype=np.float32).reshape(-1)
        n_pix = int(np.prod(self._img_shape))
        if flat.size and (flat.min() < 0 or flat.max() >= n_pix):
            raise IndexError("image %d has a pixel index outside the %r canvas" % (i, self._img_shape))
        if self.sparse:
            item = {'sparse_index': torch.from_numpy(flat), 'sparse_value': torch.from_numpy(values),
                    'sparse_canvas': self._img_shape}
        else:
            img = np.zeros(n_pix, dtype=np.float32)
            img[flat] = values
            img = torch.from_numpy(img.reshape(self._img_shape)).permute(2, 0, 1)
            item = {'input': _unfold(img, self.patch_size, self.patch_stride)}
        for task in self.tasks.values():
            item[task['name']] = data[task['name']]
        return item


def collate_sparse(items):
    """DataLoader ``collate_fn`` for ``MegapixelMNIST(sparse=True)``: labels as the default collate makes them,
    the images as one ``SparseImages`` under ``'sparse'``."""
    index = torch.cat([it['sparse_index'] for it in items])
    value = torch.cat([it['sparse_value'] for it in items])
    counts = torch.tensor([0] + [it['sparse_index'].numel() for it in items], dtype=torch.int64)
    batch = {'sparse': SparseImages(index, value, torch.cumsum(counts, 0), items[0]['sparse_canvas'])}
    for k in items[0]:
        if not k.startswith('sparse_'):
            batch[k] = default_collate([it[k] for it in items])
    return batch


# ------------------------------------------------------------------ synthetic writer
_SEGMENTS = {   # seven-segment strokes on a 28x28 cell: (y0, x0, y1, x1)
    'a': (4, 8, 4, 19), 'b': (4, 19, 13, 19), 'c': (14, 19, 23, 19), 'd': (23, 8, 23, 19),
    'e': (14, 8, 23, 8), 'f': (4, 8, 13, 8), 'g': (13, 8, 13, 19)}
_DIGIT_SEGMENTS = ['abcdef', 'bc', 'abged', 'abgcd', 'fgbc', 'afgcd', 'afgedc', 'abc', 'abcdefg', 'abfgcd']


def _glyph(digit, g):
    """One 28x28 float32 glyph of class ``digit``: its segments, 2-3 px thick, shifted by up to 2 px, intensity 0.6-1."""
    cell = np.zeros((28, 28), dtype=np.float32)
    dy, dx = g.integers(-2, 3, 2)
    thick = int(g.integers(2, 4))
    for s in _DIGIT_SEGMENTS[digit]:
        y0, x0, y1, x1 = _SEGMENTS[s]
        ys = slice(max(0, y0 + dy), min(28, max(y0, y1) + dy + thick))
        xs = slice(max(0, x0 + dx), min(28, max(x0, x1) + dx + thick))
        cell[ys, xs] = 1.0
    return cell * (0.6 + 0.4 * g.random((28, 28), dtype=np.float32))


def _noise_bank(n_noise, g):
    """Random straight-line scribbles like make_mnist.py:82-106 (two mirrored lines per pattern, intensity 0.8-1)."""
    bank = np.zeros((n_noise, 28, 28), dtype=np.float32)
    slope = np.tan(g.random(n_noise) * np.pi / 2.5)
    for i in range(n_noise):
        m = min(27.49, 27.49 / slope[i])
        x = np.linspace(0, m, 56)
        bank[i, np.round(x).astype(int), np.round(slope[i] * x).astype(int)] = 1.0
    other = bank[g.permutation(n_noise)]
    flip = g.random(n_noise) < 0.33
    other[flip] = other[flip][:, :, ::-1]
    bank = ((bank + other) > 0).astype(np.float32)
    return bank * (0.8 + 0.2 * g.random((n_noise, 28, 28), dtype=np.float32))


def synthetic_split(n, width, height, n_noise=50, seed=0):
    """``n`` images as the list of dicts ``np.save`` writes.  Per image: three glyphs of a target class and two
    of other classes at non-overlapping positions (make_mnist.py:108-196), ``n_noise`` scribbles, labels
    majority / max / top / multi-hot (make_mnist.py:198-230)."""
    if height < 84 or width < 84:
        raise ValueError("canvas must be at least 84x84")
    g = np.random.Generator(np.random.PCG64(seed))
    bank = _noise_bank(n_noise, g)
    out = []
    for _ in range(n):
        canvas = np.zeros((height, width), dtype=np.float32)
        for k in range(n_noise):
            y, x = (g.random(2) * [height - 56, width - 56] + 28).astype(int)
            canvas[y:y + 28, x:x + 28] = bank[int(g.random() * n_noise)]
        target = int(g.integers(0, 10))
        others = g.choice([d for d in range(10) if d != target], 2)
        digits = np.array([target] * 3 + list(others), dtype=np.int64)
        spots = []
        for d in digits:
            for _try in range(10000):
                pos = np.round(g.random(2) * [height - 28, width - 28]).astype(int)
                if all(abs(pos[0] - q[0]) >= 28 or abs(pos[1] - q[1]) >= 28 for q in spots):
                    break
            else:
                raise ValueError("a %dx%d canvas has no room for five non-overlapping 28x28 glyphs" % (height, width))
            spots.append(pos)
            canvas[pos[0]:pos[0] + 28, pos[1]:pos[1] + 28] = _glyph(int(d), g)
        flat = canvas.reshape(-1)                              # (H, W, 1) raveled
        where = np.where(flat != 0)
        out.append({'input': (where, flat[where]),
                    'majority': np.int64(target), 'max': np.int64(digits.max()),
                    'top': np.int64(digits[int(np.argmin([p[0] for p in spots]))]),
                    'multi': np.eye(10)[digits].sum(0).clip(0, 1)})
    return out


def write_synthetic(data_dir, n_train=64, n_test=16, width=1500, height=1500, n_noise=50, seed=0):
    """Write ``parameters.json``, ``train.npy`` and ``test.npy`` in the reference's format (make_mnist.py:317-356)."""
    os.makedirs(data_dir, exist_ok=True)
    with open(os.path.join(data_dir, "parameters.json"), "w") as f:
        json.dump({"n_train": n_train, "n_test": n_test, "width": width, "height": height, "noise": True,
                   "n_noise": n_noise, "seed": seed, "glyphs": "synthetic seven-segment (no MNIST download)"}, f, indent=4)
    for name, n, s in (("train.npy", n_train, seed + 1), ("test.npy", n_test, seed)):
        items = synthetic_split(n, width, height, n_noise, s)
        arr = np.empty(len(items), dtype=object)
        arr[:] = items
        np.save(os.path.join(data_dir, name), arr, allow_pickle=True)
    return data_dir
