"""Patch-axis permutations that feed ``IPSNet.ips`` (host-side, stays on ATen).

Mirrors ``shuffle_batch`` / ``shuffle_instance`` of
/root/reference/utils/utils.py:33-58.  They consume torch's RNG streams
(``torch.randperm`` on the CPU generator; ``torch.rand`` on the generator of the
tensor's device), so for a given ``torch.manual_seed`` the permutation is the
one the reference draws - which is why this stays Python/ATen rather than a
kernel with its own generator.
"""

import torch


def shuffle_batch(x, shuffle_idx=None):
    """Same permutation of axis 1 for every instance of the batch."""
    if not torch.is_tensor(shuffle_idx):
        shuffle_idx = torch.randperm(x.shape[1])
    return x[:, shuffle_idx], shuffle_idx


def shuffle_instance(x, axis, shuffle_idx=None):
    """Independent permutation of ``axis`` per leading index (argsort of uniforms)."""
    if not torch.is_tensor(shuffle_idx):
        shuffle_idx = torch.rand(x.shape[:axis + 1], device=x.device).argsort(axis)
    take = shuffle_idx.to(x.device)
    take = take.reshape(*take.shape, *(1,) * (x.ndim - axis - 1)).expand(*x.shape[:axis + 1], *x.shape[axis + 1:])
    return x.gather(axis, take), shuffle_idx
