"""Shared helpers of the test-suite: golden fixtures, deterministic nets and inputs."""

import json
import os

import numpy as np
import torch

from ips_amd import synth
from ips_amd.architecture import IPSNet

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

GOLDEN_CASES = sorted(f[:-4] for f in os.listdir(GOLDEN_DIR)
                      if f.endswith(".npz") and not f.startswith(("loop_", "bench_", "seeds_", "margin")))
# what the CPU oracle replays in seconds (the rest is checked on the GPU only)
ORACLE_FAST_CASES = [c for c in GOLDEN_CASES if c not in ("traffic_full", "mnist_native50")]


class Golden:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.name = name
        self.conf = synth.Conf(**json.loads(str(z["conf"])))
        self.B = int(z["B"])
        self.weight_seed, self.patch_seed, self.torch_seed = (int(z[k]) for k in
                                                              ("weight_seed", "patch_seed", "torch_seed"))
        self.trace_idx = z["trace_idx"]          # (B, n_iter, M)
        self.trace_score = z["trace_score"]
        self.mem_idx = self.trace_idx[:, -1]
        self.min_rel_gap = float(z["min_rel_gap"])
        self.emb_head = z["emb_head"]
        self.state_checksum = float(z["state_checksum"])
        self.perm = z["perm"] if "perm" in z.files else None
        self.preds = {k[5:]: z[k] for k in z.files if k.startswith("pred_")}
        self.mem_patch_sum = z["mem_patch_sum"]
        self.mem_pos_sum = z["mem_pos_sum"] if "mem_pos_sum" in z.files else None

    def net(self, device="cpu"):
        net = IPSNet(torch.device(device), self.conf)
        synth.fill_weights(net, self.weight_seed)
        chk = float(sum(v.double().abs().sum().item() for k, v in net.state_dict().items()
                        if not k.endswith("num_batches_tracked")))
        assert abs(chk - self.state_checksum) <= 1e-9 * abs(self.state_checksum), "weights differ from fixture"
        return net.to(device).eval()

    def patches(self):
        return synth.make_patches(self.conf, self.B, seed=self.patch_seed)

    def shuffled(self, x):
        """patches in the order ips() sees them (applies the fixture's permutation)."""
        if self.perm is None:
            return x
        take = torch.from_numpy(self.perm).view(x.shape[0], -1, *(1,) * (x.dim() - 2)).expand_as(x)
        return torch.gather(x, 1, take)


def max_rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def ulp_diff(a, b):
    """max distance in units in the last place between two float32 arrays"""
    a = np.ascontiguousarray(a, dtype=np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, dtype=np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7FFFFFFF), a)
    b = np.where(b < 0, -(b & 0x7FFFFFFF), b)
    return int(np.abs(a - b).max()) if a.size else 0
