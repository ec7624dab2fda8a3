#!/usr/bin/env python
"""scan_fast_kernel against scan_resident_kernel (IPSX_SCAN_FAST=0 in a child process) on random logits of several
shapes, incl. ragged last chunks, resumed ranges and NaN / infinity rows: identical indices, scores and tie flags."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

SHAPES = [(2, 2500, 64, 64, 8, 4), (1, 8000, 256, 256, 8, 1), (3, 333, 16, 24, 8, 1), (2, 700, 100, 100, 8, 4),
          (2, 300, 16, 16, 4, 2), (1, 500, 8, 100, 2, 1), (2, 260, 16, 32, 16, 4)]


def run():
    from ips_amd import hip
    dev = torch.device("cuda:0")
    out = []
    for k, (B, N, M, I, H, T) in enumerate(SHAPES):
        g = torch.Generator(device="cpu").manual_seed(k)
        lg = torch.randn((B, N, H * T), generator=g) * 3
        if k == 2:
            lg[0, 40, 3] = float("nan"); lg[1, 7, 0] = float("inf"); lg[2, 100:110, 5] = float("-inf")
        if k == 4:
            lg[:, ::3] = lg[:, :1]                      # exact ties
        lg = lg.to(dev)
        idx, sc = hip.scan(lg, M, I, H, T, want_scores=True)
        out += [idx.cpu(), sc.cpu(), hip.scan.last_tie.cpu()]
        n_iter = -(-(N - M) // I)
        mem = torch.empty((B, M), dtype=torch.int64, device=dev)
        tie = torch.zeros((B,), dtype=torch.int32, device=dev)
        cut = max(1, n_iter // 3)
        hip.scan_range(lg, M, I, H, T, 0, cut, mem, tie)
        hip.scan_range(lg, M, I, H, T, cut, n_iter, mem, tie)
        assert torch.equal(mem, idx), "resumed ranges differ from one launch (shape %d)" % k
    return out


if len(sys.argv) > 1:
    torch.save(run(), sys.argv[1])
else:
    env = dict(os.environ, IPSX_SCAN_FAST="0")
    subprocess.check_call([sys.executable, __file__, "/tmp/scan_old.pt"], env=env)
    new, old = run(), torch.load("/tmp/scan_old.pt")
    for k, (a, b) in enumerate(zip(new, old)):
        same = torch.equal(a, b) or (a.dtype.is_floating_point and torch.equal(a.view(torch.int32), b.view(torch.int32)))
        assert same, "output %d of shape %s differs" % (k % 3, SHAPES[k // 3])
    print("scan_fast_kernel == scan_resident_kernel on %d shapes (indices, scores, tie flags; resumed ranges)" % len(SHAPES))
