#!/usr/bin/env python
"""The two loop kernels against each other: scan_fast_kernel (LDS-resident, the shapes the reference ships) and the generic
scan_large_kernel (forced through the diagnostic switch ipsx_dbg_scan_generic) on random logits of several shapes, incl.
ragged last chunks, resumed ranges, exact ties and NaN / infinity rows: identical indices, scores and tie flags."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

# (B, N, M, I, H, T): every one is a shape scan_fast_kernel covers (R = H*T in {8, 32} with T in {1, 4})
SHAPES = [(2, 2500, 64, 64, 8, 4), (1, 8000, 256, 256, 8, 1), (3, 333, 16, 24, 8, 1), (2, 700, 64, 80, 8, 4), (2, 900, 100, 100, 8, 4),
          (2, 300, 16, 16, 8, 4), (1, 500, 64, 100, 8, 1), (2, 1500, 300, 300, 8, 1)]


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


def main():
    from ips_amd import hip
    L = hip.lib()
    L.ipsx_dbg_scan_generic.argtypes = [C.c_int]
    for s in SHAPES:
        assert L.ipsx_scan_workspace_bytes(*[s[0], s[2], s[3], s[4], s[5]]) == 0, "not a scan_fast_kernel shape: %s" % (s,)
    fast = run()
    L.ipsx_dbg_scan_generic(1)
    try:
        generic = run()
    finally:
        L.ipsx_dbg_scan_generic(0)
    for k, (a, b) in enumerate(zip(fast, generic)):
        same = torch.equal(a, b) or (a.dtype.is_floating_point and torch.equal(a.view(torch.int32), b.view(torch.int32)))
        assert same, "output %d of shape %s differs" % (k % 3, SHAPES[k // 3])
    print("scan_fast_kernel == scan_large_kernel on %d shapes (indices, scores, tie flags; resumed ranges)" % len(SHAPES))


if __name__ == "__main__":
    main()
