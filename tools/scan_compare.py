#!/usr/bin/env python
"""The loop kernels against each other: scan_cam_kernel (specialised: 8 logits per candidate, M = I = 256),
scan_fast_kernel (LDS-resident, one thread per element: the shapes the reference ships; forced for the specialised
shape through the diagnostic switch ipsx_dbg_scan_r8) and the generic scan_large_kernel (forced through
ipsx_dbg_scan_generic) on random logits of several shapes, incl. ragged last chunks, resumed ranges, exact ties (a few, and
in every iteration), many survivors, NaN / infinity rows: identical indices, scores and tie flags."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

# (B, N, M, I, H, T): every one is a shape scan_fast_kernel covers (R = H*T in {8, 32} with T in {1, 4}); the ones with
# R = 8, M = I = 256 run on scan_cam_kernel by default
SHAPES = [(2, 2500, 64, 64, 8, 4), (1, 8000, 256, 256, 8, 1), (3, 333, 16, 24, 8, 1), (2, 700, 64, 80, 8, 4), (2, 900, 100, 100, 8, 4),
          (2, 300, 16, 16, 8, 4), (1, 500, 64, 100, 8, 1), (2, 1500, 300, 300, 8, 1),
          (2, 3000, 64, 64, 8, 1), (2, 5000, 128, 200, 8, 1), (1, 4000, 192, 256, 8, 1), (3, 700, 64, 48, 8, 1),
          (2, 6000, 256, 256, 8, 1), (2, 3000, 256, 256, 8, 1), (1, 2100, 256, 256, 8, 1), (2, 1000, 256, 100, 8, 1),
          (3, 1900, 256, 256, 8, 1), (2, 4196, 256, 256, 8, 1)]
R8 = [k for k, s in enumerate(SHAPES) if s[4] * s[5] == 8 and s[5] == 1 and s[2] == 256 and s[3] == 256]


def run(only=None):
    from ips_amd import hip
    dev = torch.device("cuda:0")
    out = {}
    for k, (B, N, M, I, H, T) in enumerate(SHAPES):
        if only is not None and k not in only:
            continue
        g = torch.Generator(device="cpu").manual_seed(k)
        lg = torch.randn((B, N, H * T), generator=g) * 3
        if k in (2, 11, 16):
            lg[0, 40, 3] = float("nan"); lg[1, 7, 0] = float("inf"); lg[2, 100:110, 5] = float("-inf")
        if k in (4, 9, 17):
            lg[:, ::3] = lg[:, :1]                      # exact ties
        if k == 12:
            lg = torch.round(lg * 2) / 2                # quantised logits: duplicates, ties in every iteration
        if k == 13:
            lg = lg + torch.arange(N).view(1, N, 1) * 0.01     # scores grow along the scan: most of every chunk survives
        if k == 14:
            lg[:, 1000:1400] = lg[:, 600:1000]          # a stretch of duplicated rows (exact ties between memory and chunk)
        lg = lg.to(dev)
        idx, sc = hip.scan(lg, M, I, H, T, want_scores=True)
        out[k] = [idx.cpu(), sc.cpu(), hip.scan.last_tie.cpu()]
        n_iter = -(-(N - M) // I)
        mem = torch.empty((B, M), dtype=torch.int64, device=dev)
        tie = torch.zeros((B,), dtype=torch.int32, device=dev)
        cut = max(1, n_iter // 3)
        hip.scan_range(lg, M, I, H, T, 0, cut, mem, tie)
        hip.scan_range(lg, M, I, H, T, cut, n_iter, mem, tie)
        assert torch.equal(mem, idx), "resumed ranges differ from one launch (shape %d)" % k
    return out


def same(x, y):
    return torch.equal(x, y) or (x.dtype.is_floating_point and torch.equal(x.view(torch.int32), y.view(torch.int32)))


def main():
    from ips_amd import hip
    L = hip.lib()
    L.ipsx_dbg_scan_generic.argtypes = [C.c_int]
    L.ipsx_dbg_scan_r8.argtypes = [C.c_int]
    for s in SHAPES:
        assert L.ipsx_scan_workspace_bytes(*[s[0], s[2], s[3], s[4], s[5]]) == 0, "not a scan_fast_kernel shape: %s" % (s,)
    default = run()
    L.ipsx_dbg_scan_r8(0)
    try:
        fast = run(R8)
    finally:
        L.ipsx_dbg_scan_r8(1)
    L.ipsx_dbg_scan_generic(1)
    try:
        generic = run()
    finally:
        L.ipsx_dbg_scan_generic(0)
    for k in default:
        for j in range(3):
            assert same(default[k][j], generic[k][j]), "output %d of shape %s: default kernel != scan_large_kernel" % (j, SHAPES[k])
            if k in fast:
                assert same(default[k][j], fast[k][j]), "output %d of shape %s: scan_cam_kernel != scan_fast_kernel" % (j, SHAPES[k])
    print("scan_cam_kernel == scan_fast_kernel on %d shapes, both == scan_large_kernel on %d shapes (indices, scores, tie "
          "flags; resumed ranges); tie flags raised on shapes %s"
          % (len(fast), len(default), [k for k in default if int(default[k][2].sum()) > 0]))


if __name__ == "__main__":
    main()
