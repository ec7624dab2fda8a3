#!/usr/bin/env python
"""Diagnostic: per-phase cycles of scan_resident_kernel (STAMP build), summed over iterations.
    python tools/scan_stamps.py mnist|cam
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ips_amd import hip

kind = sys.argv[1] if len(sys.argv) > 1 else "mnist"
B, N, M, I, H, T = (16, 2500, 64, 64, 8, 4) if kind == "mnist" else (1, 65536, 256, 256, 8, 1)
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
lg = (torch.randn((B, N, H * T), generator=g) * 3).to(dev)
L = hip.lib()
L.ipsx_dbg_scan_stamps.argtypes = [C.c_void_p]
st = torch.zeros((B, 8), dtype=torch.int64, device=dev)
hip.scan(lg, M, I, H, T)
L.ipsx_dbg_scan_stamps(st.data_ptr())
hip.scan(lg, M, I, H, T)
torch.cuda.synchronize()
L.ipsx_dbg_scan_stamps(None)
n_iter = -(-(N - M) // I)
names = ["stage chunk+barrier", "row stats", "attention weights", "scores+keys", "rank", "gather winners"]
s = st.cpu().numpy()[0]
print("%s: %d iterations, L=%d, R=%d; total %d cycles = %.1f per iteration" % (kind, n_iter, M + I, H * T, s.sum(), s.sum() / n_iter))
for k, nme in enumerate(names):
    print("  %-22s %9.0f cycles/iter" % (nme, s[k] / n_iter))
