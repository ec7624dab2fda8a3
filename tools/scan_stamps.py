#!/usr/bin/env python
"""Diagnostic: per-phase cycles of the selection-loop kernel (STAMP build), summed over iterations.
    python tools/scan_stamps.py mnist|cam          # scan_fast_kernel; IPSX_SCAN_FAST=0 in the environment: scan_resident_kernel
Also times the un-instrumented kernel (HIP events) and checks that both kernels select the same indices.
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
if os.environ.get("IPSX_SCAN_FAST", "1") == "0":
    names = ["stage chunk+barrier", "row stats", "attention weights", "scores+keys", "rank", "gather winners"]
else:
    names = ["stage chunk+barrier", "row maxima", "exp (new rows)", "row sums", "weights+scores+keys", "rank", "gather winners"]
s = st.cpu().numpy()[0]
tot = s[:7].sum()
print("%s: %d iterations, L=%d, R=%d; total %d cycles = %.1f per iteration" % (kind, n_iter, M + I, H * T, tot, tot / n_iter))
for k, nme in enumerate(names):
    print("  %-22s %9.0f cycles/iter" % (nme, s[k] / n_iter))
if len(names) == 7:
    print("  chunk candidates ranked per iteration (score >= the lowest memory score): %.1f of %d" % (s[7] / n_iter, I))
    s[7] = 0

a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    idx = hip.scan(lg, M, I, H, T)
a.record()
for _ in range(10):
    idx = hip.scan(lg, M, I, H, T)
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / 10
print("  un-instrumented: %.3f ms per launch = %.2f us per iteration" % (ms, 1e3 * ms / n_iter))
