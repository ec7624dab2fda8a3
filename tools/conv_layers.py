#!/usr/bin/env python
"""Per-layer efficiency of conv_nhwc_kernel at the shapes of the layered trunks (channels-last activations).

    python tools/conv_layers.py [n_patches] [train]
"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ips_amd import hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
dev = torch.device("cuda:0")
L = hip.lib()
# (c_in, c_out, h, k, stride): ResNet-18 on 100x100 patches (after the stem + pool: 25x25), and 50x50 patches (13x13)
shapes = [(64, 64, 25, 3, 1), (64, 128, 25, 3, 2), (128, 128, 13, 3, 1), (128, 256, 13, 3, 2), (256, 256, 7, 3, 1),
          (256, 512, 7, 3, 2), (512, 512, 4, 3, 1), (64, 128, 25, 1, 2), (64, 64, 13, 3, 1), (64, 128, 13, 3, 2),
          (128, 128, 7, 3, 1)]
if len(sys.argv) > 2 and sys.argv[2] == "train":       # the with-grad forward of the MNIST configuration (B * M = 1024 patches of 32 px)
    shapes = [(64, 64, 8, 3, 1), (64, 128, 8, 3, 2), (128, 128, 4, 3, 1), (64, 128, 8, 1, 2)]
for ci, co, h, k, st in shapes:
    pad = k // 2
    ho = (h + 2 * pad - k) // st + 1
    w = torch.randn((co, ci, k, k), device=dev) * 0.05
    packed = hip._pack_conv(w)
    alpha, shift = torch.ones(co, device=dev), torch.zeros(co, device=dev)
    cv = hip.Conv(ci, co, k, k, st, pad, packed.data_ptr(), alpha.data_ptr(), shift.data_ptr(), None)
    x = torch.randn((n, h, h, ci), device=dev)
    y = torch.empty((n, ho, ho, co), device=dev)
    s = torch.cuda.current_stream().cuda_stream

    def run():
        rc = L.ipsx_conv2d_affine_nhwc(C.byref(cv), x.data_ptr(), None, y.data_ptr(), n, h, h, 1, s)
        assert rc == 0
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    flop = 2.0 * n * ho * ho * co * ci * k * k
    gb = (x.numel() + y.numel()) * 4 / 1e9
    print("%4d -> %4d  %2dx%-2d k%d s%d   %7.3f ms  %6.1f TFLOP/s (%.2f of 157.3)   in+out %.2f GB -> %5.0f GB/s" % (
        ci, co, h, h, k, st, dt * 1e3, flop / dt / 1e12, flop / dt / 1e12 / 157.3, gb, gb / dt))
