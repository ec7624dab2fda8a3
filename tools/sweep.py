#!/usr/bin/env python
"""Throughput of ips() over batch size and patches per image (Megapixel-MNIST shape, 32-px patches, M = I = 64).

    python tools/sweep.py > profiles/<name>.md
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ips_amd import synth
from ips_amd.architecture import IPSNet

dev = torch.device("cuda:0")


def rate(B, N, precision, steps=10):
    os.environ["IPSX_PRECISION"] = precision
    conf = synth.mnist_conf(N=N, M=64, I=64)
    net = synth.fill_weights(IPSNet(dev, conf), 7).to(dev).eval()
    x = synth.make_patches(conf, B, seed=3).to(dev)
    for _ in range(3):
        net.ips(x)
    best = float("inf")
    for _ in range(3):                       # best of three timed runs: a new shape's first runs still hit allocator churn
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            net.ips(x)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps)
    return B * N / best, best * 1e3


print("| B | N | fp32 M patches/s | ms | fp32x3 M patches/s | ms |")
print("|---|---|---|---|---|---|")
for B, N in [(1, 2500), (2, 2500), (4, 2500), (8, 2500), (16, 2500), (32, 2500), (64, 2500),
             (1, 10000), (4, 10000), (16, 10000), (1, 40000), (4, 40000)]:
    a, ta = rate(B, N, "fp32")
    b, tb = rate(B, N, "fp32x3")
    print("| %d | %d | %.2f | %.2f | %.2f | %.2f |" % (B, N, a / 1e6, ta, b / 1e6, tb), flush=True)
