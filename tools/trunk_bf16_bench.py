#!/usr/bin/env python
"""The bf16 trunk's three builds ALONE (no selection loop beside them): patches/s of plan.encode at whole rounds of each
build (a round = 8 patches per unit for builds 1 and 2, 16 for build 3) and at the headline's part sizes.
    python tools/trunk_bf16_bench.py"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["IPSX_PRECISION"] = "bf16"
from ips_amd import hip, synth
from ips_amd.architecture import IPSNet

dev = torch.device("cuda:0")
conf = synth.mnist_conf(N=2500)
net = synth.fill_weights(IPSNet(dev, conf), 7).to(dev).eval()
x = synth.make_patches(conf, 20, seed=21).reshape(-1, 1, 32, 32).contiguous().to(dev)
plan = hip.EncoderPlan(net.encoder, True)
fn = hip.lib().ipsx_dbg_bf16_build
fn.restype, fn.argtypes = None, [C.c_int]
units = hip.device_geometry(dev).cus
for n in (8 * units * 10, 16 * units * 5, 21504, 11264, 6144, 40000):
    row = []
    for b in (1, 2, 3):
        fn(b)
        xs = x[:n]
        for _ in range(3):
            plan.encode(xs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            plan.encode(xs)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        row.append("build %d: %6.1f us  %5.2f M patches/s" % (b, dt * 1e6, n / dt / 1e6))
    print("%6d patches   " % n + "   ".join(row))
fn(0)
