#!/usr/bin/env python
"""Time the CAMELYON projector (row_stats_kernel + conv_nhwc_kernel<NORM>) alone: n rows of 2048 features -> 512.

    python tools/projector_bench.py [n_rows]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ips_amd import hip, synth
from ips_amd.architecture import IPSNet

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda:0")
conf = synth.camelyon_conf(N=n, M=256, I=256)
net = synth.fill_weights(IPSNet(dev, conf), 7).to(dev).eval()
x = synth.make_patches(conf, 1, seed=21)[0].to(dev)
plan = hip.EncoderPlan(net.encoder, False)
for _ in range(3):
    plan.encode(x)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 20
a.record()
for _ in range(reps):
    plan.encode(x)
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / reps
flop = 2.0 * n * 2048 * 512
print("projector %d rows: %.3f ms  %.1f TFLOP/s = %.3f of the fp32 MFMA peak; %.1f M rows/s" %
      (n, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3, n / ms / 1e3))
