#!/usr/bin/env python
"""Soak test: many ips() calls in every mode; allocator statistics must stay flat and results identical.

    python tools/soak.py [iterations]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ips_amd import synth
from ips_amd.architecture import IPSNet

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
conf = synth.mnist_conf(N=2500, M=64, I=64)
net = synth.fill_weights(IPSNet(dev, conf), 7).to(dev).eval()
x = synth.make_patches(conf, 16, seed=21).to(dev)
xh = x.cpu().pin_memory()
ok = True
for mode, env, inp in (("eager fp32", {}, x), ("eager fp32x3", {"IPSX_PRECISION": "fp32x3"}, x),
                       ("eager bf16", {"IPSX_PRECISION": "bf16"}, x), ("dedup", {"IPSX_DEDUP_BLANK": "1"}, x),
                       ("lazy", {}, xh), ("no overlap", {"IPSX_OVERLAP_SCAN": "0"}, x)):
    for k, v in env.items():
        os.environ[k] = v
    net.ips(inp)
    ref = net.last_mem_idx.clone()
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    peak0 = torch.cuda.max_memory_allocated()
    same = True
    for it in range(n_it):
        mp, pos = net.ips(inp)
        if it % 50 == 49:
            same = same and bool(torch.equal(net.last_mem_idx, ref))
    torch.cuda.synchronize()
    del mp, pos
    grown = torch.cuda.memory_allocated() - base
    print("%-14s %d calls  same indices %s  allocated delta %+d B  reserved %.1f MB  peak %.1f MB" % (
        mode, n_it, same, grown, torch.cuda.memory_reserved() / 2**20, torch.cuda.max_memory_allocated() / 2**20))
    ok = ok and same and abs(grown) < (64 << 20)
    for k in env:
        del os.environ[k]
sys.exit(0 if ok else 1)
