#!/usr/bin/env python
"""Diagnostic: gradients of the encoder in training mode - fused path (training/fused_encoder.py) and stock ROCm ops, each
against an fp64 CPU evaluation of the same modules.
    python tools/train_grad_check.py [mnist|traffic]
"""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ips_amd import synth
from ips_amd.architecture import IPSNet
from ips_amd.training import fused_encoder

kind = sys.argv[1] if len(sys.argv) > 1 else "mnist"
dev = torch.device("cuda:0")
conf, patch = (synth.mnist_conf(N=64, M=8, I=8), 32) if kind == "mnist" else (synth.traffic_conf(N=64, M=8, I=8, patch=64), 64)
net_a = synth.fill_weights(IPSNet(dev, conf), 5).to(dev).train()
net_b = copy.deepcopy(net_a)
net_c = copy.deepcopy(net_a).double().cpu()
g = torch.Generator(device="cpu").manual_seed(1)
P = 24
x = torch.rand((P, conf.n_chan_in, patch, patch), generator=g)
t = torch.randn((P, conf.D), generator=g)


def rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max().clamp_min(1e-300))


(((fused_encoder.encode(net_a.encoder, x.to(dev)) - t.to(dev)) ** 2).mean()).backward()
(((net_b.encoder(x.to(dev)).flatten(1) - t.to(dev)) ** 2).mean()).backward()
(((net_c.encoder(x.double()).flatten(1) - t.double()) ** 2).mean()).backward()
print("%-28s %12s %12s %12s" % ("gradient", "fused-fp64", "stock-fp64", "fused-stock"))
for (n, pa), (_, pb), (_, pc) in zip(net_a.encoder.named_parameters(), net_b.encoder.named_parameters(),
                                     net_c.encoder.named_parameters()):
    print("%-28s %12.2e %12.2e %12.2e" % (n, rel(pa.grad, pc.grad), rel(pb.grad, pc.grad), rel(pa.grad, pb.grad)))

# gradients with respect to the stem's and every block's output
net_a.zero_grad(); net_b.zero_grad()
taps_a = []
emb = fused_encoder.encode(net_a.encoder, x.to(dev), taps_a)
for h in taps_a:
    h.retain_grad()
((emb - t.to(dev)) ** 2).mean().backward()
taps_b = []
mods = list(net_b.encoder.children())
h = x.to(dev)
for m in mods[:4]:
    h = m(h)
taps_b.append(h)
for stage in mods[4:-1]:
    for blk in stage:
        h = blk(h)
        taps_b.append(h)
for h_ in taps_b:
    h_.retain_grad()
((mods[-1](h).flatten(1) - t.to(dev)) ** 2).mean().backward()
for k, (ha, hb) in enumerate(zip(taps_a, taps_b)):
    print("tap %d  value %.2e   grad %.2e" % (k, rel(ha.detach(), hb.detach()), rel(ha.grad, hb.grad)))
for k in (0, 1, 2):
    ma, mb = taps_a[k] > 0, taps_b[k] > 0
    print("tap %d: relu masks differ at %d of %d; sum(mask*grad) fused %s stock %s" % (
        k, int((ma != mb).sum()), ma.numel(), float((taps_a[k].grad * ma).sum()), float((taps_b[k].grad * mb).sum())))
    print("   grad where y == 0: fused max |g| %.3e, stock max |g| %.3e" % (
        float((taps_a[k].grad * (~ma)).abs().max()), float((taps_b[k].grad * (~mb)).abs().max())))
ma, mb = taps_a[1] > 0, taps_b[1] > 0
pos = (ma != mb).nonzero()
for p_ in pos:
    i = tuple(int(v) for v in p_)
    print("flip at", i, "fused y %.6e stock y %.6e grad there %.3e" % (float(taps_a[1][i]), float(taps_b[1][i]), float(taps_b[1].grad[i])))
