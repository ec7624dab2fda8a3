#!/usr/bin/env python
"""Why is the training step's HIP-graph replay slower in DEVICE time than the eager step (2.89 vs 2.76 ms)?
Per-kernel device time (torch.profiler) of: the eager step as tools/train_step_breakdown.py runs it (fresh gradients:
zero_grad(set_to_none=True)), the eager step the way the graph must run it (static gradient buffers: zero_grad(
set_to_none=False), capturable AdamW), and the replay itself.
    python tools/train_graph_vs_eager.py"""
import os
import sys

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ips_amd import synth
from ips_amd.architecture import IPSNet
from ips_amd.training.graphed import GraphedStep
from ips_amd.training.iterative import compute_loss
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda:0")
B = 16
conf = synth.mnist_conf(N=2500, M=64, I=64, B=B, B_seq=B, n_epoch=10, n_epoch_warmup=1, lr=1e-3, wd=0.1)


def setup(fused):
    torch.manual_seed(0)
    net = synth.fill_weights(IPSNet(dev, conf), 7).to(dev)
    net.train()
    crit = {t['name']: (nn.NLLLoss() if t['act_fn'] == 'softmax' else nn.BCELoss()) for t in conf.tasks.values()}
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3, weight_decay=conf.wd, fused=fused)
    x = synth.make_patches(conf, B, seed=3).to(dev)
    labels = {t['name']: (torch.randint(0, 10, (B,), device=dev) if t['act_fn'] == 'softmax'
                          else (torch.rand(B, 10, device=dev) < 0.3).float()) for t in conf.tasks.values()}
    mp, pos = net.ips(x)
    return net, crit, opt, mp, pos, labels


def prof(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as p:
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    rows = [e for e in p.key_averages() if e.device_time_total > 0]
    tot = sum(e.device_time_total for e in rows) / n / 1e3
    cnt = sum(e.count for e in rows) // n
    return tot, cnt, {e.key: (e.count / n, e.device_time_total / n) for e in rows}


def group(d):
    out = {}
    for k, (c, t) in d.items():
        g = ("fill" if "FillFunctor" in k or "fillBuffer" in k else
             "add / copy elementwise" if ("CUDAFunctor_add" in k or "copy" in k.lower() or "elementwise" in k) else
             "optimizer" if ("multi_tensor" in k or "adam" in k.lower()) else
             "libipsx" if "ipsx::" in k else "other")
        a = out.setdefault(g, [0.0, 0.0])
        a[0] += c; a[1] += t
    return out


net, crit, opt, mp, pos, labels = setup(True)


def eager_fresh():
    opt.zero_grad()
    loss, _ = compute_loss(net, mp, pos, crit, labels, conf)
    loss.backward()
    opt.step()


def eager_static():
    opt.zero_grad(set_to_none=False)
    loss, _ = compute_loss(net, mp, pos, crit, labels, conf)
    loss.backward()
    opt.step()


res = {"eager, fresh gradients (set_to_none=True), fused AdamW": prof(eager_fresh)}
res["eager, static gradient buffers (set_to_none=False), fused AdamW"] = prof(eager_static)
net, crit, opt, mp, pos, labels = setup(True)
gs = GraphedStep(net, crit, opt, conf)
res["HIP-graph replay (GraphedStep)"] = prof(lambda: gs(mp, pos, labels))
for name, (tot, cnt, d) in res.items():
    print("%-70s %.3f ms device over %d kernels" % (name, tot, cnt))
    for g, (c, t) in sorted(group(d).items(), key=lambda kv: -kv[1][1]):
        print("      %-24s %6.1f kernels  %8.1f us" % (g, c, t))
