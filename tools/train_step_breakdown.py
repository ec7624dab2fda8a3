#!/usr/bin/env python
"""Where a training step's time goes (reference training/iterative.py:105-189 at config/mnist_config.yml sizes,
32-px patches): ips() under no-grad on the HIP path vs forward + backward + AdamW on stock ROCm ops, eager and
captured in a HIP graph (ips_amd/training/graphed.py).

    python tools/train_step_breakdown.py [--batch 16] [--steps 20]
"""
import argparse
import os
import sys
import time

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ips_amd import synth
from ips_amd.architecture import IPSNet
from ips_amd.training import iterative as loops

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--graph", action="store_true")
ap.add_argument("--fused", action="store_true", help="torch.optim.AdamW(fused=True): one kernel for all parameter tensors")
ap.add_argument("--channels-last", action="store_true", help="encoder weights and activations in torch.channels_last")
ap.add_argument("--profile", action="store_true", help="per-kernel device time of the steady-state forward+backward+AdamW (torch.profiler)")
args = ap.parse_args()

dev = torch.device("cuda:0")
conf = synth.mnist_conf(N=2500, M=64, I=64, B=args.batch, B_seq=args.batch, n_epoch=10, n_epoch_warmup=1, lr=1e-3, wd=0.1)
net = synth.fill_weights(IPSNet(dev, conf), 7).to(dev)
x = synth.make_patches(conf, args.batch, seed=3).to(dev)
labels = {t['name']: (torch.randint(0, 10, (args.batch,), device=dev) if t['act_fn'] == 'softmax'
                      else (torch.rand(args.batch, 10, device=dev) < 0.3).float()) for t in conf.tasks.values()}
crit = {t['name']: (nn.NLLLoss() if t['act_fn'] == 'softmax' else nn.BCELoss()) for t in conf.tasks.values()}
opt = torch.optim.AdamW(net.parameters(), lr=1e-3, weight_decay=conf.wd, fused=args.fused)
net.train()
if args.channels_last:
    net.encoder.to(memory_format=torch.channels_last)


def timed(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


state = {}


def do_ips():
    state["mp"], state["pos"] = net.ips(x)


def do_step():
    opt.zero_grad()
    preds = net(state["mp"], state["pos"])
    loss = 0
    for t in conf.tasks.values():
        p = preds[t['name']].squeeze(-1)
        loss = loss + (crit[t['name']](torch.log(p + conf.eps), labels[t['name']]) if t['act_fn'] == 'softmax'
                       else crit[t['name']](p.view(-1), labels[t['name']].view(-1)))
    (loss / len(conf.tasks)).backward()
    opt.step()


t_ips = timed(do_ips, args.steps)
if args.graph:                      # a GraphedStep wants a fresh optimizer: measure it in a run of its own
    t_step = float("nan")
else:
    t_step = timed(do_step, args.steps)
print("ips() %.2f ms   forward+backward+AdamW%s (eager) %.2f ms   -> %.1f images/s" % (
    t_ips, " (fused)" if args.fused else "", t_step, args.batch / (1e-3 * (t_ips + t_step))))
if not args.graph:
    # what a training loop sees: the host enqueues the eager step while the GPU is still busy with ips(), so the step's
    # host time (launch-bound when timed alone) hides behind the selection pass
    t_both = timed(lambda: (do_ips(), do_step()), args.steps)
    print("ips() + eager step back to back: %.2f ms per iteration -> %.1f images/s" % (t_both, args.batch / (1e-3 * t_both)))
if args.profile:
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(10):
            do_step()
        torch.cuda.synchronize()
    rows = sorted((e for e in prof.key_averages() if e.device_time_total > 0), key=lambda e: -e.device_time_total)
    tot = sum(e.device_time_total for e in rows)
    # (rows WITH device time only: rounds 2-4 also counted the profiler's runtime-side records - "679 kernels" were 274)
    print("device time per step %.3f ms over %d kernels" % (tot / 1e4, sum(e.count for e in rows) // 10))
    for e in rows[:30]:
        print("  %-90s %4d x %8.1f us  %5.1f%%" % (e.key[:90], e.count // 10, e.device_time_total / e.count, 100 * e.device_time_total / tot))
if args.graph:
    from ips_amd.training.graphed import GraphedStep
    gs = GraphedStep(net, crit, opt, conf)
    t_g = timed(lambda: gs(state["mp"], state["pos"], labels), args.steps)
    print("forward+backward+AdamW as one HIP graph %.2f ms   -> %.1f images/s" % (t_g, args.batch / (1e-3 * (t_ips + t_g))))
