#!/usr/bin/env python
"""Train / evaluate IPS end to end on one MI355X - the role of the reference's main.py (/root/reference/main.py),
for the dataset that can exist offline: Megapixel MNIST in the reference's on-disk format (written by
``ips_amd.data.megapixel_mnist.write_synthetic`` when ``--make-synthetic`` is given; a directory produced by the
reference's make_mnist.py works unchanged).

    python -m ips_amd.main --make-synthetic /tmp/mmnist --width 1500 --height 1500 --n-train 64 --n-test 16
    python -m ips_amd.main --data-dir /tmp/mmnist --epochs 2 --patch 32 --M 64 --I 64 [--sparse] [--lazy]

Configuration defaults are the reference's config/mnist_config.yml; a YAML file with the same keys can be passed
with --config.  The loops are ips_amd/training/iterative.py (same names and results as the reference's).
"""

import argparse
import json
import os
import time

import numpy as np
import torch
import yaml
from torch import nn
from torch.utils.data import DataLoader

from . import synth
from .architecture.ips_net import IPSNet
from .data import megapixel_mnist as mm
from .training.iterative import evaluate, train_one_epoch
from .utils.utils import Logger, Struct


def build_conf(args):
    if args.config:
        with open(args.config) as f:
            c = yaml.load(f, Loader=yaml.FullLoader)
    else:
        c = dict(synth.mnist_conf().__dict__, n_epoch=150, n_epoch_warmup=10, lr=1e-3, wd=0.1, n_worker=8,
                 pin_memory=True, track_efficiency=False, track_epoch=0, shuffle=True)
    c["data_dir"] = args.data_dir
    with open(os.path.join(args.data_dir, "parameters.json")) as f:
        par = json.load(f)
    ps = [args.patch, args.patch] if args.patch else c["patch_size"]
    st = [args.stride, args.stride] if args.stride else (ps if args.patch else c["patch_stride"])
    n = ((par["height"] - ps[0]) // st[0] + 1) * ((par["width"] - ps[1]) // st[1] + 1)
    c.update(patch_size=ps, patch_stride=st, N=n)
    for key, val in (("M", args.M), ("I", args.I), ("B", args.B), ("B_seq", args.B_seq), ("n_epoch", args.epochs),
                     ("n_worker", args.workers), ("seed", args.seed)):
        if val is not None:
            c[key] = val
    if args.epochs is not None:
        c["n_epoch_warmup"] = min(c["n_epoch_warmup"], max(1, args.epochs // 10))
    if args.lazy:
        c["eager"] = False
    c["hip_graph"] = bool(args.hip_graph)
    return Struct(**c)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--make-synthetic", metavar="DIR", help="write a synthetic dataset in the reference's format and exit")
    ap.add_argument("--width", type=int, default=1500)
    ap.add_argument("--height", type=int, default=1500)
    ap.add_argument("--n-train", type=int, default=64)
    ap.add_argument("--n-test", type=int, default=16)
    ap.add_argument("--data-dir")
    ap.add_argument("--config", help="YAML with the keys of the reference's config/mnist_config.yml")
    ap.add_argument("--patch", type=int)
    ap.add_argument("--stride", type=int)
    ap.add_argument("--M", type=int)
    ap.add_argument("--I", type=int)
    ap.add_argument("--B", type=int)
    ap.add_argument("--B-seq", dest="B_seq", type=int)
    ap.add_argument("--epochs", type=int)
    ap.add_argument("--workers", type=int)
    ap.add_argument("--seed", type=int)
    ap.add_argument("--sparse", action="store_true", help="loader delivers non-zero pixels; patches are built on the GPU")
    ap.add_argument("--hip-graph", action="store_true", help="replay forward+backward+AdamW of a step as one HIP graph")
    ap.add_argument("--lazy", action="store_true", help="lazy loading: patches stay on the host (eager: False)")
    ap.add_argument("--precision", choices=["fp32", "fp32x3", "bf16"],
                    help="arithmetic of the no-grad selection pass ips() (IPSX_PRECISION): fp32 = the reference's, exact "
                         "(default); fp32x3 = fp32-grade on the bf16 matrix pipe, ~1.8x faster; bf16. The with-grad forward "
                         "and backward are fp32 in every case")
    ap.add_argument("--no-fused-adamw", action="store_true", help="torch's default (per-tensor) AdamW instead of the fused kernel")
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    args = ap.parse_args(argv)

    if args.make_synthetic:
        mm.write_synthetic(args.make_synthetic, args.n_train, args.n_test, args.width, args.height)
        print("wrote", args.make_synthetic)
        return 0
    if not args.data_dir:
        ap.error("--data-dir (or --make-synthetic) is required")

    device = torch.device(args.device)
    if args.precision:
        os.environ["IPSX_PRECISION"] = args.precision
    conf = build_conf(args)
    torch.manual_seed(conf.seed)
    np.random.seed(conf.seed)

    sparse = args.sparse and device.type == "cuda" and conf.eager
    kw = dict(num_workers=conf.n_worker, persistent_workers=conf.n_worker > 0,
              collate_fn=mm.collate_sparse if sparse else None)
    kw["pin_memory"] = bool(conf.pin_memory) and device.type == "cuda"     # SparseImages has pin_memory() too
    train_loader = DataLoader(mm.MegapixelMNIST(conf, train=True, sparse=sparse), batch_size=conf.B_seq, shuffle=True, **kw)
    test_loader = DataLoader(mm.MegapixelMNIST(conf, train=False, sparse=sparse), batch_size=conf.B_seq, shuffle=False, **kw)

    net = IPSNet(device, conf).to(device)
    # the reference builds torch.optim.AdamW(net.parameters(), lr=0, weight_decay=wd) (main.py:57): same update rule, but
    # on a GPU as ONE fused kernel over all 81 parameter tensors - the default implementation's ~1,100 small launches
    # were 1.7 ms of a 7.2 ms step (tools/train_step_breakdown.py)
    fused = device.type == "cuda" and not args.no_fused_adamw
    optimizer = torch.optim.AdamW(net.parameters(), lr=0, weight_decay=conf.wd, fused=fused)
    nll, bce = nn.NLLLoss(), nn.BCELoss()
    criterions = {t['name']: nll if t['act_fn'] == 'softmax' else bce for t in conf.tasks.values()}
    log_train, log_test = Logger(conf.tasks), Logger(conf.tasks)

    for epoch in range(conf.n_epoch):
        t0 = time.time()
        train_one_epoch(net, criterions, train_loader, optimizer, device, epoch, log_train, conf)
        t1 = time.time()
        log_train.compute_metric()
        log_train.print_stats(epoch, train=True, lr=float(optimizer.param_groups[0]['lr']),
                              images_per_s="{:.1f}".format(len(train_loader.dataset) / (t1 - t0)))
        evaluate(net, criterions, test_loader, device, log_test, conf)
        log_test.compute_metric()
        log_test.print_stats(epoch, train=False, images_per_s="{:.1f}".format(len(test_loader.dataset) / (time.time() - t1)))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
