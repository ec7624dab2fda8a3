#!/usr/bin/env python
"""World-size > 1 check of ips_amd.dist.ips_sharded on the GPU path.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \\
        --master-port 29541 tools/dist_check.py [--backend gloo|nccl] [--share-gpu]

With one GPU per rank use the default (nccl = RCCL).  On a 1-GPU box `--backend gloo --share-gpu`
puts every rank on cuda:0 and moves the (tiny) messages through host memory, so the sharded HIP path -
part-wise indexed encode, logits, resumable scan on the side stream, owner all-reduce - still runs
with world_size > 1.  Every rank compares against the committed reference fixtures and against its
own single-GPU ips(); exit code 0 = all equal.
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ips_amd import dist as ipsd            # noqa: E402
from tests.util import Golden               # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--backend", default="nccl")
ap.add_argument("--share-gpu", action="store_true")
ap.add_argument("--tournament", action="store_true", help="check ips_tournament against oracle.Oracle.tournament instead")
ap.add_argument("--cases", default="mnist_ragged,mnist_full,mnist_native50,cam_b2,cam_small,traffic_tiny")
ap.add_argument("--precision", default="fp32", choices=["fp32", "fp32x3", "bf16"],
                help="IPSX_PRECISION of the run.  Other than fp32 (BASELINE configs[4]: the bf16 matrix pipe) the reference "
                     "fixtures do not apply - the sharded selection must then equal the rank's own single-GPU selection at "
                     "the same precision, bit for bit")
ap.add_argument("--storage", default="f32", choices=["f32", "f16", "bf16"], help="storage type of the patch tensor (configs[4]: f16)")
ap.add_argument("--bench-shape", action="store_true",
                help="also the headline shape (16 x 2500 patches of 32 px, synthetic): the launch-aware partition has three parts there")
args = ap.parse_args()
os.environ["IPSX_PRECISION"] = args.precision

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", rank)))
torch.cuda.set_device(dev)
if args.backend == "nccl":
    dist.init_process_group("nccl", device_id=dev)
else:
    dist.init_process_group(args.backend)
bad = 0
try:
    for case in args.cases.split(","):
        try:
            g = Golden(case)
        except (OSError, KeyError):
            continue
        net = g.net(dev)
        x = g.patches().to(dev)
        N = x.shape[1]
        if args.tournament:
            from oracle.oracle import Oracle
            lo, hi = ipsd.slab_span(N, rank, world)
            mem_patch, mem_pos, mem_idx = ipsd.ips_tournament(net, x[:, lo:hi].contiguous(), N)
            cpu = g.net("cpu")
            want = Oracle(cpu).tournament(g.patches().numpy(), cpu.pos_enc.numpy() if g.conf.use_pos else None, world)
            ok = np.array_equal(mem_idx.cpu().numpy(), want)
            ok = ok and torch.equal(mem_patch, torch.stack([x[b][mem_idx[b]] for b in range(x.shape[0])]))
            print("rank %d/%d %-14s N=%-6d tournament %s" % (rank, world, case, N, "ok" if ok else "MISMATCH"), flush=True)
            bad += not ok
            continue
        if args.storage != "f32":
            if not (net.is_image and tuple(x.shape[2:]) == (1, 32, 32)):
                continue                                   # half-stored patches exist for the fused 1x32x32 trunk only
            x = x.to({"f16": torch.float16, "bf16": torch.bfloat16}[args.storage])
        plan = ipsd.shard_plan(net, x.shape[0], N, world, tuple(x.shape[2:]))
        mine = plan.indices(rank).to(dev)
        mem_patch, mem_pos, mem_idx = ipsd.ips_sharded(net, x[:, mine].contiguous(), N)
        full_patch, full_pos = net.ips(x)
        ok = torch.equal(mem_idx, net.last_mem_idx)
        if args.precision == "fp32":
            ok = ok and np.array_equal(mem_idx.cpu().numpy(), g.mem_idx)
        ok = ok and torch.equal(mem_patch, full_patch) and (mem_pos is None or torch.equal(mem_pos, full_pos))
        print("rank %d/%d %-14s N=%-6d %s %s parts=%d %s" % (rank, world, case, N, args.precision, args.storage, len(plan.its) - 1,
                                                          "ok" if ok else "MISMATCH"), flush=True)
        bad += not ok
    if args.bench_shape and not args.tournament:
        from ips_amd import synth
        from ips_amd.architecture import IPSNet
        conf, B = synth.bench_workload("mnist")
        net = synth.fill_weights(IPSNet(dev, conf), 7).to(dev).eval()
        x = synth.make_patches(conf, B, seed=21).to(dev)
        if args.storage != "f32":
            x = x.to({"f16": torch.float16, "bf16": torch.bfloat16}[args.storage])
        plan = ipsd.shard_plan(net, B, conf.N, world, tuple(x.shape[2:]))
        mem_patch, mem_pos, mem_idx = ipsd.ips_sharded(net, x[:, plan.indices(rank).to(dev)].contiguous(), conf.N)
        full_patch, full_pos = net.ips(x)
        ok = torch.equal(mem_idx, net.last_mem_idx) and torch.equal(mem_patch, full_patch) and torch.equal(mem_pos, full_pos)
        if args.precision == "fp32":
            z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "bench_mnist.npz"))
            ok = ok and np.array_equal(mem_idx.cpu().numpy(), z["trace_idx"][:, -1].astype(np.int64))
        print("rank %d/%d %-14s N=%-6d %s %s parts=%d launches=%s %s" % (rank, world, "bench_mnist", conf.N, args.precision, args.storage,
              len(plan.its) - 1, plan.launches(rank), "ok" if ok else "MISMATCH"), flush=True)
        bad += not ok
    t = torch.tensor([bad], dtype=torch.int64)
    if args.backend == "nccl":
        t = t.to(dev)
    dist.all_reduce(t)
    bad = int(t.item())
finally:
    dist.destroy_process_group()
sys.exit(1 if bad else 0)
