#!/usr/bin/env python
"""Generate tests/golden/bench_*.npz by RUNNING THE REFERENCE (imported from /root/reference) on the exact inputs
bench.py times: weights ``synth.fill_weights(seed 7)``, patches ``synth.make_patches(seed 21)``.

Per workload the fixture holds the memory indices after EVERY iteration of the reference's loop
(/root/reference/architecture/ips_net.py:218-241; uint16 when N <= 65536), the relative gap between the M-th and
(M+1)-th score of every iteration (how far the selection is from a tie), and the final ``preds``.  bench.py compares
``net.last_mem_idx`` after its timed loop with the last iteration and prints the verdict in its JSON line
("parity"); tests/test_bench_parity.py replays every iteration on the GPU.  Data only - nothing of the reference's
source travels.

    python tools/gen_golden_bench.py [workload ...]
"""

import json
import os
import sys
import time

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np
import torch

from ips_amd import synth
from tools.refimport import import_reference

GOLDEN = os.path.join(REPO, "tests", "golden")
WEIGHT_SEED, PATCH_SEED = 7, 21


def run(name, ref_ips):
    conf, B = synth.bench_workload(name)
    net = ref_ips.IPSNet(torch.device("cpu"), conf)
    synth.fill_weights(net, WEIGHT_SEED)
    net.eval()
    x = synth.make_patches(conf, B, seed=PATCH_SEED)
    trace, gaps, ogaps = [], [], []
    orig = net.score_and_select

    def select(emb, emb_pos, M, idx):
        scored = emb_pos if torch.is_tensor(emb_pos) else emb
        sc = net.transf.get_scores(scored)
        mem_emb, mem_idx = orig(emb, emb_pos, M, idx)
        srt = torch.sort(sc, dim=-1, descending=True)[0]
        trace.append(mem_idx.clone())
        gaps.append(((srt[:, M - 1] - srt[:, M]) / srt[:, M - 1]).float() if srt.shape[1] > M
                    else torch.full((sc.shape[0],), float("inf")))
        top = srt[:, :M + 1]                                  # smallest relative step between neighbours of the sorted
        ogaps.append(((top[:, :-1] - top[:, 1:]) / top[:, :-1]).min(-1)[0].float())      # top M + 1: decides the ORDER
        return mem_emb, mem_idx

    net.score_and_select = select
    t0 = time.time()
    with torch.no_grad():
        mem_patch, mem_pos = net.ips(x)
        preds = net(mem_patch, mem_pos)
    dt = time.time() - t0
    idx = torch.stack(trace, 1).numpy()                          # (B, n_iter, M)
    assert idx.min() >= 0 and idx.max() < conf.N
    out = dict(conf=json.dumps(conf.__dict__), B=B, weight_seed=WEIGHT_SEED, patch_seed=PATCH_SEED,
               trace_idx=idx.astype(np.uint16 if conf.N <= 65536 else np.int32),
               rel_gap=torch.stack(gaps, 1).numpy().astype(np.float32),      # (B, n_iter): M-th vs (M+1)-th score
               order_gap=torch.stack(ogaps, 1).numpy().astype(np.float32),   # (B, n_iter): closest neighbours in the top M + 1
               mem_patch_sum=mem_patch.double().sum(dim=tuple(range(2, mem_patch.dim()))).numpy())
    for k, v in preds.items():
        out["pred_" + k] = v.numpy()
    if conf.use_pos:
        out.update(synth.pos_table_record(conf))
    np.savez_compressed(os.path.join(GOLDEN, "bench_" + name + ".npz"), **out)
    g = out["rel_gap"]
    print("{:10s} B={:2d} N={:5d} M={:3d} I={:3d} n_iter={:3d} min gap {:.2e} (final iteration {:.2e}; order: {:.2e} / {:.2e})  {:.1f} s  {:.0f} KB".format(
        name, B, conf.N, conf.M, conf.I, idx.shape[1], g.min(), g[:, -1].min(), out["order_gap"].min(), out["order_gap"][:, -1].min(), dt,
        os.path.getsize(os.path.join(GOLDEN, "bench_" + name + ".npz")) / 1024), flush=True)


def main():
    ref_ips, _, _ = import_reference()
    for name in sys.argv[1:] or list(synth.BENCH_WORKLOADS):
        run(name, ref_ips)


if __name__ == "__main__":
    main()
