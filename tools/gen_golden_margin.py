#!/usr/bin/env python
"""Generate tests/golden/margin.npz by RUNNING THE REFERENCE (imported from /root/reference) once more on the inputs of
the committed fixtures, recording what the parity MARGIN is measured against (tests/test_parity_margin.py):

    <fixture>:edge_score  (B, n_iter, 2) float32   the reference's M-th and (M+1)-th largest score of every iteration
    <fixture>:edge_idx    (B, n_iter, 2) int32     the patches that hold them (global patch indices)

The two candidates either side of the top-M boundary decide which patches stay; with the reference's own scores for
them on record, "how far is this arithmetic from flipping the reference's selection" becomes a number: the boundary gap
divided by how far the oracle's (or a kernel's) score of either candidate lies from the reference's.

While it runs, every case must reproduce the memory indices of its committed fixture (so the new record and the old
ones describe the same runs).  Fixtures with a shuffle or with structural ties (gap 0 by construction) are not covered.
Data only - nothing of the reference's source travels.

    python tools/gen_golden_margin.py
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
SMALL = ["mnist_mini", "mnist_ragged", "mnist_onechunk", "mnist_tok1", "mnist_full", "mnist_native50", "traffic_tiny",
         "traffic_full", "cam_small", "cam_b2"]


def run(ref_ips, conf, wseed, q_gain, x, want_idx):
    net = ref_ips.IPSNet(torch.device("cpu"), conf)
    synth.fill_weights(net, wseed, q_gain=q_gain)
    net.eval()
    trace, es, ei = [], [], []
    orig = net.score_and_select

    def select(emb, emb_pos, M, idx):
        scored = emb_pos if torch.is_tensor(emb_pos) else emb
        sc = net.transf.get_scores(scored)
        mem_emb, mem_idx = orig(emb, emb_pos, M, idx)
        trace.append(mem_idx.clone())
        if sc.shape[1] > M:
            srt, order = torch.sort(sc, dim=-1, descending=True)
            es.append(srt[:, M - 1:M + 1].float().clone())
            ei.append(torch.gather(idx, 1, order[:, M - 1:M + 1]).clone())
        else:                                                    # a chunk that adds nothing to choose from
            es.append(torch.full((sc.shape[0], 2), float("nan")))
            ei.append(torch.full((sc.shape[0], 2), -1, dtype=torch.int64))
        return mem_emb, mem_idx

    net.score_and_select = select
    with torch.no_grad():
        net.ips(x)
    got = torch.stack(trace, 1).numpy().astype(np.int64)
    assert np.array_equal(got, want_idx.astype(np.int64)), "the reference's run differs from the committed fixture"
    return torch.stack(es, 1).numpy().astype(np.float32), torch.stack(ei, 1).numpy().astype(np.int32)


def main():
    ref_ips, _, _ = import_reference()
    pack = {}

    def put(name, e_s, e_i, t0):
        pack[name + ":edge_score"], pack[name + ":edge_idx"] = e_s, e_i
        gap = (e_s[..., 0] - e_s[..., 1]) / e_s[..., 0]
        print("%-24s %s  min rel gap %.2e  %.1f s" % (name, e_s.shape, np.nanmin(gap), time.time() - t0), flush=True)

    for name in SMALL:
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        conf = synth.Conf(**json.loads(str(z["conf"])))
        assert not conf.shuffle
        t0 = time.time()
        x = synth.make_patches(conf, int(z["B"]), seed=int(z["patch_seed"]))
        put(name, *run(ref_ips, conf, int(z["weight_seed"]), 8.0, x, z["trace_idx"]), t0)
    for family in synth.SEED_FAMILIES:
        z = np.load(os.path.join(GOLDEN, "seeds_%s.npz" % family))
        for k in range(int(z["n_case"])):
            t0 = time.time()
            conf, B, wseed, q_gain, x = synth.seed_case(family, k)
            put("seeds_%s:c%d" % (family, k), *run(ref_ips, conf, wseed, q_gain, x, z["c%d_trace_idx" % k]), t0)
    for name in synth.BENCH_WORKLOADS:
        z = np.load(os.path.join(GOLDEN, "bench_%s.npz" % name))
        conf, B = synth.bench_workload(name)
        t0 = time.time()
        x = synth.make_patches(conf, B, seed=21)
        put("bench_" + name, *run(ref_ips, conf, 7, 8.0, x, z["trace_idx"]), t0)
    path = os.path.join(GOLDEN, "margin.npz")
    np.savez_compressed(path, **pack)
    print("%s: %d records, %d KB" % (os.path.relpath(path, REPO), len(pack) // 2, os.path.getsize(path) // 1024))


if __name__ == "__main__":
    main()
