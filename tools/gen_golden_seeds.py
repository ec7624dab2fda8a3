#!/usr/bin/env python
"""Generate tests/golden/seeds_<family>.npz by RUNNING THE REFERENCE (imported from /root/reference) on >= 20
(weights, inputs) seeds per configuration family (``ips_amd.synth.SEED_FAMILIES`` / ``seed_case``): half of the cases
with the learned queries at their default scale (flat attention -> small top-M gaps), half sharpened; Megapixel-MNIST
families alternate stroke-like sparse images with noise patches.

Per case: the memory indices after every iteration (uint16), the relative top-M boundary gap of every iteration (decides WHICH patches
stay), the smallest relative step between neighbouring scores of the sorted top M + 1 (decides their ORDER) and the final predictions.  Data only.  tests/test_seed_sweep.py holds the oracle (CPU) and the HIP path (GPU) to them.

    python tools/gen_golden_seeds.py [family ...]
"""

import os
import sys

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np
import torch

from ips_amd import synth
from tools.refimport import import_reference

GOLDEN = os.path.join(REPO, "tests", "golden")


def run_case(ref_ips, family, k):
    conf, B, wseed, q_gain, x = synth.seed_case(family, k)
    net = ref_ips.IPSNet(torch.device("cpu"), conf)
    synth.fill_weights(net, wseed, q_gain=q_gain)
    net.eval()
    trace, gaps, ogaps = [], [], []
    orig = net.score_and_select

    def select(emb, emb_pos, M, idx):
        scored = emb_pos if torch.is_tensor(emb_pos) else emb
        sc = net.transf.get_scores(scored)
        mem_emb, mem_idx = orig(emb, emb_pos, M, idx)
        srt = torch.sort(sc, dim=-1, descending=True)[0]
        trace.append(mem_idx.clone())
        gaps.append(((srt[:, M - 1] - srt[:, M]) / srt[:, M - 1]).float())
        top = srt[:, :M + 1]                                  # smallest relative step between neighbours of the sorted
        ogaps.append(((top[:, :-1] - top[:, 1:]) / top[:, :-1]).min(-1)[0].float())      # top M + 1: decides the ORDER
        return mem_emb, mem_idx

    net.score_and_select = select
    with torch.no_grad():
        mem_patch, mem_pos = net.ips(x)
        preds = net(mem_patch, mem_pos)
    out = {"trace_idx": torch.stack(trace, 1).numpy().astype(np.uint16),
           "rel_gap": torch.stack(gaps, 1).numpy().astype(np.float32),
           "order_gap": torch.stack(ogaps, 1).numpy().astype(np.float32),
           "x_sum": np.float64(x.double().sum().item())}
    for name, v in preds.items():
        out["pred_" + name] = v.numpy()
    return out


def main():
    ref_ips, _, _ = import_reference()
    for family in sys.argv[1:] or list(synth.SEED_FAMILIES):
        conf, B, n = synth.SEED_FAMILIES[family]()
        pack, gmin = {"n_case": n}, []
        if conf.use_pos:
            pack.update(synth.pos_table_record(conf))
        for k in range(n):
            for key, v in run_case(ref_ips, family, k).items():
                pack["c%d_%s" % (k, key)] = v
            gmin.append(float(pack["c%d_rel_gap" % k].min()))
        path = os.path.join(GOLDEN, "seeds_%s.npz" % family)
        np.savez_compressed(path, **pack)
        below = sum(int((pack["c%d_rel_gap" % k] <= 1e-5).sum()) for k in range(n))
        total = sum(pack["c%d_rel_gap" % k].size for k in range(n))
        print("%-8s %2d cases, B=%d N=%d M=%d I=%d: min gap per case %.1e .. %.1e, %d of %d iterations at or below 1e-5, %d KB"
              % (family, n, B, conf.N, conf.M, conf.I, min(gmin), max(gmin), below, total, os.path.getsize(path) // 1024), flush=True)


if __name__ == "__main__":
    main()
