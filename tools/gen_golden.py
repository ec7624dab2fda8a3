#!/usr/bin/env python
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (imported from /root/reference).

Only runs where the reference is mounted (this container); the fixtures it writes
are data - configuration, seeds and the reference's outputs - and are committed.
The reference publishes no tests or golden vectors of its own,
so these are the vectors that pin the oracle and the HIP path to the reference.

For every case: build the reference ``IPSNet`` on CPU, fill weights from
``ips_amd.synth.fill_weights(seed)``, draw inputs from ``ips_amd.synth.make_patches``,
run ``ips()`` (recording the memory indices after every iteration and the scores
the selection was based on) and ``forward()``.

    python tools/gen_golden.py            # all cases
    python tools/gen_golden.py mnist_mini # selected cases
"""

import json
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

_TOK1 = {'task0': {'id': 0, 'name': 'majority', 'act_fn': 'softmax', 'metric': 'accuracy'}}

# name -> (conf, B, weight seed, patch seed, torch seed)
CASES = {
    "mnist_mini": (synth.mnist_conf(N=300, M=16, I=16), 2, 1, 3, 0),
    "mnist_ragged": (synth.mnist_conf(N=301, M=16, I=24), 2, 2, 4, 0),
    "mnist_onechunk": (synth.mnist_conf(N=40, M=16, I=64), 3, 3, 5, 0),
    "mnist_tok1": (synth.mnist_conf(N=200, M=8, I=32, n_token=1, tasks=_TOK1), 2, 4, 6, 0),
    "mnist_shuffle_batch": (synth.mnist_conf(N=300, M=16, I=16, shuffle=True), 2, 5, 7, 11),
    "mnist_shuffle_instance": (synth.mnist_conf(N=300, M=16, I=16, shuffle=True,
                                                shuffle_style='instance'), 2, 6, 8, 12),
    "mnist_full": (synth.mnist_conf(N=2500, M=64, I=64), 1, 7, 9, 0),
    "mnist_native50": (synth.mnist_conf(N=900, M=100, I=100, patch=50), 1, 8, 10, 0),
    "traffic_tiny": (synth.traffic_conf(N=20, M=4, I=6, patch=64), 1, 9, 11, 0),
    "traffic_full": (synth.traffic_conf(N=192, M=16, I=32, patch=100), 1, 10, 12, 0),
    "cam_small": (synth.camelyon_conf(N=4096, M=256, I=256), 1, 11, 13, 0),
    "cam_b2": (synth.camelyon_conf(N=1000, M=32, I=48), 2, 12, 14, 0),
    # no positional encoding + 93 % identical (blank) patches: exact score ties straddle the top-M boundary in
    # every iteration, so the indices are whatever torch.topk's CPU implementation returns under ties
    "mnist_ties": (synth.mnist_conf(N=300, M=16, I=16, use_pos=False), 2, 13, 15, 0),
    "mnist_ties_wide": (synth.mnist_conf(N=1200, M=8, I=1000, use_pos=False, blank_frac=0.995), 1, 14, 16, 0),   # k*64 <= n: partial_sort branch
}


def state_checksum(net):
    return float(sum(v.double().abs().sum().item() for k, v in net.state_dict().items()
                     if not k.endswith("num_batches_tracked")))


def run_case(name, ref_ips):
    conf, B, wseed, pseed, tseed = CASES[name]
    net = ref_ips.IPSNet(torch.device("cpu"), conf)
    synth.fill_weights(net, wseed)
    net.eval()
    x = synth.make_patches(conf, B, seed=pseed)

    trace_idx, trace_score, gaps, seen = [], [], [], {}
    orig_sel, orig_shuffle = net.score_and_select, net.do_shuffle

    def select(emb, emb_pos, M, idx):
        scored = emb_pos if torch.is_tensor(emb_pos) else emb
        sc = net.transf.get_scores(scored)
        mem_emb, mem_idx = orig_sel(emb, emb_pos, M, idx)
        srt = torch.sort(sc, dim=-1, descending=True)[0]
        trace_idx.append(mem_idx.clone())
        trace_score.append(srt[:, :M].clone())
        if srt.shape[1] > M:
            gaps.append(((srt[:, M - 1] - srt[:, M]) / srt[:, M - 1]).min().item())
        return mem_emb, mem_idx

    def shuffle(patches, pos_enc):
        p, pe = orig_shuffle(patches, pos_enc)
        seen["shuffled"] = p
        return p, pe

    net.score_and_select = select
    net.do_shuffle = shuffle
    torch.manual_seed(tseed)
    if conf.shuffle:
        # draw the permutation exactly as do_shuffle will, then rewind the generator
        state = torch.get_rng_state()
        if conf.shuffle_style == 'batch':
            perm = torch.randperm(conf.N).unsqueeze(0).expand(B, -1)
        else:
            perm = torch.rand((B, conf.N)).argsort(1)
        torch.set_rng_state(state)
    with torch.no_grad():
        mem_patch, mem_pos = net.ips(x)
        preds = net(mem_patch, mem_pos)
        emb_head = net.encoder(x[0, :8].reshape(-1, *x.shape[2:])).flatten(1)

    out = dict(
        conf=json.dumps(conf.__dict__), B=B, weight_seed=wseed, patch_seed=pseed, torch_seed=tseed,
        state_checksum=state_checksum(net),
        trace_idx=torch.stack(trace_idx, 1).numpy(),          # (B, n_iter, M), in the order ips() saw
        trace_score=torch.stack(trace_score, 1).numpy(),      # (B, n_iter, M) sorted scores
        min_rel_gap=np.float64(min(gaps) if gaps else np.inf),
        emb_head=emb_head.numpy(),                            # encoder output of patches [0, :8]
        mem_patch_sum=mem_patch.double().sum(dim=tuple(range(2, mem_patch.dim()))).numpy(),
    )
    if conf.shuffle:
        out["perm"] = perm.numpy()
        # the recorded permutation must reproduce what the reference's do_shuffle produced
        take = perm.view(B, -1, *(1,) * (x.dim() - 2)).expand_as(x)
        assert torch.equal(torch.gather(x, 1, take), seen["shuffled"])
    if mem_pos is not None:
        out["mem_pos_sum"] = mem_pos.double().sum(-1).numpy()
    for k, v in preds.items():
        out["pred_" + k] = v.numpy()
    os.makedirs(GOLDEN, exist_ok=True)
    np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), **out)
    print("{:24s} B={} N={} M={} I={} n_iter={} min_rel_gap={:.3e}".format(
        name, B, conf.N, conf.M, conf.I, out["trace_idx"].shape[1], out["min_rel_gap"]))


def main():
    ref_ips, _, _ = import_reference()
    names = sys.argv[1:] or list(CASES)
    for n in names:
        run_case(n, ref_ips)


if __name__ == "__main__":
    main()
