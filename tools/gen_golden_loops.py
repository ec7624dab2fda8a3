#!/usr/bin/env python
"""Generate tests/golden/loop_*.npz by RUNNING THE REFERENCE's training/iterative.py (train_one_epoch and
evaluate, imported from /root/reference) on small seeded synthetic loaders: eager-sequential accumulation
(B_seq < B), a last batch that has to be shrunk, the M >= N shortcut with zero-padded rows, shuffling.

The fixtures hold what the loops hand to the log writer at every step (task losses, predictions, labels),
the learning rate, and a checksum of the weights after training - data only.  See tests/test_loops_golden.py.

    python tools/gen_golden_loops.py
"""

import json
import os
import sys

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np
import torch
from torch import nn

from ips_amd import synth
from tools.refimport import import_reference, import_reference_training

GOLDEN = os.path.join(REPO, "tests", "golden")

_OPT = dict(n_epoch=2, n_epoch_warmup=1, lr=1e-3, wd=0.1, track_efficiency=False, track_epoch=0)

# name -> (conf, number of loader items, weight seed, data seed, torch seed)
CASES = {
    # B_seq = 1 < B = 4: four ips() calls per step; 6 items -> a full step and a shrunk one
    "loop_mnist_seq": (synth.mnist_conf(N=64, M=8, I=16, B=4, B_seq=1, shuffle=True, **_OPT), 6, 21, 31, 5),
    # B_seq = B = 2, N = 12 <= M = 16: ips() shortcut, buffer rows [12, 16) stay zero; 3 items
    "loop_traffic_short": (synth.traffic_conf(N=12, M=16, I=8, patch=32, B=2, B_seq=2, n_res_blocks=2, D=128, D_k=16,
                                              D_v=16, D_inner=256, **_OPT), 3, 22, 32, 6),
    # no dropout anywhere: every training step is deterministic, so a GPU run can be held to the losses of every step
    # and to the trained weights (tests/test_loops_golden.py::test_training_on_the_gpu_follows_the_reference)
    "loop_mnist_nodrop": (synth.mnist_conf(N=64, M=8, I=16, B=4, B_seq=2, shuffle=False, attn_dropout=0.0, dropout=0.0,
                                           **_OPT), 4, 24, 34, 8),
    # features, sigmoid / auc task, B_seq = 1, 5 items -> 4 + 1
    "loop_cam_seq": (synth.camelyon_conf(N=96, M=16, I=32, B=4, B_seq=1, n_chan_in=256, D=128, D_k=16, D_v=16,
                                         D_inner=256, **_OPT), 5, 23, 33, 7),
}


class Recorder:
    def __init__(self):
        self.steps = []

    def update(self, losses, preds, labels):
        self.steps.append((dict(losses), {k: np.array(v) for k, v in preds.items()},
                           {k: np.array(v) for k, v in labels.items()}))


def pack(prefix, rec, conf, out):
    out[prefix + "_n_step"] = len(rec.steps)
    for s, (losses, preds, labels) in enumerate(rec.steps):
        for task in conf.tasks.values():
            t = task['name']
            out["%s_%d_loss_%s" % (prefix, s, t)] = np.float64(losses[t])
            out["%s_%d_pred_%s" % (prefix, s, t)] = preds[t]
            out["%s_%d_label_%s" % (prefix, s, t)] = labels[t]


def run_case(name, ref_ips, ref_loop):
    conf, n_item, wseed, dseed, tseed = CASES[name]
    loader = synth.make_loader(conf, n_item, seed=dseed)
    net = ref_ips.IPSNet(torch.device("cpu"), conf)
    synth.fill_weights(net, wseed)
    crit = {t['name']: (nn.NLLLoss() if t['act_fn'] == 'softmax' else nn.BCELoss()) for t in conf.tasks.values()}
    opt = torch.optim.AdamW(net.parameters(), lr=0, weight_decay=conf.wd)
    torch.manual_seed(tseed)
    out = dict(conf=json.dumps(conf.__dict__), n_item=n_item, weight_seed=wseed, data_seed=dseed, torch_seed=tseed)
    dev = torch.device("cpu")
    ev0, tr0, ev, tr1 = Recorder(), Recorder(), Recorder(), Recorder()
    ref_loop.evaluate(net, crit, loader, dev, ev0, conf)           # seeded weights, before any training step
    ref_loop.train_one_epoch(net, crit, loader, opt, dev, 0, tr0, conf)
    out["lr_after_epoch0"] = np.float64(opt.param_groups[0]['lr'])
    ref_loop.evaluate(net, crit, loader, dev, ev, conf)
    ref_loop.train_one_epoch(net, crit, loader, opt, dev, 1, tr1, conf)
    out["lr_after_epoch1"] = np.float64(opt.param_groups[0]['lr'])
    pack("eval0", ev0, conf, out)
    pack("train0", tr0, conf, out)
    pack("eval", ev, conf, out)
    pack("train1", tr1, conf, out)
    sd = net.state_dict()
    out["state_checksum"] = np.float64(sum(v.double().abs().sum().item() for k, v in sd.items()
                                           if not k.endswith("num_batches_tracked")))
    out["q_after"] = sd["transf.crs_attn.q"].numpy()
    np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), **out)
    print("{:20s} train steps {}+{}  eval steps {}  checksum {:.6f}".format(
        name, len(tr0.steps), len(tr1.steps), len(ev.steps), float(out["state_checksum"])))


def main():
    ref_ips, _, _ = import_reference()
    ref_loop = import_reference_training()
    for n in sys.argv[1:] or list(CASES):
        run_case(n, ref_ips, ref_loop)


if __name__ == "__main__":
    main()
