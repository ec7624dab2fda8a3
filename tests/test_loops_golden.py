"""The caller loops (ips_amd/training/iterative.py) against fixtures produced by the reference's own
training/iterative.py (tools/gen_golden_loops.py): same losses, predictions, labels, learning rate and
trained weights on CPU; on the GPU the evaluation pass goes through the HIP path (with the embeddings of
the winners reused instead of a second encoder pass) and must agree within the north-star tolerance."""

import json
import os

import numpy as np
import pytest
import torch
from torch import nn

from ips_amd import synth
from ips_amd.architecture.ips_net import IPSNet
from ips_amd.training import iterative as loops
from ips_amd.utils.utils import Logger, adjust_learning_rate

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["loop_mnist_seq", "loop_traffic_short", "loop_cam_seq", "loop_mnist_nodrop"]


class Recorder:
    def __init__(self):
        self.steps = []

    def update(self, losses, preds, labels):
        self.steps.append((losses, preds, labels))


def _setup(case, device):
    z = np.load(os.path.join(GOLDEN, case + ".npz"), allow_pickle=False)
    conf = synth.Conf(**json.loads(str(z["conf"])))
    net = IPSNet(torch.device(device), conf).to(device)
    synth.fill_weights(net, int(z["weight_seed"]))
    loader = synth.make_loader(conf, int(z["n_item"]), seed=int(z["data_seed"]))
    crit = {t['name']: (nn.NLLLoss() if t['act_fn'] == 'softmax' else nn.BCELoss()) for t in conf.tasks.values()}
    opt = torch.optim.AdamW(net.parameters(), lr=0, weight_decay=conf.wd)
    torch.manual_seed(int(z["torch_seed"]))
    return z, conf, net, loader, crit, opt


def _check(z, prefix, rec, conf, tol):
    assert len(rec.steps) == int(z[prefix + "_n_step"])
    for s, (losses, preds, labels) in enumerate(rec.steps):
        for task in conf.tasks.values():
            t = task['name']
            want_pred = z["%s_%d_pred_%s" % (prefix, s, t)]
            assert preds[t].shape == want_pred.shape and preds[t].dtype == want_pred.dtype
            np.testing.assert_allclose(preds[t], want_pred, rtol=0, atol=tol)
            np.testing.assert_allclose(losses[t], float(z["%s_%d_loss_%s" % (prefix, s, t)]), rtol=tol * 10, atol=tol * 10)
            want_label = z["%s_%d_label_%s" % (prefix, s, t)]
            assert labels[t].dtype == want_label.dtype and np.array_equal(labels[t], want_label)


@pytest.mark.parametrize("case", CASES)
def test_loops_reproduce_the_reference_on_cpu(case):
    z, conf, net, loader, crit, opt = _setup(case, "cpu")
    dev = torch.device("cpu")
    ev0, tr0, ev, tr1 = Recorder(), Recorder(), Recorder(), Recorder()
    loops.evaluate(net, crit, loader, dev, ev0, conf)
    loops.train_one_epoch(net, crit, loader, opt, dev, 0, tr0, conf)
    assert opt.param_groups[0]['lr'] == pytest.approx(float(z["lr_after_epoch0"]), rel=1e-12)
    loops.evaluate(net, crit, loader, dev, ev, conf)
    loops.train_one_epoch(net, crit, loader, opt, dev, 1, tr1, conf)
    assert opt.param_groups[0]['lr'] == pytest.approx(float(z["lr_after_epoch1"]), rel=1e-12)
    _check(z, "eval0", ev0, conf, 1e-6)
    _check(z, "train0", tr0, conf, 1e-6)
    _check(z, "eval", ev, conf, 1e-5)
    _check(z, "train1", tr1, conf, 1e-5)
    sd = net.state_dict()
    checksum = sum(v.double().abs().sum().item() for k, v in sd.items() if not k.endswith("num_batches_tracked"))
    assert checksum == pytest.approx(float(z["state_checksum"]), rel=1e-6)
    np.testing.assert_allclose(sd["transf.crs_attn.q"].numpy(), z["q_after"], rtol=0, atol=1e-5)


def test_evaluate_with_and_without_embedding_reuse_agree_on_cpu():
    z, conf, net, loader, crit, _ = _setup("loop_mnist_seq", "cpu")
    a, b = Recorder(), Recorder()
    state = torch.get_rng_state()
    loops.evaluate(net, crit, loader, torch.device("cpu"), a, conf)

    class NoReuse(type(net)):
        last_mem_emb = property(lambda self: None)
    net.__class__ = NoReuse
    torch.set_rng_state(state)
    loops.evaluate(net, crit, loader, torch.device("cpu"), b, conf)
    for (la, pa, _), (lb, pb, _) in zip(a.steps, b.steps):
        for t in pa:
            np.testing.assert_allclose(pa[t], pb[t], rtol=0, atol=1e-6)


def test_learning_rate_schedule_and_logger():
    opt = torch.optim.AdamW([nn.Parameter(torch.zeros(1))], lr=0)
    loader = synth.ListLoader([None] * 10)
    lrs = []
    for step in range(1, 51):
        adjust_learning_rate(1, 5, 1e-3, opt, loader, step)
        lrs.append(opt.param_groups[0]['lr'])
    assert lrs[0] == pytest.approx(1e-4) and lrs[8] == pytest.approx(9e-4)       # linear warm-up over 10 steps
    assert lrs[9] == pytest.approx(1e-3) and lrs[-1] == pytest.approx(1e-6)      # cosine to max_lr / 1000
    assert all(a >= b for a, b in zip(lrs[9:], lrs[10:]))

    tasks = {'a': {'id': 0, 'name': 'cls', 'act_fn': 'softmax', 'metric': 'accuracy'},
             'b': {'id': 1, 'name': 'multi', 'act_fn': 'sigmoid', 'metric': 'multilabel_accuracy'},
             'c': {'id': 2, 'name': 'bin', 'act_fn': 'sigmoid', 'metric': 'auc'}}
    log = Logger(tasks)
    log.update({'cls': 1.0, 'multi': 2.0, 'bin': 3.0},
               {'cls': np.array([[.1, .9], [.8, .2]]), 'multi': np.array([[.9, .1], [.4, .6]]), 'bin': np.array([.9, .2])},
               {'cls': np.array([1, 1]), 'multi': np.array([[1., 0.], [1., 1.]]), 'bin': np.array([1, 0])})
    log.update({'cls': 3.0, 'multi': 2.0, 'bin': 1.0},
               {'cls': np.array([[.3, .7]]), 'multi': np.array([[.2, .7]]), 'bin': np.array([.5])},
               {'cls': np.array([1]), 'multi': np.array([[0., 1.]]), 'bin': np.array([0])})
    log.compute_metric()
    assert log.losses_epoch['cls'] == [2.0] and log.metrics['cls'] == [pytest.approx(2 / 3)]
    assert log.metrics['multi'] == [pytest.approx(2 / 3)] and log.metrics['bin'] == [1.0]
    assert log.losses_it['cls'] == [] and log.y_preds['bin'] == []
    log.print_stats(0, train=False, lr=0.1)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_evaluate_on_the_hip_path_matches_the_reference(case):
    """evaluate() before any training step: ips() on the HIP path, winners' embeddings reused in forward."""
    z, conf, net, loader, crit, _ = _setup(case, "cuda:0")
    rec = Recorder()
    loops.evaluate(net, crit, loader, torch.device("cuda:0"), rec, conf)
    _check(z, "eval0", rec, conf, 1e-4)


@pytest.mark.gpu
def test_embedding_reuse_is_bit_identical_on_the_hip_path():
    z, conf, net, loader, crit, _ = _setup("loop_mnist_seq", "cuda:0")
    net.eval()
    x = loader.items[0]['input'].to("cuda:0")
    with torch.no_grad():
        torch.manual_seed(3)
        mem_patch, mem_pos = net.ips(x)
        emb = net.last_mem_emb
        assert emb is not None and emb.shape == (x.shape[0], conf.M, conf.D)
        again = net._embed(mem_patch.reshape(-1, *mem_patch.shape[2:])).view_as(emb)
        assert torch.equal(emb, again)
        p0 = net(mem_patch, mem_pos)
        p1 = net(mem_patch, mem_pos, mem_emb=emb)
        assert all(torch.equal(p0[k], p1[k]) for k in p0)


@pytest.mark.gpu
def test_training_epoch_runs_on_the_gpu_and_tracks_the_reference():
    """train_one_epoch on the GPU: ips() on the HIP path, forward/backward on stock ROCm ops.  Dropout masks come
    from the device generator, so here only the step count and finiteness are checked; the deterministic comparison
    is test_training_on_the_gpu_follows_the_reference."""
    z, conf, net, loader, crit, opt = _setup("loop_cam_seq", "cuda:0")
    rec = Recorder()
    loops.train_one_epoch(net, crit, loader, opt, torch.device("cuda:0"), 0, rec, conf)
    assert len(rec.steps) == int(z["train0_n_step"])
    assert all(np.isfinite(v) for losses, _, _ in rec.steps for v in losses.values())


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
def test_training_on_the_gpu_follows_the_reference(fused):
    """Two epochs WITHOUT dropout (fixture loop_mnist_nodrop, recorded from the reference's own train_one_epoch /
    evaluate on CPU): ips() runs on the HIP path between optimizer steps (packed weights and BatchNorm statistics are
    refreshed after every step), forward / backward / AdamW on stock ROCm ops.  The first step sees identical weights:
    its losses and predictions match the reference to 1e-4.  Later steps inherit the rounding differences of MIOpen's
    convolutions through AdamW (an update is +-lr whatever the gradient's size, so a gradient that is zero up to rounding
    can move a weight by 2 lr the other way): losses within 2e-3, predictions within 5e-3, trained queries within a few
    learning rates, and the evaluation pass between the epochs within 5e-3.  ``fused``: torch's fused AdamW kernel (what
    ips_amd/main.py uses on a GPU) - the same update as one kernel over all parameter tensors.  It does NOT bump the
    parameters' ``_version`` counters, which the packed-weight caches of the HIP path are keyed on: without the optimizer
    step hook of ips_amd/hip.py, ips() would keep selecting with the weights of step 0 (this test's fourth step was 10 %
    off the reference before the hook existed)."""
    z, conf, net, loader, crit, opt = _setup("loop_mnist_nodrop", "cuda:0")
    if fused:          # ips_amd/main.py's choice on a GPU: the same update as one kernel over all parameter tensors
        opt = torch.optim.AdamW(net.parameters(), lr=0, weight_decay=conf.wd, fused=True)
    dev = torch.device("cuda:0")
    ev0, tr0, ev, tr1 = Recorder(), Recorder(), Recorder(), Recorder()
    loops.evaluate(net, crit, loader, dev, ev0, conf)
    loops.train_one_epoch(net, crit, loader, opt, dev, 0, tr0, conf)
    loops.evaluate(net, crit, loader, dev, ev, conf)
    loops.train_one_epoch(net, crit, loader, opt, dev, 1, tr1, conf)
    _check(z, "eval0", ev0, conf, 1e-4)
    assert len(tr0.steps) == int(z["train0_n_step"]) and len(tr1.steps) == int(z["train1_n_step"])
    for prefix, rec in (("train0", tr0), ("train1", tr1)):
        for s_, (losses, preds, _) in enumerate(rec.steps):
            first = prefix == "train0" and s_ == 0
            for task in conf.tasks.values():
                t = task['name']
                want = float(z["%s_%d_loss_%s" % (prefix, s_, t)])
                later = 4e-3 if fused else 2e-3
                assert losses[t] == pytest.approx(want, rel=1e-4 if first else later, abs=1e-4 if first else later), (prefix, s_, t)
                np.testing.assert_allclose(preds[t], z["%s_%d_pred_%s" % (prefix, s_, t)], rtol=0,
                                           atol=1e-4 if first else (1e-2 if fused else 5e-3))
    _check(z, "eval", ev, conf, 1e-2 if fused else 5e-3)
    q = net.state_dict()["transf.crs_attn.q"].cpu().numpy()
    lr_max = float(conf.lr)
    assert np.abs(q - z["q_after"]).max() <= 4 * 2 * lr_max          # 4 steps, at most 2 lr apart each
    assert np.abs(q - z["q_after"]).mean() <= (0.1 if fused else 0.05) * lr_max      # ... and almost everywhere much closer


@pytest.mark.gpu
def test_hip_graph_step_equals_the_eager_step():
    """conf.hip_graph: forward + losses + backward + AdamW replayed as one HIP graph (training/graphed.py).  Without
    dropout the two runs do the same arithmetic up to MIOpen's choice of convolution algorithm (it may differ under
    capture; Winograd-type kernels are ~1e-4 apart) and AdamW's capturable formula, so losses and weights agree
    closely but not bitwise.  Sharp checks: capturing (three warm-up steps on a side stream) must not train - the
    BatchNorm batch counters and the optimizer's step counters equal the number of real steps."""
    results = []
    for use_graph in (False, True):
        z, conf, net, loader, crit, opt = _setup("loop_mnist_seq", "cuda:0")
        conf.attn_dropout = conf.dropout = 0.0
        conf.hip_graph = use_graph
        conf.shuffle = False
        net = IPSNet(torch.device("cuda:0"), conf).to("cuda:0")
        synth.fill_weights(net, int(z["weight_seed"]))
        opt = torch.optim.AdamW(net.parameters(), lr=0, weight_decay=conf.wd)
        rec = Recorder()
        for epoch in range(2):
            loops.train_one_epoch(net, crit, loader, opt, torch.device("cuda:0"), epoch, rec, conf)
        steps = {float(st['step']) for st in opt.state.values()}
        assert steps == {4.0}, steps
        results.append((rec, {k: v.detach().clone() for k, v in net.state_dict().items()}, float(opt.param_groups[0]['lr'])))
    (ra, sa, lra), (rb, sb, lrb) = results
    assert lra == pytest.approx(lrb, rel=1e-6)
    assert len(ra.steps) == len(rb.steps) == 4          # per epoch: one full batch (graph) and one shrunk batch (eager)
    for k, ((la, pa, _), (lb, pb, _)) in enumerate(zip(ra.steps, rb.steps)):
        for t in la:
            assert np.isfinite(la[t]) and np.isfinite(lb[t])
            if k == 0:       # later steps see ips() selections made with weights that already differ in the last digits
                assert la[t] == pytest.approx(lb[t], rel=5e-3, abs=1e-4)
                np.testing.assert_allclose(pa[t], pb[t], rtol=0, atol=5e-3)
    for k in sa:
        if not k.endswith("num_batches_tracked"):
            assert torch.allclose(sa[k], sb[k], rtol=0, atol=6e-3), k       # 4 AdamW steps of lr <= 1e-3 each
        else:
            assert torch.equal(sa[k], sb[k]), k


@pytest.mark.gpu
def test_graph_replays_do_not_leave_stale_packed_weights():
    """A HIP-graph replay moves parameters and BatchNorm statistics without bumping ``_version`` or changing storage:
    the caches of the HIP path (EncoderPlan's packed weights, the folded query) are keyed on exactly those, so
    GraphedStep must invalidate them (hip.weights_changed).  Three consecutive full-batch graph steps; after each one
    ``ips()`` with the net's caches must equal ``ips()`` of a fresh net loaded with the current weights - embeddings,
    logits operand and selected indices."""
    from ips_amd import hip
    from ips_amd.training.graphed import GraphedStep
    dev = torch.device("cuda:0")
    conf = synth.mnist_conf(N=200, M=16, I=16, B=4, B_seq=4, attn_dropout=0.0, dropout=0.0)
    conf.wd = 0.1
    net = synth.fill_weights(IPSNet(dev, conf), 3).to(dev)
    crit = {t['name']: (nn.NLLLoss() if t['act_fn'] == 'softmax' else nn.BCELoss()) for t in conf.tasks.values()}
    opt = torch.optim.AdamW(net.parameters(), lr=0.05, weight_decay=conf.wd)       # large steps: stale weights would show
    step = GraphedStep(net, crit, opt, conf)
    item = synth.make_loader(conf, 1, seed=5).items[0]
    x = item['input'].to(dev)
    labels = {t['name']: item[t['name']].to(dev) for t in conf.tasks.values()}
    net.train()
    first_idx = None
    for k in range(3):
        mem_patch, mem_pos = net.ips(x)
        if first_idx is None:
            first_idx = net.last_mem_idx.clone()
        emb_cached = net.last_mem_emb.clone()
        idx_cached = net.last_mem_idx.clone()
        fresh = IPSNet(dev, conf).to(dev)
        fresh.load_state_dict(net.state_dict())
        fresh.train()
        fresh.ips(x)
        assert torch.equal(idx_cached, fresh.last_mem_idx), "step %d: stale weights selected other patches" % k
        assert torch.equal(emb_cached, fresh.last_mem_emb), "step %d: stale packed weights" % k
        assert torch.equal(net.transf.crs_attn.folded_query(), fresh.transf.crs_attn.folded_query()), k
        step(mem_patch, mem_pos, labels)
    net.ips(x)
    assert not torch.equal(net.last_mem_emb, emb_cached)          # the steps did move the encoder


def test_every_optimizer_step_invalidates_the_packed_weights():
    """Fused optimizers update parameters without bumping ``_version`` (the key of EncoderPlan / folded-query caches): the
    global optimizer-step hook of ips_amd/hip.py must advance the weights generation for ANY optimizer."""
    from ips_amd import hip
    for kw in (dict(), dict(fused=True), dict(foreach=True)):
        p = nn.Parameter(torch.randn(7))
        p.grad = torch.randn(7)
        opt = torch.optim.AdamW([p], lr=1e-3, **kw)
        before = hip.weights_generation()
        opt.step()
        assert hip.weights_generation() > before, kw
    p = nn.Parameter(torch.randn(7))
    p.grad = torch.randn(7)
    before = hip.weights_generation()
    torch.optim.SGD([p], lr=0.1).step()
    assert hip.weights_generation() > before
