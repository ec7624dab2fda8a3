"""The workloads bench.py times, pinned to the REFERENCE: tests/golden/bench_<name>.npz hold the memory indices the
reference's loop (/root/reference/architecture/ips_net.py:218-241) produced after every iteration on exactly the
inputs bench.py uses (weights seed 7, patches seed 21; tools/gen_golden_bench.py ran the imported reference in the
build container).

What must hold on the GPU:
  * the final selection ``net.last_mem_idx`` equals the reference's, index for index, on every image - through the
    same ``net.ips(x)`` call bench.py times (overlapped pipeline and all);
  * replaying the loop one iteration at a time (``ipsx_scan_range``, each iteration started from the reference's
    memory of the previous one): the same SET of patches whenever the reference's top-M boundary gap of that
    iteration (``rel_gap``: M-th vs (M+1)-th score) is above 1e-5, and the same ORDER whenever no two neighbouring
    scores of its sorted top M + 1 are closer than 1e-5 (``order_gap``); iterations below those floors are inside
    the reference's own noise (its oneDNN convolutions move by ~1e-6 with the batch size) and are only
    counted;
  * final predictions within 1e-4.
"""

import json
import math
import os

import numpy as np
import pytest
import torch

from ips_amd import synth
from tests.util import GOLDEN_DIR

WORKLOADS = ["mnist", "mnist3000", "native50", "traffic", "cam", "cam_native"]
GAP_FLOOR = 1e-5
# relative score gap below which a host-made positional table that differs in the last ulp of its frequency vector (times
# positions up to N = 10,000: entries move by ~6e-4) may legitimately change a selection
TABLE_FLOOR = 1e-4
# cam_native ranks 10,000 candidates whose scores all lie within a binade or two of 1e-4: in EVERY iteration some
# neighbours of the reference's sorted top M + 1 are bit-equal (order_gap = 0), so the ORDER inside the memory is the
# reference's own noise at a percent of the positions, while the SET has a clear boundary (rel_gap 2.5e-5 ... 4e-4 in
# six of seven iterations).  There the final selection is held to the reference as a set, and position for position
# on at least this share of the slots:
SAME_SLOT_SHARE = {"cam_native": 0.98}


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, "bench_%s.npz" % name))
    conf = synth.Conf(**json.loads(str(z["conf"])))
    return z, conf


@pytest.mark.parametrize("name", WORKLOADS)
def test_fixture_is_what_bench_py_runs(name):
    """CPU: the fixture's configuration / seeds are the ones synth.bench_workload (and with it bench.py) uses."""
    z, conf = load(name)
    want, B = synth.bench_workload(name)
    assert conf.__dict__ == want.__dict__ and int(z["B"]) == B
    assert (int(z["weight_seed"]), int(z["patch_seed"])) == (7, 21)
    n_iter = math.ceil((conf.N - conf.M) / conf.I)
    assert z["trace_idx"].shape == (B, n_iter, conf.M) and z["rel_gap"].shape == (B, n_iter)
    idx = z["trace_idx"].astype(np.int64)
    assert idx.max() < conf.N
    for b in range(B):                               # a memory never holds a patch twice
        assert all(len(set(row)) == conf.M for row in idx[b, ::max(1, n_iter // 7)])


def test_oracle_follows_the_reference_on_cam_native():
    """CPU: the oracle on the reference's shipped CAMELYON memory / chunk sizes (M = I = 5000: 10,000 candidates per
    iteration) keeps the reference's SET of patches in every iteration and its patch in >= 98 % of the slots (the rest:
    neighbours with bit-equal or noise-level scores, see SAME_SLOT_SHARE)."""
    from ips_amd.architecture import IPSNet
    from oracle.oracle import Oracle
    z, conf = load("cam_native")
    net = synth.fill_weights(IPSNet(torch.device("cpu"), conf), 7).eval()
    x = synth.make_patches(conf, int(z["B"]), seed=21)
    out = Oracle(net).ips(x.numpy(), None, aten_ties=True)
    want, trace = z["trace_idx"].astype(np.int64), out["trace_idx"]
    assert trace.shape == want.shape
    for it in range(want.shape[1]):
        if z["rel_gap"][0, it] > GAP_FLOOR:
            assert np.array_equal(np.sort(trace[0, it]), np.sort(want[0, it])), it
    assert np.array_equal(np.sort(trace[0, -1]), np.sort(want[0, -1]))
    assert float((trace[:, -1] == want[:, -1]).mean()) >= SAME_SLOT_SHARE["cam_native"]


def hip_trace(net, x, want):
    """Memory indices after every iteration on the HIP path: encode -> logits once, then one ipsx_scan_range per
    iteration, each iteration STARTING FROM THE REFERENCE'S memory of the previous one (``want`` (B, n_iter, M)) so
    that a near-tie resolved the other way in one iteration cannot make later, clear iterations look wrong.
    (B, n_iter, M) int64 on the host."""
    from ips_amd import hip
    B, N = x.shape[:2]
    ca = net.transf.crs_attn
    M, I = net.M, net.I
    emb = net._embed(x.reshape(-1, *x.shape[2:])).view(B, N, -1)
    pos = net.pos_enc.expand(B, -1, -1) if net.use_pos else None
    lg = hip.logits(emb, pos, ca.folded_query(), ca.H * ca.n_token)
    mem_idx = torch.empty((B, M), dtype=torch.int64, device=x.device)
    tie = torch.zeros((B,), dtype=torch.int32, device=x.device)
    out = []
    for it in range(math.ceil((N - M) / I)):
        if it > 0:
            mem_idx.copy_(torch.from_numpy(want[:, it - 1]))
        hip.scan_range(lg, M, I, ca.H, ca.n_token, it, it + 1, mem_idx, tie)
        out.append(mem_idx.clone())
    return torch.stack(out, 1).cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("name", WORKLOADS)
def test_bench_workload_selects_the_reference_indices(name):
    from ips_amd.architecture import IPSNet
    z, conf = load(name)
    B = int(z["B"])
    dev = torch.device("cuda:0")
    net = synth.fill_weights(IPSNet(dev, conf), 7).to(dev).eval()
    own_table = synth.use_fixture_pos_table(net, z)        # "identical inputs" includes the host-made positional table
    x = synth.make_patches(conf, B, seed=21).to(dev)
    want = z["trace_idx"].astype(np.int64)
    gap, ogap = z["rel_gap"], z["order_gap"]

    mem_patch, mem_pos = net.ips(x)                                   # the call bench.py times
    got = net.last_mem_idx.cpu().numpy()
    if name in SAME_SLOT_SHARE:
        assert np.array_equal(np.sort(got, -1), np.sort(want[:, -1], -1)), "final selection keeps other patches"
        share = float((got == want[:, -1]).mean())
        print("%s: %.2f %% of the memory slots hold the reference's patch (same set on every image)" % (name, 100 * share))
        assert share >= SAME_SLOT_SHARE[name]
    else:
        assert np.array_equal(got, want[:, -1]), "final selection differs on images %s" % (
            np.nonzero((got != want[:, -1]).any(1))[0].tolist(),)
    with torch.no_grad():
        preds = net(mem_patch, mem_pos)
    for k in preds:
        assert np.abs(preds[k].cpu().numpy() - z["pred_" + k]).max() <= 1e-4, k
    s = mem_patch.double().sum(dim=tuple(range(2, mem_patch.dim()))).cpu().numpy()
    if name in SAME_SLOT_SHARE:
        assert np.allclose(np.sort(s, -1), np.sort(z["mem_patch_sum"], -1), rtol=1e-12, atol=1e-9)
    else:
        assert np.allclose(s, z["mem_patch_sum"], rtol=1e-12, atol=1e-9)

    trace = hip_trace(net, x, want)
    assert trace.shape == want.shape
    same_seq = (trace == want).all(-1)                               # (B, n_iter)
    same_set = (np.sort(trace, -1) == np.sort(want, -1)).all(-1)
    bad_set = (gap > GAP_FLOOR) & ~same_set
    bad_seq = (ogap > GAP_FLOOR) & (gap > GAP_FLOOR) & ~same_seq
    assert not bad_set.any(), "other patches kept at a clear boundary: %s" % (
        [(b, i, float(gap[b, i])) for b, i in np.argwhere(bad_set)[:8]],)
    assert not bad_seq.any(), "other order with clearly separated scores: %s" % (
        [(b, i, float(ogap[b, i])) for b, i in np.argwhere(bad_seq)[:8]],)
    # BOTH tables: the product builds the positional table on whatever host it runs on (as the reference does), and the
    # table's last ulp differs between CPU models.  Everything above ran with the recording machine's table; with THIS
    # host's own table the final selection must still be the reference's on every image whose run never came within
    # TABLE_FLOOR of a tie (the table moves scores by ~N * 2^-24 relative) - a workload that only passes with the
    # recorded table fails here instead of hiding behind it
    if conf.use_pos:
        clear = (gap.min(1) > TABLE_FLOOR) & (ogap.min(1) > TABLE_FLOOR)
        if own_table:
            got_own = got
        else:
            net_own = synth.fill_weights(IPSNet(dev, conf), 7).to(dev).eval()
            net_own.ips(x)
            got_own = net_own.last_mem_idx.cpu().numpy()
        same_own = (got_own == want[:, -1]).all(1)
        assert same_own[clear].all(), "with this host's own positional table images %s differ although no iteration came within %g of a tie" % (
            np.nonzero(clear & ~same_own)[0].tolist(), TABLE_FLOOR)
        print("%s: host-made positional table %s the fixture machine's; with it %d of %d images select the reference's patches in "
              "its order (%d images have an iteration within %g of a tie and are not held to that)"
              % (name, "equals" if own_table else "DIFFERS from", int(same_own.sum()), same_own.size, int((~clear).sum()), TABLE_FLOOR))
    # below the floors: report, do not judge (the reference itself is not reproducible there)
    print("%s: this host's positional table %s the fixture machine's" % (name, "equals" if own_table else "DIFFERS from"))
    print("%s: %d iterations; boundary gap <= %.0e in %d (%d of them keep other patches); neighbours closer than that "
          "in %d (%d of them in another order)" % (name, gap.size, GAP_FLOOR, int((gap <= GAP_FLOOR).sum()),
                                                   int(((gap <= GAP_FLOOR) & ~same_set).sum()),
                                                   int((ogap <= GAP_FLOOR).sum()), int(((ogap <= GAP_FLOOR) & ~same_seq).sum())))


@pytest.mark.gpu
def test_configs2_with_this_hosts_own_positional_table():
    """BASELINE configs[2] (10,000 patches per image) with the positional table THIS host builds (the product never
    swaps tables; the other tests install the recording machine's, ``synth.use_fixture_pos_table``, because the table's
    last ulp differs between CPU models and at N = 10,000 that reorders near-identical blank patches).  What can be
    held here without the reference: the HIP path equals the oracle's loop on the same embeddings and the same
    host-made table, index for index and iteration by iteration's end; against the recorded reference run the
    differences are counted - none where this host's table equals the recording machine's."""
    from ips_amd.architecture import IPSNet
    from oracle.oracle import Oracle
    z, conf = load("mnist3000")
    dev = torch.device("cuda:0")
    net = synth.fill_weights(IPSNet(dev, conf), 7).to(dev).eval()
    B = 4
    x = synth.make_patches(conf, int(z["B"]), seed=21)[:B].to(dev)
    same_table = bool(torch.equal(synth.pos_table_from(z["pos_freq"], conf.N).unsqueeze(0).to(dev), net.pos_enc))
    net.ips(x)                                                        # the product call, host-made table
    got = net.last_mem_idx.cpu().numpy()
    with torch.no_grad():
        emb = net._embed(x.reshape(-1, *x.shape[2:])).view(B, conf.N, -1).cpu().numpy()
    cpu = synth.fill_weights(IPSNet(torch.device("cpu"), conf), 7).eval()
    assert torch.equal(cpu.pos_enc, net.pos_enc.cpu())                # built by the same ATen ops on the same host
    want = Oracle(cpu).scan(emb, cpu.pos_enc.numpy(), aten_ties=True)["mem_idx"]
    assert np.array_equal(got, want), "HIP loop differs from the oracle's on images %s" % np.nonzero((got != want).any(1))[0].tolist()
    ref = z["trace_idx"][:B, -1].astype(np.int64)
    equal = (got == ref).all(1)
    print("mnist3000, host-made positional table (%s the recording machine's): %d of %d images select the reference's "
          "patches in the reference's order, %.2f %% of all slots" % ("equal to" if same_table else "DIFFERENT from",
                                                                    int(equal.sum()), B, 100 * float((got == ref).mean())))
    if same_table:
        assert equal.all()
