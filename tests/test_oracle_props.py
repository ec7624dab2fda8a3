"""Properties of the oracle's primitives (they define the bits the HIP kernels reproduce)."""

import math

import numpy as np
import torch

from oracle import oracle as orc


def test_det_expf_is_accurate_and_monotone():
    xs = np.concatenate([np.linspace(-103.9, 88.7, 20001), -np.logspace(-8, 2, 500)]).astype(np.float32)
    ys = np.array([orc.expf(x) for x in xs], dtype=np.float64)
    ref = np.exp(xs.astype(np.float64))
    ok = ref > 1e-37                     # normal range: <= 2 ulp
    rel = np.abs(ys[ok] - ref[ok]) / ref[ok]
    assert rel.max() < 2.5e-7
    assert orc.expf(0.0) == 1.0 and orc.expf(-200.0) == 0.0 and math.isinf(orc.expf(89.0))
    assert math.isnan(orc.expf(float("nan")))
    s = np.sort(xs)
    assert np.all(np.diff([orc.expf(x) for x in s[::50]]) >= 0)


def test_wave_sum64_order():
    rng = np.random.default_rng(0)
    x = rng.standard_normal(1000).astype(np.float32)
    part = np.zeros(64, dtype=np.float32)
    for j in range(64):
        s = np.float32(0)
        for i in range(j, 1000, 64):
            s = np.float32(s + x[i])
        part[j] = s
    off = 32
    while off:
        part = (part + part[np.arange(64) ^ off]).astype(np.float32)
        off >>= 1
    assert orc.wave_sum64(x) == float(part[0])


def test_topm_equals_torch_when_scores_are_distinct():
    rng = np.random.default_rng(1)
    for L, M in ((128, 64), (48, 16), (512, 256), (10000, 5)):
        s = rng.permutation(L).astype(np.float32) / L
        top, tie = orc.topm(s, M)
        assert tie == 0
        assert np.array_equal(top, torch.topk(torch.from_numpy(s), M)[1].numpy())
        top2, _ = orc.topm(s, M, aten_ties=True)
        assert np.array_equal(top2, top)


def test_topm_aten_restatement_matches_torch_under_ties():
    # blank patches without positional encoding tie exactly; torch's order is then
    # libstdc++'s nth_element/partial_sort order, which orc_topm_aten restates
    for L, M in ((128, 64), (48, 16), (512, 256), (200, 2)):
        s = np.zeros(L, dtype=np.float32)
        s[::7] = 0.5
        top, _ = orc.topm(s, M, aten_ties=True)
        assert np.array_equal(top, torch.topk(torch.from_numpy(s), M)[1].numpy())
        canon, tie = orc.topm(s, M)
        assert sorted(s[canon], reverse=True) == sorted(s[top], reverse=True)


def test_scores_rows_sum_to_one():
    from tests.util import Golden
    g = Golden("mnist_mini")
    o = orc.Oracle(g.net("cpu"))
    x = np.random.default_rng(2).standard_normal((32, g.conf.D)).astype(np.float32)
    sc, attn = o.scores(x, want_attn=True)
    assert abs(sc.sum() - 1.0) < 1e-5
    assert np.allclose(attn.sum(-1), 1.0, atol=1e-5)
    ref = g.net("cpu").transf.get_scores(torch.from_numpy(x)[None])[0].detach().numpy()
    assert np.abs(ref - sc).max() < 1e-6


def test_device_tie_order_restatement_equals_libstdcxx():
    """csrc/ipsx_stdorder.h (nth_element / sort / partial_sort as the kernels replay them under score ties), compiled
    for the host, against std:: called the way ATen's CPU top-k calls it - tie-heavy random inputs, NaNs, the
    depth-limit (heap) fallbacks."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "check_stdorder"])
    out = subprocess.run([os.path.join(root, "oracle", "check_stdorder"), "40000"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "mismatching cases 0" in out.stdout, out.stdout + out.stderr


def test_device_tie_order_restatement_is_clean_under_sanitizers():
    """The same restatement under AddressSanitizer + UBSan (host build): no out-of-bounds access in the hole-sifting
    heap routines, the unguarded partition / insertion loops or the explicit stack."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "oracle"), "check_stdorder_asan"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    out = subprocess.run([os.path.join(root, "oracle", "check_stdorder_asan"), "6000"], capture_output=True, text=True,
                         timeout=600, env=env)
    assert out.returncode == 0 and "mismatching cases 0" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_projector_stays_accurate_where_the_mean_dwarfs_the_spread():
    """Advisor r05 (medium).  The projector folds its LayerNorm into the Linear's epilogue (one pass over every feature row);
    E[x^2] - mean^2 and acc - mean * colsum both cancel when a row's mean is large against its spread.  Rows with
    mean^2 / var > 16 are therefore centred like nn.LayerNorm (reference architecture/ips_net.py:56) does it: moments
    around the mean, mean refined by E[x - mean], Linear on x - mean.  Against float64 LayerNorm -> Linear -> BatchNorm1d ->
    ReLU at mean / std from 0 to 1000 and on constant rows, through the WHOLE projector (not only the moments)."""
    import ctypes as C
    import torch
    from ips_amd import synth
    from ips_amd.architecture import IPSNet
    from oracle.oracle import Oracle
    conf = synth.camelyon_conf(N=64, M=8, I=8)
    net = synth.fill_weights(IPSNet(torch.device("cpu"), conf), 3).eval()
    oracle = Oracle(net)
    enc64 = synth.fill_weights(IPSNet(torch.device("cpu"), conf), 3).eval().encoder.double()
    rng = np.random.default_rng(5)
    worst = {}
    for ratio in (0.0, 1.0, 3.5, 4.0, 4.5, 8.0, 30.0, 100.0, 1000.0):
        x = (rng.standard_normal((48, conf.n_chan_in)) + ratio).astype(np.float32)
        got = oracle.encode(x)
        with torch.no_grad():
            want = enc64(torch.from_numpy(x).double()).numpy()
            ref32 = net.encoder(torch.from_numpy(x)).numpy()              # the reference's own fp32 arithmetic
        worst[ratio] = (float(np.abs(got - want).max()), float(np.abs(ref32 - want).max()))
        st = np.empty((48, 2), dtype=np.float32)
        orc.lib().orc_projector_moments(orc._f(x)[1], C.c_int64(48), conf.n_chan_in, C.c_float(1e-5), st.ctypes.data_as(orc.f32p))
        assert (st[:, 1] < 0).all() if ratio >= 4.5 else (st[:, 1] > 0).all() if ratio <= 3.5 else True
        rstd64 = 1.0 / np.sqrt(x.astype(np.float64).var(1) + 1e-5)
        assert (np.abs(np.abs(st[:, 1]) - rstd64) / rstd64).max() <= 2e-5
    print({k: "%.1e (torch fp32: %.1e)" % v for k, v in worst.items()})
    # (at mean / std = 1000 the row itself carries 1e-4 std of rounding per element: the reference's own fp32 path is no better)
    assert all(v[0] <= 6e-5 for k, v in worst.items() if k <= 100.0) and worst[1000.0][0] <= max(5e-4, worst[1000.0][1])
    # constant rows: LayerNorm gives exact zeros, so every such row is relu(BatchNorm(bias)) - whatever the constant
    x = np.empty((4, conf.n_chan_in), dtype=np.float32)
    x[0], x[1], x[2], x[3] = 3.7, -1e-3, 65504.0, 0.0
    got = oracle.encode(x)
    with torch.no_grad():
        want = enc64(torch.from_numpy(x).double()).numpy()
    assert np.abs(got - want).max() <= 1e-6 and np.array_equal(got[0], got[3]) and np.array_equal(got[0], got[2])


def test_loop_tie_rule_sees_duplicates_across_a_colliding_row():
    """Advisor r05 (low), round 6: the loops replay torch.topk's order when two candidates of ONE RUN of equal scores have
    bit-identical logit rows - not only when they are neighbours.  A, B, C with equal scores, A and C duplicates, B a
    different row that collides with them: no neighbouring pair is identical, the run is still a structural tie.  Also a run
    that starts inside the first M ranks and continues beyond the boundary; and a run of colliding DIFFERENT rows, which
    keeps the canonical order."""
    import ctypes as C
    L, R, M = 12, 8, 5
    rng = np.random.default_rng(3)
    lg = rng.standard_normal((L, R)).astype(np.float32)
    base = np.linspace(1.0, 0.1, L).astype(np.float32)

    def run(scores, rows):
        top = np.empty(M, dtype=np.int64)
        orc.lib().orc_topm_loop(orc._f(scores)[1], orc._f(rows)[1], L, R, M, top.ctypes.data_as(orc.i64p), None)
        return top

    def canonical(scores):
        return np.array(sorted(range(L), key=lambda i: (-scores[i], i))[:M])

    def torch_order(scores):
        return torch.topk(torch.from_numpy(scores), M)[1].numpy()

    # (1) A, B, C at ranks 1, 2, 3: equal scores, rows A == C != B
    s = base.copy(); s[[4, 6, 9]] = s[1]                  # candidates 1, 4, 6, 9 share a score; make 1 unique again
    s[1] = 2.0
    rows = lg.copy(); rows[9] = rows[4]                   # 4 and 9 duplicates, 6 differs
    assert not np.array_equal(rows[4], rows[6])
    got = run(s, rows)
    assert np.array_equal(got, torch_order(s))
    # without the duplicate: three different rows collide -> canonical order, whatever torch does
    assert np.array_equal(run(s, lg), canonical(s))
    # (2) a run that reaches across the boundary: ranks M-1, M, M+1 tie, the duplicate of rank M-1 sits at rank M+1
    s = base.copy(); s[[7, 10]] = s[4]                    # canonical ranks: 0,1,2,3,4 | 7,10 tie with 4 -> ranks 4,5,6
    rows = lg.copy(); rows[10] = rows[4]
    assert np.array_equal(run(s, rows), torch_order(s))
    assert np.array_equal(run(s, lg), canonical(s))
    # (3) duplicates that both lie beyond the boundary change nothing
    s = base.copy(); s[11] = s[9]
    rows = lg.copy(); rows[11] = rows[9]
    assert np.array_equal(run(s, rows), canonical(s))
