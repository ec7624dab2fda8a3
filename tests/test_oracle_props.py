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
