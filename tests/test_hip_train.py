"""GPU: the fused training-mode encoder path (ips_amd/training/fused_encoder.py, csrc/bn_train.hip) against the stock
ROCm ops it replaces - the gradient check VERDICT item 9 asks for.  Reference: training/iterative.py:158-163,
architecture/ips_net.py:273 (encoder under net.train() with autograd)."""
import copy

import pytest
import torch
import torch.nn.functional as F

from ips_amd import hip, hip_train, synth
from ips_amd.architecture import IPSNet
from ips_amd.training import fused_encoder

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach(), b.detach()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("P,ci,co,k,s,H", [(16, 64, 64, 3, 1, 8), (33, 64, 128, 3, 2, 8), (7, 64, 128, 1, 2, 8), (40, 128, 128, 3, 1, 4),
                                             (5, 128, 256, 3, 2, 13), (3, 256, 256, 3, 1, 7), (1024, 64, 64, 3, 1, 8), (2, 512, 512, 3, 1, 2),
                                             (9, 64, 64, 3, 1, 13), (4, 64, 64, 1, 1, 1), (37, 1, 64, 7, 2, 32), (1024, 1, 64, 7, 2, 32), (5, 1, 64, 7, 2, 50),
                                             (1023, 128, 128, 3, 1, 4), (6, 64, 128, 3, 2, 8), (1, 64, 64, 3, 1, 8),
                                             (4, 256, 512, 3, 2, 7), (6, 128, 256, 1, 2, 13), (4, 256, 512, 1, 2, 7), (3, 256, 512, 3, 2, 4)])
def test_conv_train_kernels_match_torch(P, ci, co, k, s, H):
    """The training step's convolutions on libipsx's kernels (training/fused_encoder.py::_Conv): forward and data gradient
    on conv_nhwc_kernel, weight gradient on conv_wgrad_kernel - against float64 autograd of F.conv2d, and bit-identical
    from run to run (the pixel reduction of the weight gradient is split and added in a fixed order)."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(P * 131 + ci + co + k)
    pad = (k - 1) // 2
    conv = torch.nn.Conv2d(ci, co, k, s, pad, bias=False).to(dev)
    assert hip.conv_train_supported(conv)
    x = torch.randn((P, ci, H, H), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((co, ci, k, k), generator=g) / (ci * k * k) ** 0.5).to(dev)
    Ho = (H + 2 * pad - k) // s + 1
    dy = torch.randn((P, co, Ho, Ho), generator=g).to(dev).contiguous(memory_format=torch.channels_last)

    xs, ws = x.double().requires_grad_(), w.double().requires_grad_()
    want = F.conv2d(xs, ws, None, s, pad)
    want.backward(dy.double())

    xh, wh = x.clone().requires_grad_(ci > 1), w.clone().requires_grad_()     # (the stem's input - the patches - needs no gradient)
    y = fused_encoder._Conv.apply(xh, wh, s, pad)
    assert y.is_contiguous(memory_format=torch.channels_last) and tuple(y.shape) == tuple(want.shape)
    y.backward(dy)
    assert _rel(y.double(), want.detach()) < 1e-5
    if ci > 1:
        assert _rel(xh.grad.double(), xs.grad) < 1e-5
    assert _rel(wh.grad.double(), ws.grad) < 1e-5
    again = hip.conv2d_nhwc_wgrad(x, dy, w.shape, s, pad)
    assert torch.equal(again, hip.conv2d_nhwc_wgrad(x, dy, w.shape, s, pad))
    assert torch.equal(again.contiguous(), wh.grad.contiguous())


@pytest.mark.parametrize("P", [1, 5, 1024, 1027])
def test_training_stem_on_the_matrix_cores_equals_the_direct_convolution(P, monkeypatch):
    """The 1-channel 7x7 / 2 stem of the 32-px trunk (csrc/stem_train.hip: the fused trunk's stem without its epilogue)
    against the generic direct convolution it replaces in the training step (``IPSX_TRAIN_STEM_MFMA=0``): the same fma chains in
    the same order - bit for bit - and float64 torch to rounding."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(P)
    x = torch.randn((P, 1, 32, 32), generator=g).to(dev)
    x = x.as_strided(x.shape, (1024, 1, 32, 1))                      # (channels-last strides of a 1-channel tensor)
    w = (torch.randn((64, 1, 7, 7), generator=g) * 0.2).to(dev)
    got = hip.conv2d_nhwc(x, w, 2, 3)
    monkeypatch.setenv("IPSX_TRAIN_STEM_MFMA", "0")
    want = hip.conv2d_nhwc(x, w, 2, 3)
    assert got.shape == (P, 64, 16, 16) and got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got, want)
    ref = F.conv2d(x.double().contiguous(), w.double(), None, 2, 3)
    assert _rel(got.double(), ref) < 1e-6
    monkeypatch.delenv("IPSX_TRAIN_STEM_MFMA")
    # ... and with the BatchNorm statistics off its accumulators (slabs of 4 patches, around a shift)
    shift = (torch.randn(64, generator=g) * 0.3).to(dev)
    y, partial, slabs = hip.conv2d_nhwc(x, w, 2, 3, stats_shift=shift)
    assert torch.equal(y, got) and slabs == (P + 3) // 4 and partial.shape == (slabs, 2, 64)
    d = y.double() - shift.double().view(1, -1, 1, 1)
    assert _rel(partial[:, 0].double().sum(0), d.sum((0, 2, 3))) < 1e-5
    assert _rel(partial[:, 1].double().sum(0), (d * d).sum((0, 2, 3))) < 1e-6


@pytest.mark.parametrize("P", [1, 7, 1026])
def test_strided_projection_data_gradient(P, monkeypatch):
    """The 1x1 / 2 projection's data gradient on the same kernel (one tap; three of four input pixels get zeros)."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(P)
    w = (torch.randn((128, 64, 1, 1), generator=g) * 0.1).to(dev)
    dy = torch.randn((P, 128, 4, 4), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    got = hip.conv2d_nhwc_dgrad(dy, w, 2, 0, (8, 8))
    x = torch.zeros((P, 64, 8, 8), dtype=torch.float64, device=dev, requires_grad=True)
    F.conv2d(x, w.double(), None, 2, 0).backward(dy.double())
    assert _rel(got.double(), x.grad) < 2e-6
    assert float(got[:, :, 1::2].abs().max()) == 0.0 and float(got[:, :, :, 1::2].abs().max()) == 0.0
    monkeypatch.setenv("IPSX_TRAIN_DGRAD_S2", "0")
    assert _rel(got, hip.conv2d_nhwc_dgrad(dy, w, 2, 0, (8, 8))) < 5e-6


@pytest.mark.parametrize("P", [1, 6, 1024, 1027])
def test_strided_data_gradient_by_parity_class(P, monkeypatch):
    """The data gradient of the 32-px trunk's 64 -> 128 channel 3x3 / 2 layer (csrc/dgrad_s2.hip: input pixels by parity
    class, nine taps instead of thirty-six) against float64 autograd, and against the stride-1 convolution over dy spread on a
    zero map that it replaces (``IPSX_TRAIN_DGRAD_S2=0``) to fp32 rounding (another summation order)."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(P)
    w = (torch.randn((128, 64, 3, 3), generator=g) * 0.05).to(dev)
    dy = torch.randn((P, 128, 4, 4), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    got = hip.conv2d_nhwc_dgrad(dy, w, 2, 1, (8, 8))
    assert got.shape == (P, 64, 8, 8) and got.is_contiguous(memory_format=torch.channels_last)
    x = torch.zeros((P, 64, 8, 8), dtype=torch.float64, device=dev, requires_grad=True)
    F.conv2d(x, w.double(), None, 2, 1).backward(dy.double())
    assert _rel(got.double(), x.grad) < 2e-6
    monkeypatch.setenv("IPSX_TRAIN_DGRAD_S2", "0")
    spread = hip.conv2d_nhwc_dgrad(dy, w, 2, 1, (8, 8))
    assert _rel(got, spread) < 5e-6
    # with the weights packed ahead (what the training step hands over), and a channels-last weight tensor
    monkeypatch.delenv("IPSX_TRAIN_DGRAD_S2")
    wcl = w.contiguous(memory_format=torch.channels_last)
    packed = hip.pack_conv_views([(wcl, True)])[0]
    assert torch.equal(hip.conv2d_nhwc_dgrad(dy, wcl, 2, 1, (8, 8), packed=packed), got)


@pytest.mark.parametrize("P,C", [(1, 64), (5, 32), (1024, 64), (3, 128)])
def test_maxpool_backward_is_atens_bit_for_bit(P, C):
    """``maxpool_3x3s2_nhwc`` / ``maxpool_3x3s2_bwd_nhwc`` (the training step's pooling behind the stem) against ATen on
    post-ReLU-like inputs - most windows hold several equal zeros, so WHICH maximum receives the gradient matters: ATen gives
    it to the first one in row-major order, and so must this - values and gradients bit for bit."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(P + C)
    x = torch.relu(torch.randn((P, C, 16, 16), generator=g) - 0.4).to(dev).contiguous(memory_format=torch.channels_last)
    x[0, :, :3, :3] = 0.0                                            # (an all-equal window at the padded corner)
    dy = torch.randn((P, C, 8, 8), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    xs = x.clone().requires_grad_()
    want = F.max_pool2d(xs, 3, 2, 1)
    want.backward(dy)
    got = hip.maxpool_3x3s2_nhwc(x)
    assert torch.equal(got, want.detach())
    dx = hip.maxpool_3x3s2_bwd_nhwc(x, dy)
    assert dx.is_contiguous(memory_format=torch.channels_last) and torch.equal(dx, xs.grad)


def test_batched_weight_packing_equals_the_single_launches():
    """``hip.pack_conv_views`` (ONE launch per 32 weights: what ``fused_encoder.pack_all`` hands the step's convolutions)
    against ``_pack_conv_view`` per weight and direction - the same bits, for contiguous and channels-last weights, the
    1-channel stem, 1x1 layers, and more than 32 jobs."""
    g = torch.Generator(device="cpu").manual_seed(5)
    shapes = [(64, 1, 7, 7), (64, 64, 3, 3), (128, 64, 3, 3), (128, 64, 1, 1), (128, 128, 3, 3), (256, 128, 3, 3), (64, 3, 7, 7)]
    views = []
    for rep in range(3):
        for k, sh in enumerate(shapes):
            w = torch.randn(sh, generator=g).cuda()
            if (k + rep) % 2:
                w = w.contiguous(memory_format=torch.channels_last)
            views.append((w, False))
            if sh[1] >= 64:
                views.append((w, True))
    assert len(views) > 32
    got = hip.pack_conv_views(views)
    for (w, dgrad), p in zip(views, got):
        want = hip._pack_conv_view(w, dgrad)[0]
        assert p.shape == want.shape and torch.equal(p, want), (tuple(w.shape), dgrad)


def test_weight_gradient_of_activations_beyond_one_buffer_is_sliced(monkeypatch):
    """conv2d_nhwc_wgrad on activations of 2 GiB and more (ipsx_conv2d_wgrad_nhwc addresses one 2 GiB buffer per call):
    whole-image slices, added in slice order.  First with the limit lowered so that a small case is cut into four slices
    (against float64 autograd and against the slices added by hand, bit for bit), then at a real size - 2,700 maps of
    56 x 56 x 64 are 2.17 GB each way -, where the stock path and the forward kernel work and this used to raise."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(11)
    P, ci, co, H = 33, 64, 128, 8
    x = torch.randn((P, ci, H, H), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    dy = torch.randn((P, co, 4, 4), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    whole = hip.conv2d_nhwc_wgrad(x, dy, (co, ci, 3, 3), 2, 1)
    monkeypatch.setattr(hip_train, "_WGRAD_MAX_BYTES", 9 * 4 * ci * H * H)          # nine images per call: 9 + 9 + 9 + 6
    sliced = hip.conv2d_nhwc_wgrad(x, dy, (co, ci, 3, 3), 2, 1)
    monkeypatch.undo()
    by_hand = None
    for i0 in range(0, P, 9):
        part = hip.conv2d_nhwc_wgrad(x[i0:i0 + 9], dy[i0:i0 + 9], (co, ci, 3, 3), 2, 1)
        by_hand = part if by_hand is None else by_hand + part
    assert torch.equal(sliced, by_hand)
    xs, ws = x.double(), torch.zeros((co, ci, 3, 3), dtype=torch.float64, device=dev, requires_grad=True)
    F.conv2d(xs, ws, None, 2, 1).backward(dy.double())
    assert _rel(sliced.double(), ws.grad) < 1e-5 and _rel(whole.double(), ws.grad) < 1e-5

    P, C, H = 2700, 64, 56
    x = torch.empty((P, C, H, H), device=dev).contiguous(memory_format=torch.channels_last).normal_(generator=torch.Generator(device=dev).manual_seed(1))
    dy = torch.empty((P, C, H, H), device=dev).contiguous(memory_format=torch.channels_last).normal_(generator=torch.Generator(device=dev).manual_seed(2))
    assert x.numel() * 4 >= 1 << 31
    dw = hip.conv2d_nhwc_wgrad(x, dy, (C, C, 3, 3), 1, 1)
    step = hip._WGRAD_MAX_BYTES // (4 * C * H * H)
    assert 0 < step < P
    by_hand = hip.conv2d_nhwc_wgrad(x[:step], dy[:step], (C, C, 3, 3), 1, 1) + hip.conv2d_nhwc_wgrad(x[step:], dy[step:], (C, C, 3, 3), 1, 1)
    assert torch.equal(dw, by_hand) and bool(torch.isfinite(dw).all())
    # a sample of the entries against float64 on a slice of the images (the full float64 convolution would need 20 GB)
    probe = hip.conv2d_nhwc_wgrad(x[:64], dy[:64], (C, C, 3, 3), 1, 1)
    ws = torch.zeros((C, C, 3, 3), dtype=torch.float64, device=dev, requires_grad=True)
    F.conv2d(x[:64].double(), ws, None, 1, 1).backward(dy[:64].double())
    assert _rel(probe.double(), ws.grad) < 1e-5


@pytest.mark.parametrize("P,C,H,relu,res", [(8, 64, 16, True, False), (5, 64, 8, True, True), (3, 128, 4, False, False),
                                            (7, 256, 2, True, True), (16, 512, 1, True, False), (1024, 64, 16, True, False),
                                            (9, 4, 3, True, True), (33, 8, 5, False, True), (3, 1024, 2, True, False)])
def test_bn_train_kernels_match_torch(P, C, H, relu, res):
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(P * 1000 + C)
    x = (torch.randn((P, C, H, H), generator=g) * 2 + 3).to(dev).contiguous(memory_format=torch.channels_last)
    r = torch.randn((P, C, H, H), generator=g).to(dev).contiguous(memory_format=torch.channels_last) if res else None
    gamma = (torch.rand(C, generator=g) + 0.5).to(dev)
    beta = torch.randn(C, generator=g).to(dev)
    dy = torch.randn((P, C, H, H), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    rm0, rv0 = torch.randn(C, generator=g).to(dev), (torch.rand(C, generator=g) + 0.5).to(dev)

    xs, gs, bs = x.clone().requires_grad_(), gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    rs = r.clone().requires_grad_() if res else None
    rm, rv = rm0.clone(), rv0.clone()
    want = F.batch_norm(xs.double(), rm.double(), rv.double(), gs.double(), bs.double(), True, 0.1, 1e-5)
    if res:
        want = want + rs.double()
    if relu:
        want = torch.relu(want)
    want.backward(dy.double())

    rm1, rv1 = rm0.clone(), rv0.clone()
    y, mean, invstd = hip.bn_train_forward(x, r, gamma, beta, 1e-5, 0.1, rm1, rv1, relu)
    assert y.is_contiguous(memory_format=torch.channels_last)
    assert _rel(y.double(), want.detach()) < 2e-6
    n = P * H * H
    ref_mean = x.double().mean((0, 2, 3))
    ref_var = x.double().var((0, 2, 3), unbiased=False)
    assert _rel(mean.double(), ref_mean) < 1e-6
    assert _rel(invstd.double(), 1 / torch.sqrt(ref_var + 1e-5)) < 1e-6
    assert _rel(rm1.double(), 0.9 * rm0.double() + 0.1 * ref_mean) < 1e-6
    unb = ref_var * n / max(n - 1, 1)
    assert _rel(rv1.double(), 0.9 * rv0.double() + 0.1 * unb) < 1e-6

    dx, dres, dgamma, dbeta = hip.bn_train_backward(dy, y if relu else None, x, gamma, mean, invstd, relu, res)
    tol = 2e-5
    assert _rel(dx.double(), xs.grad.double()) < tol
    assert _rel(dgamma.double(), gs.grad.double()) < tol
    assert _rel(dbeta.double(), bs.grad.double()) < tol
    if res:
        assert _rel(dres.double(), rs.grad.double()) < tol
    # deterministic: a second run gives the same bits
    y2, mean2, _ = hip.bn_train_forward(x, r, gamma, beta, 1e-5, 0.1, rm0.clone(), rv0.clone(), relu)
    assert torch.equal(y, y2) and torch.equal(mean, mean2)


@pytest.mark.parametrize("P,ci,co,k,s,H", [(16, 64, 64, 3, 1, 8), (1023, 64, 64, 3, 1, 8), (5, 64, 128, 3, 2, 8), (34, 64, 128, 1, 2, 8),
                                             (7, 128, 128, 3, 1, 4), (1024, 128, 128, 3, 1, 4), (1, 64, 64, 3, 1, 8)])
def test_conv_epilogue_statistics_feed_the_batch_norm(P, ci, co, k, s, H):
    """``conv2d_nhwc(..., stats_shift=)``: the LDS-resident convolutions take the BatchNorm's batch statistics off their
    accumulators - per slab of 4 patches, around a shift near the mean - and ``bn_train_forward_partials`` combines them:
    the same convolution output bit for bit, sums that match float64 sums of it, and a BatchNorm (+ residual + ReLU) that
    agrees with the one that makes a reduction pass of its own (ragged patch counts: the last slab is partly empty)."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(P + ci + 7 * k)
    x = (torch.randn((P, ci, H, H), generator=g) + 0.5).to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((co, ci, k, k), generator=g) * (2.0 / (ci * k * k)) ** 0.5).to(dev)
    pad = k // 2
    shift = (torch.randn(co, generator=g) * 0.3).to(dev)
    plain = hip.conv2d_nhwc(x, w, s, pad)
    y, partial, slabs = hip.conv2d_nhwc(x, w, s, pad, stats_shift=shift)
    assert torch.equal(y, plain) and slabs == (P + 3) // 4 and partial.shape == (slabs, 2, co)
    d = y.double() - shift.double().view(1, -1, 1, 1)
    assert _rel(partial[:, 0].double().sum(0), d.sum((0, 2, 3))) < 1e-5
    assert _rel(partial[:, 1].double().sum(0), (d * d).sum((0, 2, 3))) < 1e-6
    gamma, beta = (torch.rand(co, generator=g) + 0.5).to(dev), torch.randn(co, generator=g).to(dev)
    res = torch.randn(y.shape, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    rm0, rv0 = shift.clone(), (torch.rand(co, generator=g) + 0.5).to(dev)
    rm1, rv1, rm2, rv2 = rm0.clone(), rv0.clone(), rm0.clone(), rv0.clone()
    want, mean_w, inv_w = hip.bn_train_forward(y, res, gamma, beta, 1e-5, 0.1, rm1, rv1, True)
    # (the shift IS the running mean buffer, as the training path passes it: read before it is updated)
    y2, partial2, _ = hip.conv2d_nhwc(x, w, s, pad, stats_shift=rm2)
    got, mean_g, inv_g = hip.bn_train_forward_partials(y2, res, gamma, beta, 1e-5, 0.1, rm2, rv2, True, partial2, slabs, rm2)
    assert _rel(mean_g, mean_w) < 1e-6 and _rel(inv_g, inv_w) < 2e-6
    assert _rel(got, want) < 5e-6
    assert _rel(rm2, rm1) < 1e-6 and _rel(rv2, rv1) < 2e-6
    ref_mean = y.double().mean((0, 2, 3))
    ref_var = y.double().var((0, 2, 3), unbiased=False)
    assert _rel(mean_g.double(), ref_mean) < 1e-6 and _rel(inv_g.double(), 1 / torch.sqrt(ref_var + 1e-5)) < 2e-6
    # deterministic
    y3, partial3, _ = hip.conv2d_nhwc(x, w, s, pad, stats_shift=shift)
    assert torch.equal(partial3, partial)


def test_bn_train_rejects_unsupported_channel_counts():
    assert not hip.bn_train_supported(16, 48)
    assert not hip.bn_train_supported(16, 6)
    assert hip.bn_train_supported(16, 64)
    x = torch.zeros((2, 48, 2, 2), device="cuda:0").contiguous(memory_format=torch.channels_last)
    w = torch.ones(48, device="cuda:0")
    with pytest.raises(RuntimeError, match="power of two"):
        hip.bn_train_forward(x, None, w, w, 1e-5, 0.1, None, None, True)


def _stock_taps(encoder, x):
    """encoder(x).flatten(1) on stock ops, also returning every post-ReLU activation (BasicBlock.forward spelled out)."""
    mods = list(encoder.children())
    h = x
    for m in mods[:4]:
        h = m(h)
    taps = [h]
    for stage in mods[4:-1]:
        for blk in stage:
            o = torch.relu(blk.bn1(blk.conv1(h)))
            taps.append(o)
            o = blk.bn2(blk.conv2(o))
            idt = h if blk.downsample is None else blk.downsample(h)
            h = torch.relu(o + idt)
            taps.append(h)
    return mods[-1](h).flatten(1), taps


@pytest.mark.parametrize("conf_fn,patch", [(synth.mnist_conf, 32), (synth.traffic_conf, 64)])
def test_fused_encoder_matches_float64_autograd(conf_fn, patch):
    """Same modules, same input: embeddings, loss, every parameter gradient and the BatchNorm running statistics of the
    fused path against the SAME modules evaluated by stock autograd in float64 (round 5; rounds 2-4 compared with the
    stock float32 path, whose convolution algorithm MIOpen picks per process - the set of usable seeds moved from run to
    run and the bound had to be loosened until a real bug could have passed).

    One caveat is inherent to ReLU networks, not to this path: an activation that is zero to rounding (|y| ~ 1e-7) can
    come out as +1 ulp in float32 and as -1e-9 in float64, which switches that element's gradient on or off.  The ReLU
    masks of both evaluations are compared first; a seed with such a flip (rare: ~0.4 M activations in the 2-stage
    trunk, 3 M in the 4-stage trunk on 64-px patches - where nine seeds of ten have one: BatchNorm centres every map on
    zero) is held to a loose bound only, and at least ONE seed of at most ten must be free of flips and pass the tight
    bound on every parameter gradient (float64 and these kernels are deterministic: the same seeds every run)."""
    dev = torch.device("cuda:0")
    conf = conf_fn(N=64, M=8, I=8, patch=patch)
    clean = 0
    for seed in range(10):
        if seed >= 3 and clean >= 1:
            break
        net_a = synth.fill_weights(IPSNet(dev, conf), 5 + seed).to(dev).train()
        enc_b = copy.deepcopy(net_a.encoder).double()
        assert fused_encoder.supported(net_a.encoder)
        g = torch.Generator(device="cpu").manual_seed(seed)
        P = 24
        x = torch.rand((P, conf.n_chan_in, patch, patch), generator=g).to(dev)
        t = torch.randn((P, conf.D), generator=g).to(dev)

        taps_a = []
        emb_a = fused_encoder.encode(net_a.encoder, x, taps_a)
        loss_a = ((emb_a - t) ** 2).mean()
        loss_a.backward()
        emb_b, taps_b = _stock_taps(enc_b, x.double())
        loss_b = ((emb_b - t.double()) ** 2).mean()
        loss_b.backward()

        assert _rel(emb_a.detach().double(), emb_b.detach()) < 1e-5
        assert abs(float(loss_a.detach()) - float(loss_b.detach())) <= 1e-5 * abs(float(loss_b.detach()))
        flips = sum(int(((ha > 0) != (hb > 0)).sum()) for ha, hb in zip(taps_a, taps_b))
        assert flips <= 8
        clean += flips == 0
        tol = 5e-5 if flips == 0 else 0.2
        for (na, pa), (nb, pb) in zip(net_a.encoder.named_parameters(), enc_b.named_parameters()):
            assert na == nb and pa.grad is not None
            assert _rel(pa.grad.double(), pb.grad) < tol, (seed, na, flips)
        for (na, ba), (nb, bb) in zip(net_a.encoder.named_buffers(), enc_b.named_buffers()):
            if na.endswith("num_batches_tracked"):
                assert int(ba) == int(bb) == 1
            else:
                assert _rel(ba.double(), bb) < 1e-5, na
    assert clean >= 1


def test_conv_bn_node_equals_the_two_nodes(monkeypatch):
    """``fused_encoder.encode`` with the statistics off the convolutions' epilogues (one conv+bn node per BasicBlock half)
    against the same encoder with ``IPSX_TRAIN_CONV_STATS=0`` (convolution and BatchNorm as two nodes, the BatchNorm making
    its own reduction pass): embeddings, every gradient and the running statistics agree to fp32 rounding."""
    dev = torch.device("cuda:0")
    conf = synth.mnist_conf(N=64, M=8, I=8)
    base = synth.fill_weights(IPSNet(dev, conf), 3).to(dev).train()
    x = synth.make_patches(conf, 2, seed=5).to(dev)[:, :40].reshape(80, 1, 32, 32)
    outs = []
    for stats in ("1", "0"):
        monkeypatch.setenv("IPSX_TRAIN_CONV_STATS", stats)
        enc = copy.deepcopy(base.encoder)
        y = fused_encoder.encode(enc, x)
        (y * torch.linspace(-1, 1, y.shape[1], device=dev)).sum().backward()
        outs.append((y.detach(), [p.grad.clone() for p in enc.parameters()],
                     [b.clone() for n, b in enc.named_buffers() if "running" in n]))
    (y1, g1, b1), (y0, g0, b0) = outs
    assert _rel(y1, y0) < 2e-5
    for a, b in zip(g1, g0):
        assert _rel(a, b) < 2e-4
    for a, b in zip(b1, b0):
        assert _rel(a, b) < 1e-5


def test_conv_bn_statistics_do_not_depend_on_where_the_running_mean_is():
    """The conv+bn node's batch statistics are sums around a SHIFT taken in the convolution's epilogue; the shift is data
    derived (the first patch's channel means, then the previous step's batch mean) - a ``running_mean`` 50 standard
    deviations away from the batch mean (foreign statistics loaded into the module) costs no precision (advisor, round 5:
    the shift used to be ``running_mean`` itself)."""
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(11)
    P, ci, co, H = 96, 64, 64, 8
    conv = torch.nn.Conv2d(ci, co, 3, 1, 1, bias=False).to(dev)
    bn = torch.nn.BatchNorm2d(co).to(dev).train()
    with torch.no_grad():
        conv.weight.copy_((torch.randn((co, ci, 3, 3), generator=g) * (2.0 / (ci * 9)) ** 0.5).to(dev))
    x = (torch.randn((P, ci, H, H), generator=g) + 0.5).to(dev).contiguous(memory_format=torch.channels_last)
    c = torch.nn.functional.conv2d(x.double(), conv.weight.double(), None, 1, 1)
    ref_mean, ref_var = c.mean((0, 2, 3)), c.var((0, 2, 3), unbiased=False)
    ref = (c - ref_mean.view(1, -1, 1, 1)) / torch.sqrt(ref_var.view(1, -1, 1, 1) + bn.eps)
    with torch.no_grad():
        bn.running_mean.copy_((ref_mean + 50.0 * torch.sqrt(ref_var)).float())
    for step in range(2):                       # the first step's shift: the first patch; the second's: the first step's mean
        y = fused_encoder.conv_bn_act(conv, bn, x, relu=False)
        assert hasattr(bn, "_ipsx_batch_mean")
        assert _rel(bn._ipsx_batch_mean.double(), ref_mean) < 1e-6
        assert _rel(y.double(), ref) < 2e-5, step
    # an empty batch: an empty output, the statistics untouched (it used to raise out of the convolution's launch)
    rm = bn.running_mean.clone()
    empty = fused_encoder.conv_bn_act(conv, bn, x[:0], relu=False)
    assert empty.shape == (0, co, H, H) and torch.equal(bn.running_mean, rm)


def test_training_forward_uses_the_fused_path_and_env_switches_it_off(monkeypatch):
    dev = torch.device("cuda:0")
    conf = synth.mnist_conf(N=64, M=8, I=8)
    net = synth.fill_weights(IPSNet(dev, conf), 5).to(dev).train()
    x = synth.make_patches(conf, 2, seed=0).to(dev)
    calls = []
    real = fused_encoder.encode
    monkeypatch.setattr(fused_encoder, "encode", lambda enc, t: (calls.append(1), real(enc, t))[1])
    mp, pos = net.ips(x)
    assert not calls                       # ips() is no-grad / eval: the fused TRUNK kernels, not this path
    net(mp, pos)
    assert calls == [1]
    monkeypatch.setenv("IPSX_TRAIN_FUSED", "0")
    net(mp, pos)
    assert calls == [1]
    monkeypatch.delenv("IPSX_TRAIN_FUSED")
    with torch.no_grad():                  # train mode without autograd: stock ops (nothing to fuse a backward for)
        net(mp, pos)
    assert calls == [1]
