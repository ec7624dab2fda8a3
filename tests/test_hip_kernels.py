"""Kernel-level parity: every entry point of libipsx.so, called through the C ABI
(ips_amd.hip -> ctypes), against the CPU oracle on the same seeded inputs.

The kernels restate the oracle's arithmetic order exactly (fp32 MFMA = ordered fma
chain; own exp; wave-order sums), so floating-point results are compared BIT FOR BIT
(ulp distance 0); indices are compared exactly.
"""

import ctypes as C

import numpy as np
import pytest
import torch

from ips_amd import hip, synth
from oracle import oracle as orc
from tests.util import Golden, ulp_diff

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def rnd(shape, seed, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def test_device_is_gfx950_and_library_loaded():
    L = hip.lib()
    assert L.ipsx_device_count() >= 1
    assert L.ipsx_device_is_gfx950(0) == 1


def test_det_expf_and_wave_sum_bits_via_head_kernel():
    """softmax head = det_expf + wave butterfly on device; must equal the oracle bit for bit."""
    B, T, D, ncls = 5, 4, 128, 10
    emb, w, b = rnd((B, T, D), 1), rnd((ncls, D), 2, 0.3), rnd((ncls,), 3)
    lin = torch.nn.Linear(D, ncls)
    lin.weight.data, lin.bias.data = torch.from_numpy(w), torch.from_numpy(b)
    lin = lin.to(DEV)
    for act, code in (("softmax", 0), ("sigmoid", 1)):
        got = hip.head(dev(emb), 2, lin, act).cpu().numpy()
        want = np.empty_like(got)
        for i in range(B):
            orc.lib().orc_head(orc._f(emb[i, 2])[1], D, orc._f(w)[1], orc._f(b)[1], ncls, code,
                               want[i].ctypes.data_as(orc.f32p))
        assert ulp_diff(got, want) == 0, act


@pytest.mark.parametrize("c_in,c_out,k,stride,pad,h,w,n,res,relu", [
    (64, 64, 3, 1, 1, 8, 8, 5, True, True),        # layer1 conv2 shape (MNIST-32)
    (64, 128, 3, 2, 1, 8, 8, 7, False, True),      # layer2.0.conv1
    (64, 128, 1, 2, 0, 8, 8, 3, False, False),     # layer2.0.downsample
    (128, 128, 3, 1, 1, 4, 4, 9, True, True),      # layer2 convs, 4x4 maps
    (1, 64, 7, 2, 3, 32, 32, 3, False, True),      # 1-channel stem
    (3, 64, 7, 2, 3, 37, 29, 2, False, True),      # 3-channel stem, odd sizes
    (64, 64, 3, 1, 1, 13, 13, 3, True, True),      # 50-px patches: 13x13 maps (no vec4 stores)
    (16, 32, 3, 1, 1, 5, 7, 4, False, False),      # C_out = 32: one n-tile only
])
def test_conv2d_affine_bit_exact(c_in, c_out, k, stride, pad, h, w, n, res, relu):
    x = rnd((n, c_in, h, w), 10)
    wt = rnd((c_out, c_in, k, k), 11, (2.0 / (c_in * k * k)) ** 0.5)
    alpha, shift = (1 + 0.2 * rnd((c_out,), 12)), rnd((c_out,), 13, 0.1)
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    r = rnd((n, c_out, ho, wo), 14) if res else None
    # oracle
    cv = orc._Conv(c_in, c_out, k, k, stride, pad, orc._f(wt)[1], orc._f(alpha)[1], orc._f(shift)[1])
    want = np.empty((n, c_out, ho, wo), dtype=np.float32)
    orc.lib().orc_conv2d_affine(C.byref(cv), orc._f(x)[1], orc._f(r)[1] if res else None,
                                want.ctypes.data_as(orc.f32p), C.c_int64(n), h, w, int(relu))
    # device
    packed = hip._pack_conv(dev(wt))
    a, s = dev(alpha), dev(shift)
    hcv = hip.Conv(c_in, c_out, k, k, stride, pad, packed.data_ptr(), a.data_ptr(), s.data_ptr())
    y = torch.full((n, c_out, ho, wo), float("nan"), device=DEV)
    xd, rd = dev(x), (dev(r) if res else None)
    hip._ck(hip.lib().ipsx_conv2d_affine(C.byref(hcv), hip._p(xd), hip._p(rd), hip._p(y), n, h, w, int(relu),
                                         hip._stream()), "conv")
    got = y.cpu().numpy()
    assert not np.isnan(got).any(), "kernel left outputs unwritten"
    assert ulp_diff(got, want) == 0, "max abs diff %g" % np.abs(got - want).max()


@pytest.mark.parametrize("c_in,c_out,k,stride,pad,h,w,n,res,relu", [
    (64, 64, 3, 1, 1, 8, 8, 5, True, True),        # layer1 shape
    (64, 128, 3, 2, 1, 13, 13, 3, False, True),    # strided, odd map (50-px patches)
    (64, 128, 1, 2, 0, 25, 25, 2, False, False),   # 1x1 strided projection (traffic layer2 shortcut)
    (128, 256, 3, 2, 1, 13, 13, 3, False, True),   # wide layer: waves along N
    (256, 512, 3, 1, 1, 4, 4, 5, True, True),
    (2048, 512, 1, 1, 0, 1, 1, 300, False, True),  # the projector Linear: rows x 2048 -> 512
    (32, 32, 3, 1, 1, 5, 7, 4, False, False),      # smallest supported: one n-tile, 4 stages per tap
    # large maps
    (64, 64, 3, 1, 1, 13, 13, 5, True, True),      # 50-px patches, layer1: whole images per workgroup, odd image count
    (64, 64, 3, 1, 1, 25, 25, 3, True, True),      # 100-px patches, layer1: strips of rows
    (64, 128, 3, 2, 1, 25, 25, 3, False, True),    # strided, four n-tiles
    (128, 128, 3, 1, 1, 13, 13, 4, True, True),    # 128 input channels
    (64, 64, 3, 1, 1, 10, 12, 3, False, False),    # not square, no BatchNorm tail
    (64, 64, 3, 2, 1, 26, 22, 2, True, False),     # strided, even map, residual without ReLU
])
def test_conv2d_affine_nhwc_bit_exact(c_in, c_out, k, stride, pad, h, w, n, res, relu):
    """Channels-last conv (16-byte buffer loads, hardware-zeroed halo) vs the oracle (NCHW)."""
    x = rnd((n, c_in, h, w), 50)
    wt = rnd((c_out, c_in, k, k), 51, (2.0 / (c_in * k * k)) ** 0.5)
    alpha, shift = (1 + 0.2 * rnd((c_out,), 52)), rnd((c_out,), 53, 0.1)
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    r = rnd((n, c_out, ho, wo), 54) if res else None
    cv = orc._Conv(c_in, c_out, k, k, stride, pad, orc._f(wt)[1], orc._f(alpha)[1], orc._f(shift)[1])
    want = np.empty((n, c_out, ho, wo), dtype=np.float32)
    orc.lib().orc_conv2d_affine(C.byref(cv), orc._f(x)[1], orc._f(r)[1] if res else None,
                                want.ctypes.data_as(orc.f32p), C.c_int64(n), h, w, int(relu))
    packed = hip._pack_conv(dev(wt))
    a, s = dev(alpha), dev(shift)
    hcv = hip.Conv(c_in, c_out, k, k, stride, pad, packed.data_ptr(), a.data_ptr(), s.data_ptr())
    xd = dev(np.ascontiguousarray(x.transpose(0, 2, 3, 1)))
    rd = dev(np.ascontiguousarray(r.transpose(0, 2, 3, 1))) if res else None
    y = torch.full((n, ho, wo, c_out), float("nan"), device=DEV)
    hip._ck(hip.lib().ipsx_conv2d_affine_nhwc(C.byref(hcv), hip._p(xd), hip._p(rd), hip._p(y), n, h, w, int(relu),
                                              hip._stream()), "conv nhwc")
    got = y.cpu().numpy().transpose(0, 3, 1, 2)
    assert not np.isnan(got).any(), "kernel left outputs unwritten"
    assert ulp_diff(got, want) == 0, "max abs diff %g" % np.abs(got - want).max()


def test_nhwc_pools_bit_exact():
    c = 64
    x = rnd((3, c, 13, 13), 60)
    xd = dev(np.ascontiguousarray(x.transpose(0, 2, 3, 1)))
    y = torch.empty((3, 7, 7, c), device=DEV)
    hip._ck(hip.lib().ipsx_maxpool_3x3s2_nhwc(hip._p(xd), hip._p(y), 3, c, 13, 13, hip._stream()), "maxpool nhwc")
    ref = torch.nn.functional.max_pool2d(torch.from_numpy(x), 3, 2, 1).numpy()
    assert np.array_equal(y.cpu().numpy().transpose(0, 3, 1, 2), ref)
    z = torch.empty((3, c), device=DEV)
    hip._ck(hip.lib().ipsx_avgpool_nhwc(hip._p(xd), hip._p(z), 3, c, 169, hip._stream()), "avgpool nhwc")
    want = np.empty((3, c), dtype=np.float32)
    orc.lib().orc_avgpool(orc._f(x)[1], want.ctypes.data_as(orc.f32p), C.c_int64(3), c, 169)
    assert ulp_diff(z.cpu().numpy(), want) == 0


def test_bn_affine_and_pools_bit_exact():
    c = 64
    bn = torch.nn.BatchNorm2d(c)
    g = np.random.default_rng(5)
    bn.weight.data = torch.from_numpy(g.uniform(0.5, 1.5, c).astype(np.float32))
    bn.bias.data = torch.from_numpy(rnd((c,), 6, 0.1))
    bn.running_mean = torch.from_numpy(rnd((c,), 7, 0.1))
    bn.running_var = torch.from_numpy(g.uniform(0.5, 1.5, c).astype(np.float32))
    want = orc.bn_affine(bn)
    got = hip._bn_affine(bn.to(DEV)).cpu().numpy()
    assert ulp_diff(got[0], want[0]) == 0 and ulp_diff(got[1], want[1]) == 0

    x = rnd((3, c, 16, 16), 8)
    y = torch.empty((3, c, 8, 8), device=DEV)
    xd = dev(x)
    hip._ck(hip.lib().ipsx_maxpool_3x3s2(hip._p(xd), hip._p(y), 3, c, 16, 16, hip._stream()), "maxpool")
    ref = torch.nn.functional.max_pool2d(torch.from_numpy(x), 3, 2, 1).numpy()
    assert np.array_equal(y.cpu().numpy(), ref)
    x2 = rnd((3, c, 13, 13), 9)                      # odd size -> ho = 7
    y2 = torch.empty((3, c, 7, 7), device=DEV)
    x2d = dev(x2)
    hip._ck(hip.lib().ipsx_maxpool_3x3s2(hip._p(x2d), hip._p(y2), 3, c, 13, 13, hip._stream()), "maxpool")
    assert np.array_equal(y2.cpu().numpy(), torch.nn.functional.max_pool2d(torch.from_numpy(x2), 3, 2, 1).numpy())

    z = torch.empty((3, c), device=DEV)
    hip._ck(hip.lib().ipsx_avgpool(hip._p(xd), hip._p(z), 3, c, 256, hip._stream()), "avgpool")
    want = np.empty((3, c), dtype=np.float32)
    orc.lib().orc_avgpool(orc._f(x)[1], want.ctypes.data_as(orc.f32p), C.c_int64(3), c, 256)
    assert ulp_diff(z.cpu().numpy(), want) == 0


@pytest.mark.parametrize("case,n", [("mnist_mini", 70), ("traffic_tiny", 5), ("mnist_native50", 6),
                                    ("cam_b2", 300)])
def test_encoder_bit_exact(case, n):
    g = Golden(case)
    net = g.net(DEV)
    x = g.patches()[0, :n]
    plan = hip.EncoderPlan(net.encoder, g.conf.is_image)
    got = plan.encode(x.to(DEV)).cpu().numpy()
    want = orc.Oracle(g.net("cpu")).encode(x.numpy())
    assert got.shape == want.shape
    assert ulp_diff(got, want) == 0, "max abs diff %g" % np.abs(got - want).max()


@pytest.mark.parametrize("n", [1, 2, 3, 7, 100])
def test_fused_64_channel_stage_equals_layered_kernels(n, monkeypatch):
    """50-px patches (the reference's shipped Megapixel-MNIST patch size): layer1 - four 64 -> 64 convolutions on 13x13
    maps - runs as ONE LDS-resident kernel (fused_stage.hip, 3 patches per workgroup: 1, 2 patches and a multiple of 3
    exercise the short last workgroup).  Same bits as the layer-by-layer kernels and as the oracle."""
    g = Golden("mnist_native50")
    net = g.net(DEV)
    x = g.patches()[0, :n].to(DEV)
    plan = hip.EncoderPlan(net.encoder, True)
    fused = plan.encode(x)
    monkeypatch.setenv("IPSX_NO_FUSED", "1")
    layered = plan.encode(x)
    monkeypatch.delenv("IPSX_NO_FUSED")
    assert torch.equal(fused, layered), "max abs diff %g" % float((fused - layered).abs().max())
    if n <= 7:
        want = orc.Oracle(g.net("cpu")).encode(x.cpu().numpy())
        assert ulp_diff(fused.cpu().numpy(), want) == 0


@pytest.mark.parametrize("n", [1, 5])
def test_fused_traffic_stem_equals_layered_kernels(n, monkeypatch):
    """3x100x100 patches (config/traffic_config.yml): conv7x7/2 + BN + ReLU + max-pool as one kernel
    (stem_pool100x3_kernel, one patch per workgroup, pool on the accumulators) - same bits as conv_any_kernel +
    maxpool_3x3s2_nhwc_kernel through the whole ResNet-18 x 4 trunk, and as the oracle."""
    g = Golden("traffic_full")
    net = g.net(DEV)
    x = g.patches()[0, :n].to(DEV)
    plan = hip.EncoderPlan(net.encoder, True)
    fused = plan.encode(x)
    assert hip.encoder_kernel_name(plan).startswith("stem_pool100x3_kernel")
    monkeypatch.setenv("IPSX_NO_FUSED", "1")
    layered = plan.encode(x)
    assert hip.encoder_kernel_name(plan).startswith("conv_nhwc_kernel")
    monkeypatch.delenv("IPSX_NO_FUSED")
    assert torch.equal(fused, layered), "max abs diff %g" % float((fused - layered).abs().max())
    if n == 1:
        want = orc.Oracle(g.net("cpu")).encode(x.cpu().numpy())
        assert ulp_diff(fused.cpu().numpy(), want) == 0


@pytest.fixture
def trunk_kernel_choice():
    """ipsx_dbg_fused_trunk_pair: 0 the product's rule, 1 fused_trunk_kernel only, 2 fused_trunk_pair_kernel only."""
    fn = hip.lib().ipsx_dbg_fused_trunk_pair
    fn.restype, fn.argtypes = None, [C.c_int]
    yield fn
    fn(0)


@pytest.mark.parametrize("mode", [1, 2])
def test_fused_trunk_equals_layered_kernels(mode, monkeypatch, trunk_kernel_choice):
    """The LDS-resident fused trunk (1x32x32 patches) vs the layer-by-layer kernels vs the oracle: the kernel with one
    wavefront per patch (mode 1) and the one with two (mode 2, fused_trunk_pair.h; by the product's rule only what is
    left over after the whole rounds of the first)."""
    g = Golden("mnist_full")
    net = g.net(DEV)
    x = g.patches()[0, :203].to(DEV)                 # 203 = 50*4 + 3 = 101*2 + 1: exercises the tail workgroup of either
    plan = hip.EncoderPlan(net.encoder, True)
    trunk_kernel_choice(mode)
    fused = plan.encode(x)
    assert hip.encoder_kernel_name(plan) == "fused_trunk_kernel"
    monkeypatch.setenv("IPSX_NO_FUSED", "1")
    layered = plan.encode(x)
    assert hip.encoder_kernel_name(plan).startswith("conv_nhwc_kernel")
    monkeypatch.delenv("IPSX_NO_FUSED")
    assert torch.equal(fused, layered)
    want = orc.Oracle(g.net("cpu")).encode(x.cpu().numpy())
    assert ulp_diff(fused.cpu().numpy(), want) == 0


@pytest.mark.parametrize("n", [1, 2, 453, 2048 + 452, 2048 + 513, 4096 + 37])
def test_fused_trunk_remainder_rule_is_bit_neutral(n, trunk_kernel_choice):
    """What does not fill a round of fused_trunk_kernel (8 patches per CU) is encoded by fused_trunk_pair_kernel when it is
    at most a quarter round: one image of the headline workload is 2,048 + 452 patches.  Whatever the rule picks, the
    embeddings are those of the one-wavefront-per-patch kernel bit for bit - plain and through an index list."""
    g = Golden("mnist_full")
    plan = hip.EncoderPlan(g.net(DEV).encoder, True)
    gen = torch.Generator().manual_seed(n)
    x = torch.randn((n, 1, 32, 32), generator=gen)
    x[torch.rand(n, generator=gen) < 0.3] = 0.0                       # blank patches as on the canvases
    x = x.to(DEV)
    index = torch.randperm(n, generator=gen).to(torch.int32).to(DEV)
    trunk_kernel_choice(1)
    want, want_ix = plan.encode(x), plan.encode_indexed(x, index)
    assert torch.equal(want[index.long()], want_ix)
    for mode in (0, 2):
        trunk_kernel_choice(mode)
        assert torch.equal(plan.encode(x), want), mode
        assert torch.equal(plan.encode_indexed(x, index), want_ix), mode


@pytest.mark.parametrize("n,wgs,use_pos,quads", [(1, 0, True, -1), (2, 3, False, 0), (77, 5, True, 2), (79, 5, True, 100),
                                                 (2500, 0, True, -1), (2500, 0, False, 0), (2501, 0, True, 1)])
def test_trunk_stream_equals_encode_plus_logits(n, wgs, use_pos, quads):
    """ipsx_trunk_stream - one image's patches through the fused trunk four or two at a time by resident workgroups, their logits
    from emb + pos in the same workgroup, patches published in order as they complete - leaves the bits of
    ipsx_trunk_encode + ipsx_logits and a progress word that ends on (all but the last finishers' share of) the patch count."""
    g = Golden("mnist_full")
    net = g.net(DEV)
    plan = hip.EncoderPlan(net.encoder, True)
    ca = net.transf.crs_attn
    vq, R = ca.folded_query(), ca.H * ca.n_token
    gen = torch.Generator().manual_seed(n)
    x = torch.randn((n, 1, 32, 32), generator=gen)
    x[torch.rand(n, generator=gen) < 0.5] = 0.0
    x = x.to(DEV)
    pos = torch.randn((n, 128), generator=gen).to(DEV) if use_pos else None
    assert plan.image_stream_supported(x.shape, 128, R)
    want_emb = plan.encode(x)
    want_lg = hip.logits(want_emb.view(1, n, -1), pos.view(1, n, -1) if use_pos else None, vq, R)[0]
    emb = torch.full_like(want_emb, float("nan"))
    lg = torch.full_like(want_lg, float("nan"))
    for rep in range(2):
        ctl = torch.zeros((plan.image_stream_ctl_words(n),), dtype=torch.int32, device=DEV)
        ready = torch.zeros((1,), dtype=torch.int32, device=DEV)
        plan.image_stream(x, pos, vq, R, emb, lg, ctl, ready, workgroups=wgs, quad_pulls=quads)
        torch.cuda.synchronize()
        pairs = -(-n // 2)
        assert bool((ctl[2:2 + pairs] == 1).all())
        cursor = int(ctl[1].item())
        assert cursor <= pairs and int(ready.item()) == min(n, 2 * cursor) and 2 * cursor >= n - 4 * max(1, wgs or 264)
        assert torch.equal(emb, want_emb), "max abs diff %g" % float((emb - want_emb).abs().max())
        assert torch.equal(lg, want_lg), "max abs diff %g" % float((lg - want_lg).abs().max())


def test_ips_finish_is_the_two_gathers_the_index_copy_and_the_status_mirror():
    """ipsx_ips_finish - the end of an ips() call with a resident loop as ONE launch - against ipsx_gather_rows on the same
    indices: image patches with a per-image positional table, feature rows with ONE table expanded over the batch, no
    table at all; the index buffer comes back as a fresh tensor and the status word lands in the pinned host mirror."""
    g = torch.Generator(device="cpu").manual_seed(3)
    for shape, pos_mode in (((3, 50, 1, 32, 32), "per_image"), ((2, 300, 2048), "expanded"), ((1, 70, 3, 8, 8), None)):
        src = torch.randn(shape, generator=g).to(DEV)
        B, N = shape[:2]
        M = 17
        idx = torch.randint(0, N, (B, M), generator=g).to(DEV)
        pos = None
        if pos_mode == "per_image":
            pos = torch.randn((B, N, 128), generator=g).to(DEV)
        elif pos_mode == "expanded":
            pos = torch.randn((1, N, 512), generator=g).to(DEV).expand(B, -1, -1)
        status = torch.tensor([2], dtype=torch.int32, device=DEV)
        mirror = torch.zeros((1,), dtype=torch.int32).pin_memory()
        assert hip.ips_finish_supported(src, pos)
        got_idx, got_patch, got_pos = hip.ips_finish(src, pos, idx, status, mirror)
        torch.cuda.synchronize()
        assert got_idx.data_ptr() != idx.data_ptr() and torch.equal(got_idx, idx)
        assert torch.equal(got_patch, hip.gather_rows(src, idx))
        assert (got_pos is None) == (pos is None) and (pos is None or torch.equal(got_pos, hip.gather_rows(pos, idx)))
        assert int(mirror.item()) == 2
    assert not hip.ips_finish_supported(torch.zeros((1, 4, 3), device=DEV), None)          # rows of 12 bytes: the gathers take over


@pytest.mark.parametrize("head", [16, 48])
def test_projector_stream_with_the_first_units_in_column_quarters(head):
    """The optional head of a lone slide's stream (IPSX_CAM_HEAD / ipsx_dbg_stream_head): the FIRST units go out as column
    quarters to 3/8 of the workgroups, beside the others' whole tiles - embeddings and logits stay those of the launch by
    launch projector, every unit is published, the progress word ends on the row count."""
    conf, _ = synth.bench_workload("cam")
    from ips_amd.architecture.ips_net import IPSNet
    net = synth.fill_weights(IPSNet(torch.device(DEV), conf), 7).to(DEV).eval()
    plan = hip.EncoderPlan(net.encoder, False)
    ca = net.transf.crs_attn
    vq, R = ca.folded_query(), ca.H * ca.n_token
    n = 65536 - 19
    x = torch.from_numpy(rnd((n, conf.n_chan_in), 77, 2.0) + 0.25).to(DEV)
    want_emb = plan.encode(x)
    want_lg = hip.logits(want_emb.view(1, n, -1), None, vq, R)[0]
    fn = hip.lib().ipsx_dbg_stream_head
    fn.restype, fn.argtypes = None, [C.c_int]
    fn(head)
    try:
        for rep in range(2):
            emb = torch.full_like(want_emb, float("nan"))
            lg = torch.full_like(want_lg, float("nan"))
            ctl = torch.zeros((plan.stream_ctl_words(n),), dtype=torch.int32, device=DEV)
            ready = torch.zeros((1,), dtype=torch.int32, device=DEV)
            plan.stream(x, vq, R, emb, lg, ctl, ready, workgroups=255, short_first=-11)
            torch.cuda.synchronize()
            units = -(-n // 32)
            assert bool((ctl[2:2 + units] == 1).all()) and int(ready.item()) == n and int(ctl[1].item()) == units
            assert torch.equal(emb, want_emb) and torch.equal(lg, want_lg)
    finally:
        fn(0)


@pytest.mark.parametrize("n,f", [(1, 2048), (31, 2048), (33, 64), (130, 1024), (4097, 2048), (257, 8), (64, 2056)])
def test_projector_moments_and_column_sums_are_the_oracles(n, f):
    """ipsx_projector_stats (row_moments_kernel: the moments in the order the GEMM's operand stream gives them, round 5)
    and ipsx_weight_colsum against the oracle's restatement, bit for bit: ragged row counts, rows of 8 ... 2,056 floats,
    post-ReLU-like rows with a large mean (where E[x^2] - mean^2 cancels most) and constant rows (variance clamped at 0)."""
    x = np.abs(rnd((n, f), 100 + n + f)) * 3.0 + 5.0
    x[0, :] = 2.5                                                 # a constant row: var = 0 exactly or clamped
    if n > 2:
        x[2, :] = 0.0
    want = np.empty((n, 2), dtype=np.float32)
    orc.lib().orc_projector_moments(orc._f(x)[1], C.c_int64(n), f, C.c_float(1e-5), want.ctypes.data_as(orc.f32p))
    got = torch.full((n, 2), float("nan"), device=DEV)
    hip._ck(hip.lib().ipsx_projector_stats(hip._p(dev(x)), n, f, C.c_float(1e-5), hip._p(got), None), "ipsx_projector_stats")
    assert ulp_diff(got.cpu().numpy(), want) == 0
    # float64 truth.  Rows whose mean is beyond 4 std come back CENTRED - (mean', -rstd), second moment taken around the
    # mean (round 6; round 5's single pass needed 5e-4 here) - these rows straddle that threshold
    mean64, var64 = x.astype(np.float64).mean(1), x.astype(np.float64).var(1)
    assert np.abs(want[:, 0] - mean64).max() <= 2e-6 * np.abs(mean64).max()
    rstd64 = 1.0 / np.sqrt(var64 + 1e-5)
    ok = var64 > 1e-3                                             # (a constant row's variance is rounding noise against eps)
    assert not ok.any() or (np.abs(np.abs(want[:, 1]) - rstd64) / rstd64)[ok].max() <= 2e-5
    assert want[0, 1] < 0 and want[0, 0] == 2.5                   # the constant row: centred, its mean exact
    assert f < 64 or n < 3 or ((want[3:, 1] < 0).any() and (want[3:, 1] > 0).any() and want[2, 1] > 0)
    w = rnd((96, f), 7 + f, 0.3)
    cs = np.empty((96,), dtype=np.float32)
    orc.lib().orc_weight_colsum(orc._f(w)[1], 96, f, cs.ctypes.data_as(orc.f32p))
    got_cs = torch.empty((96,), device=DEV)
    hip._ck(hip.lib().ipsx_weight_colsum(hip._p(dev(w)), 96, f, hip._p(got_cs), None), "ipsx_weight_colsum")
    assert ulp_diff(got_cs.cpu().numpy(), cs) == 0
    assert np.array_equal(cs, w.astype(np.float64).sum(1).astype(np.float32)) or np.abs(cs - w.astype(np.float64).sum(1)).max() < 1e-5


@pytest.mark.parametrize("n,wgs,short,slides", [(64, 0, -1, 1), (97, 3, 1, 1), (1000, 7, 3, 1), (4099, 0, -1, 1), (20000, 0, 0, 1),
                                                (3 * 1056, 5, 2, 3), (8 * 2048, 0, -1, 8), (5000, 9, -2, 1),
                                                # guided sizes; what is left over goes out in column quarters (8 / 3 / 16 units)
                                                (65536, 255, -20, 1), (203 * 32 - 7, 40, -12, 1), (208 * 32, 64, -20, 1)])
def test_projector_stream_equals_the_launch_by_launch_projector(n, wgs, short, slides):
    """ipsx_projector_stream - LayerNorm moments, Linear + BatchNorm + ReLU and the logits of every 64-row tile by resident
    workgroups, rows published in order as they complete - leaves the bits of ipsx_projector_stats + ipsx_projector_apply
    + ipsx_logits: embeddings, logits, and progress words that end on the row counts (ragged last unit, fewer
    workgroups than tiles, 32-row first tiles, several slides one after the other)."""
    conf, _ = synth.bench_workload("cam")
    from ips_amd.architecture.ips_net import IPSNet
    net = synth.fill_weights(IPSNet(torch.device(DEV), conf), 7).to(DEV).eval()
    plan = hip.EncoderPlan(net.encoder, False)
    ca = net.transf.crs_attn
    vq, R = ca.folded_query(), ca.H * ca.n_token
    x = torch.from_numpy(rnd((n, conf.n_chan_in), n, 3.0) + 0.5).to(DEV)
    assert plan.stream_supported(n, R)
    want_emb = plan.encode(x)
    want_lg = hip.logits(want_emb.view(1, n, -1), None, vq, R)[0]
    emb = torch.full_like(want_emb, float("nan"))
    lg = torch.full_like(want_lg, float("nan"))
    for rep in range(2):                                   # the control words are the caller's to zero, every call
        ctl = torch.zeros((plan.stream_ctl_words(n),), dtype=torch.int32, device=DEV)
        ready = torch.zeros((slides,), dtype=torch.int32, device=DEV)
        plan.stream(x, vq, R, emb, lg, ctl, ready, workgroups=wgs, short_first=short, slide_rows=n // slides)
        torch.cuda.synchronize()
        units = -(-n // 32)
        assert bool((ctl[2:2 + units] == 1).all())
        # (two workgroups that finish neighbouring units at the same moment may both leave the cursor to the next one
        #  that finishes; at the end of a slide that is the caller's ipsx_publish_rows - never more than the tail)
        cursor, per = int(ctl[1].item()), n // slides
        assert cursor <= units and cursor * 32 >= n - 64 * max(1, min(wgs or 224, units))
        want_ready = [max(0, min(per, min(n, 32 * cursor) - s_ * per)) for s_ in range(slides)]
        assert ready.tolist() == want_ready      # several slides (3 x 1,056 rows: tiles run across their ends): a word each
        assert torch.equal(emb, want_emb), "max abs diff %g" % float((emb - want_emb).abs().max())
        assert torch.equal(lg, want_lg), "max abs diff %g" % float((lg - want_lg).abs().max())


def test_projector_rows_whose_mean_dwarfs_their_spread():
    """Advisor r05 (medium): the folded LayerNorm - E[x^2] - mean^2 and acc - mean * colsum - cancels where a row's mean is
    large against its spread.  Such rows (mean^2 / var > 16) are centred like nn.LayerNorm does: moments around the mean,
    the Linear on x - mean (ipsx_rowstats.h).  Rows with mean / std from 0 to 1000, constant rows and zero rows, mixed
    inside the tiles: the launch-by-launch projector and the stream kernel (embeddings AND logits) leave the oracle's bits,
    and the values stay within 6e-5 (5e-4 at mean / std = 1000) of float64 LayerNorm -> Linear -> BatchNorm -> ReLU (the reference's own fp32 LayerNorm is
    4e-4 off at mean / std = 1000)."""
    conf, _ = synth.bench_workload("cam")
    from ips_amd.architecture.ips_net import IPSNet
    net = synth.fill_weights(IPSNet(torch.device(DEV), conf), 7).to(DEV).eval()
    cpu = synth.fill_weights(IPSNet(torch.device("cpu"), conf), 7).eval()
    n, f = 64 * 9 + 17, conf.n_chan_in
    x = rnd((n, f), 4242, 1.0)
    ratios = np.array([0.0, 3.5, 4.5, 30.0, 100.0, 1000.0, 1.0, 8.0])[np.arange(n) % 8]
    x = (x + ratios[:, None]).astype(np.float32)
    x[5] = 3.7; x[70] = -1e-3; x[71] = 0.0; x[200] = 65504.0
    want = orc.Oracle(cpu).encode(x)
    stats = np.empty((n, 2), dtype=np.float32)
    orc.lib().orc_projector_moments(orc._f(x)[1], C.c_int64(n), f, C.c_float(1e-5), stats.ctypes.data_as(orc.f32p))
    assert (stats[:, 1] < 0).sum() >= n // 2 and (stats[:, 1] > 0).sum() >= n // 4         # both kinds in every tile
    x64 = torch.from_numpy(x).double()
    enc = cpu.encoder.double()
    with torch.no_grad():
        truth = enc(x64).numpy()
    assert np.abs(want - truth).max() <= 5e-4 and np.abs(want - truth)[ratios <= 100].max() <= 6e-5
    assert np.array_equal(want[5], want[71]) and np.array_equal(want[5], want[200])         # constant rows: LayerNorm gives zeros
    plan = hip.EncoderPlan(net.encoder, False)
    xd = dev(x)
    got = plan.encode(xd)
    assert ulp_diff(got.cpu().numpy(), want) == 0
    ca = net.transf.crs_attn
    vq, R = ca.folded_query(), ca.H * ca.n_token
    want_lg = hip.logits(got.view(1, n, -1), None, vq, R)[0]
    for wgs, short in ((0, -1), (3, 1), (255, -20)):
        emb = torch.full_like(got, float("nan"))
        lg = torch.full_like(want_lg, float("nan"))
        ctl = torch.zeros((plan.stream_ctl_words(n),), dtype=torch.int32, device=DEV)
        ready = torch.zeros((1,), dtype=torch.int32, device=DEV)
        plan.stream(xd, vq, R, emb, lg, ctl, ready, workgroups=wgs, short_first=short, slide_rows=n)
        torch.cuda.synchronize()
        assert torch.equal(emb, got) and torch.equal(lg, want_lg)


@pytest.mark.parametrize("blank_frac", [0.93, 0.0, 1.0, 0.5])
def test_blank_patch_dedup_is_exact(monkeypatch, blank_frac):
    """IPSX_DEDUP_BLANK=1: encode non-blank patches + one blank, copy - must equal encoding every patch, bit for bit."""
    g = Golden("mnist_full")
    net = g.net(DEV)
    conf = g.conf
    x = synth.make_patches(conf, 1, seed=77, blank_frac=blank_frac, N=1237)[0]
    if 0.0 < blank_frac < 1.0:
        x[5] = -0.0                                   # negative zeros count as blank
        x[7, 0, 3, 3] = 1e-30                         # a single tiny value does not
    x = x.to(DEV)
    plan = hip.EncoderPlan(net.encoder, True)
    full = plan.encode(x)
    monkeypatch.setenv("IPSX_DEDUP_BLANK", "1")
    dd = plan.encode(x)
    n_enc = int(plan.n_encoded.item())
    monkeypatch.delenv("IPSX_DEDUP_BLANK")
    nonblank = int((x.reshape(x.shape[0], -1) != 0).any(1).sum().item())
    assert n_enc == nonblank + (1 if nonblank < x.shape[0] else 0)
    assert torch.equal(full, dd)


def test_blank_patch_dedup_is_exact_on_the_layered_trunk(monkeypatch):
    """The reference-native 50-px patches (about 85 % blank) go through the layer-by-layer trunk; the dedup there gathers
    the distinct patches with torch indexing and must still be bit-identical to encoding every patch."""
    g = Golden("mnist_native50")
    net = g.net(DEV)
    x = synth.make_patches(g.conf, 1, seed=5, blank_frac=0.85, N=333)[0].to(DEV)
    x[3] = -0.0
    plan = hip.EncoderPlan(net.encoder, True)
    full = plan.encode(x)
    assert "layer by layer" in hip.encoder_kernel_name(plan)
    monkeypatch.setenv("IPSX_DEDUP_BLANK", "1")
    dd = plan.encode(x)
    monkeypatch.delenv("IPSX_DEDUP_BLANK")
    nonblank = int((x.reshape(x.shape[0], -1) != 0).any(1).sum().item())
    assert int(plan.n_encoded.item()) == nonblank + 1 and nonblank < 100
    assert torch.equal(full, dd)
    # all blank / none blank
    monkeypatch.setenv("IPSX_DEDUP_BLANK", "1")
    z = torch.zeros_like(x[:7])
    assert torch.equal(plan.encode(z), full[3:4].expand(7, -1))
    dense = x[(x.flatten(1) != 0).any(1)][:9]
    assert torch.equal(plan.encode(dense), plan.encode_plain(dense))
    monkeypatch.delenv("IPSX_DEDUP_BLANK")


def test_bf16_trunk_tracks_fp32_within_tolerance(monkeypatch):
    """IPSX_PRECISION=bf16 (BASELINE configs[4]): bf16 operands / fp32 accumulate in the residual stages.
    The reference has no reduced-precision path; the check is against this repo's own fp32 kernel and against a
    float64 emulation that rounds weights and activations to bf16 at the same places."""
    g = Golden("mnist_full")
    net = g.net(DEV)
    x = g.patches()[0, :203].to(DEV)
    plan = hip.EncoderPlan(net.encoder, True)
    ref = plan.encode(x).double()
    monkeypatch.setenv("IPSX_PRECISION", "bf16")
    got = plan.encode(x)
    assert hip.encoder_kernel_name(plan) == "fused_trunk_bf16_kernel"
    monkeypatch.delenv("IPSX_PRECISION")
    assert torch.isfinite(got).all()
    rel = ((got.double() - ref).norm(dim=1) / ref.norm(dim=1).clamp_min(1e-12))
    assert float(rel.max()) < 3e-2, float(rel.max())

    # emulation: every conv (stem included) sees bf16-rounded input and bf16-rounded weights
    import torch.nn.functional as F
    sd = {k: v.detach().double().cpu() for k, v in net.encoder.state_dict().items()}
    r16 = lambda t: t.float().to(torch.bfloat16).double()
    def bn(y, p):
        return (y - sd[p + ".running_mean"][None, :, None, None]) / torch.sqrt(sd[p + ".running_var"] + 1e-5)[None, :, None, None] \
            * sd[p + ".weight"][None, :, None, None] + sd[p + ".bias"][None, :, None, None]
    y = F.relu(bn(F.conv2d(r16(x.double().cpu()), r16(sd["0.weight"]), None, 2, 3), "1"))
    y = F.max_pool2d(y, 3, 2, 1)
    for st, stride in ((4, 1), (5, 2)):
        for blk in (0, 1):
            p = "%d.%d" % (st, blk)
            s1 = stride if blk == 0 else 1
            idt = y
            z = F.relu(bn(F.conv2d(r16(y), r16(sd[p + ".conv1.weight"]), None, s1, 1), p + ".bn1"))
            z = bn(F.conv2d(r16(z), r16(sd[p + ".conv2.weight"]), None, 1, 1), p + ".bn2")
            if p + ".downsample.0.weight" in sd:
                idt = bn(F.conv2d(r16(y), r16(sd[p + ".downsample.0.weight"]), None, s1, 0), p + ".downsample.1")
            y = F.relu(z + idt)
    emu = y.mean(dim=(2, 3))
    err = (got.double().cpu() - emu).abs().max() / emu.abs().max()
    assert float(err) < 3e-3, float(err)      # a bf16 rounding flip of one activation is 4e-3 of that activation


@pytest.mark.parametrize("storage", [torch.float32, torch.float16])
def test_bf16_trunk_builds_agree(monkeypatch, storage):
    """Round 6: the bf16 trunk's second and third builds (csrc/fused_trunk_bf16v2.h: 8x8 stage by channel tile x patch pair,
    deep operand rings; fused_trunk_bf16v3.h, the default: eight patches per workgroup, the 4x4 stage once over all eight)
    do the FIRST build's arithmetic - same operand rounding, same products in the same order, fp32 identity: the embeddings
    are bit-identical, whole workgroups and ragged ends (1 .. 17 patches: every remainder of 4 and of 8), an index list,
    half-stored patches."""
    g = Golden("mnist_full")
    net = g.net(DEV)
    plan = hip.EncoderPlan(net.encoder, True)
    fn = hip.lib().ipsx_dbg_bf16_build
    fn.restype, fn.argtypes = None, [C.c_int]
    x_all = g.patches()[0, :1203].to(DEV).to(storage)
    monkeypatch.setenv("IPSX_PRECISION", "bf16")
    try:
        for n in (1, 2, 3, 4, 5, 7, 8, 9, 12, 13, 15, 16, 17, 203, 1203):
            x = x_all[:n].contiguous()
            fn(1)
            first = plan.encode(x)
            fn(2)
            second = plan.encode(x)
            fn(3)
            third = plan.encode(x)
            assert torch.isfinite(second).all() and torch.equal(first, second), n
            assert torch.isfinite(third).all() and torch.equal(first, third), n
        idx = torch.randperm(1203, generator=torch.Generator().manual_seed(5))[:333].to(torch.int32).to(DEV)
        outs = []
        for b in (1, 2, 3, 0):
            fn(b)
            outs.append(plan.encode_indexed(x_all, idx))
        assert all(torch.equal(outs[0], o) for o in outs[1:]) and torch.equal(outs[0], plan.encode(x_all[idx.long()].contiguous()))
    finally:
        fn(0)


def test_fp32x3_trunk_has_fp32_accuracy(monkeypatch):
    """IPSX_PRECISION=fp32x3: every fp32 operand of the residual stages as three bf16 terms, six products on the
    bf16 matrix pipe, fp32 accumulation.  Not bit-identical to the fma chains of the fp32 kernel, but as close to
    the float64 result of the same network as the exact-fp32 kernel is."""
    import torch.nn.functional as F
    g = Golden("mnist_full")
    net = g.net(DEV)
    x = g.patches()[0, :403].to(DEV)
    x[7] = 0.0
    plan = hip.EncoderPlan(net.encoder, True)
    exact = plan.encode(x).double().cpu()
    monkeypatch.setenv("IPSX_PRECISION", "fp32x3")
    got = plan.encode(x)
    assert hip.encoder_kernel_name(plan) == "fused_trunk_x3_kernel"
    again = plan.encode(x)
    monkeypatch.delenv("IPSX_PRECISION")
    assert torch.equal(got, again) and torch.isfinite(got).all()            # deterministic
    got = got.double().cpu()

    sd = {k: v.detach().double().cpu() for k, v in net.encoder.state_dict().items()}
    def bn(y, p):
        return (y - sd[p + ".running_mean"][None, :, None, None]) / torch.sqrt(sd[p + ".running_var"] + 1e-5)[None, :, None, None] \
            * sd[p + ".weight"][None, :, None, None] + sd[p + ".bias"][None, :, None, None]
    y = F.relu(bn(F.conv2d(x.double().cpu(), sd["0.weight"], None, 2, 3), "1"))
    y = F.max_pool2d(y, 3, 2, 1)
    for st, stride in ((4, 1), (5, 2)):
        for blk in (0, 1):
            p = "%d.%d" % (st, blk)
            s1 = stride if blk == 0 else 1
            idt = y
            z = F.relu(bn(F.conv2d(y, sd[p + ".conv1.weight"], None, s1, 1), p + ".bn1"))
            z = bn(F.conv2d(z, sd[p + ".conv2.weight"], None, 1, 1), p + ".bn2")
            if p + ".downsample.0.weight" in sd:
                idt = bn(F.conv2d(y, sd[p + ".downsample.0.weight"], None, s1, 0), p + ".downsample.1")
            y = F.relu(z + idt)
    truth = y.mean(dim=(2, 3))
    scale = truth.abs().max()
    err_exact = float((exact - truth).abs().max() / scale)
    err_x3 = float((got - truth).abs().max() / scale)
    mean_exact = float((exact - truth).abs().mean() / scale)
    mean_x3 = float((got - truth).abs().mean() / scale)
    print("max/mean error vs float64: fp32 kernel %.3e / %.3e, fp32x3 kernel %.3e / %.3e" % (err_exact, mean_exact, err_x3, mean_x3))
    assert err_x3 < 2e-6 and err_x3 < 3 * err_exact + 1e-7, (err_x3, err_exact)
    assert mean_x3 < 2 * mean_exact + 1e-8, (mean_x3, mean_exact)
    assert float((got - exact).abs().max() / scale) < 2e-6


@pytest.mark.parametrize("precision", ["bf16", "fp32x3"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_half_precision_patch_storage(precision, dtype, monkeypatch):
    """BASELINE configs[4]: patches STORED in bfloat16 / float16 (2 KiB per 32-px patch).  The reduced-precision trunks
    read them directly; the result must be bit-identical to handing over the same values as float32 (both half formats
    convert to float exactly), for plain and indexed launches.  The exact trunk refuses half input."""
    g = Golden("mnist_full")
    net = g.net(DEV)
    x = g.patches()[0, :300].to(DEV)
    xh = x.to(dtype)
    plan = hip.EncoderPlan(net.encoder, True)
    with pytest.raises(TypeError):
        plan.encode(xh)
    monkeypatch.setenv("IPSX_PRECISION", precision)
    want = plan.encode(xh.float())
    got = plan.encode(xh)
    assert torch.equal(got, want)
    idx = torch.arange(299, -1, -3, device=DEV, dtype=torch.int32)
    assert plan.fused(xh.shape)
    assert torch.equal(plan.encode_indexed(xh, idx), want[idx.long()])


def test_bf16_logits_track_fp32_logits(monkeypatch):
    """BASELINE configs[4]: QK^T on the bf16 matrix pipe (ipsx_logits_bf16: x = emb + pos and the folded query rounded
    to bfloat16, float32 accumulation) against the float32 logits: within 2 % of the logit scale (two operands of 8
    significant bits over a D = 128 contraction), and against a float64 evaluation of the bf16-rounded operands: 1e-5."""
    g = Golden("mnist_full")
    net = g.net(DEV)
    ca = net.transf.crs_attn
    gen = torch.Generator(device="cpu").manual_seed(3)
    emb = torch.randn((2, 333, ca.q.shape[-1]), generator=gen).to(DEV)
    pos = net.pos_enc[:, :333]
    R = ca.H * ca.n_token
    want = hip.logits(emb, pos, ca.folded_query(), R)
    monkeypatch.setenv("IPSX_PRECISION", "bf16")
    vq16 = ca.folded_query()
    assert vq16.dtype == torch.uint8
    got = hip.logits(emb, pos, vq16, R)
    monkeypatch.delenv("IPSX_PRECISION")
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) < 2e-2 * scale
    # float64 with operands rounded where the kernel rounds them
    D, H, T, Dk = emb.shape[-1], ca.H, ca.n_token, ca.D_k
    qs = hip.query_proj(ca.q[0], ca.q_w.weight, ca.attention.temperature).view(T, H, Dk)
    V = torch.einsum("thj,hjc->htc", qs.double(), ca.k_w.weight.detach().view(H, Dk, D).double()).reshape(H * T, D)
    V16 = V.float().to(torch.bfloat16).double()
    x16 = (emb + pos).to(torch.bfloat16).double()
    emu = torch.einsum("bnc,rc->bnr", x16, V16)
    assert float((got.double() - emu).abs().max()) < 1e-3 * scale      # the fp32 fold of V differs from float64 by < 1 bf16 ulp rarely


def _trunk_f64(sd, x):
    """float64 evaluation of the 2-stage ResNet-18 trunk (eval BatchNorm) from a state dict: the yardstick."""
    import torch.nn.functional as F
    sd = {k: v.detach().double().cpu() for k, v in sd.items()}

    def bn(y, p):
        return (y - sd[p + ".running_mean"][None, :, None, None]) / torch.sqrt(sd[p + ".running_var"] + 1e-5)[None, :, None, None] \
            * sd[p + ".weight"][None, :, None, None] + sd[p + ".bias"][None, :, None, None]
    y = F.relu(bn(F.conv2d(x.double().cpu(), sd["0.weight"], None, 2, 3), "1"))
    y = F.max_pool2d(y, 3, 2, 1)
    for st, stride in ((4, 1), (5, 2)):
        for blk in (0, 1):
            p = "%d.%d" % (st, blk)
            s1 = stride if blk == 0 else 1
            idt = y
            z = F.relu(bn(F.conv2d(y, sd[p + ".conv1.weight"], None, s1, 1), p + ".bn1"))
            z = bn(F.conv2d(z, sd[p + ".conv2.weight"], None, 1, 1), p + ".bn2")
            if p + ".downsample.0.weight" in sd:
                idt = bn(F.conv2d(y, sd[p + ".downsample.0.weight"], None, s1, 0), p + ".downsample.1")
            y = F.relu(z + idt)
    return y.mean(dim=(2, 3))


def _pass_through_after(enc, n_block):
    """Make every residual block after the first ``n_block`` a pass-through (second convolution zero, BatchNorm
    of the shortcut an exact identity), so the embedding is the spatial mean of what block ``n_block`` produced -
    a probe of THAT layer's output through arithmetic that adds no error of its own (x * 1 and x + 0 are exact in
    both the fma-chain and the split-product arithmetic)."""
    blocks = [enc[4][0], enc[4][1], enc[5][0], enc[5][1]]
    with torch.no_grad():
        for b in blocks[n_block:]:
            b.conv2.weight.zero_()
            b.bn2.weight.fill_(1.0); b.bn2.bias.zero_(); b.bn2.running_mean.zero_(); b.bn2.running_var.fill_(1.0 - 1e-5)
            if b.downsample is not None:
                w = b.downsample[0].weight
                w.zero_()
                for c in range(w.shape[1]):
                    w[c, c, 0, 0] = 1.0
                bn = b.downsample[1]
                bn.weight.fill_(1.0); bn.bias.zero_(); bn.running_mean.zero_(); bn.running_var.fill_(1.0 - 1e-5)


@pytest.mark.parametrize("probe", [0, 1, 2, 3, 4])
def test_fp32x3_accuracy_under_adversarial_ranges_layer_by_layer(probe, monkeypatch):
    """The evidence behind fp32x3's status: not one pooled embedding on benign data, but every stage's
    output (probed through an exact pass-through tail: 0 = stem + pool, k = after residual block k) on data built to
    hurt a split-product scheme - weights whose output channels span 2^-6 .. 2^6, BatchNorm scales and variances
    spanning 2^-5 .. 2^5 / 2^-8 .. 2^8 so that activations of neighbouring channels differ by tens of binades and the
    spread compounds from layer to layer, inputs spanning 1e-6 .. 1e3 with mixed signs, and a band of patches scaled by
    1e-20 (activations far below 1, towards the range where the low term of the split meets bf16's underflow; channels
    whose magnitude ends up below 1e-30 are outside what the scheme promises, DESIGN 6d, and are not judged).
    Yardstick: float64.  Requirement: per output channel, the error of fp32x3 relative to that channel's magnitude stays
    within 3x the exact-fp32 kernel's own error (+ 2 ulp) on dense patches - the same accuracy class everywhere, not on
    average.  Measured limit of that statement (why fp32x3 stays opt-in): on nearly blank patches, where an output is
    the small remainder of convolution sums and BatchNorm shifts that cancel, the truncated cross terms of the split
    do not average out the way fma roundings do and its error reaches ~10x the exact kernel's (bounded here at 16x)."""
    g = Golden("mnist_full")
    net = g.net(DEV)
    enc = net.encoder
    gen = torch.Generator(device="cpu").manual_seed(100 + probe)
    with torch.no_grad():
        for name, p in enc.named_parameters():
            if p.dim() == 4:                                       # output channels over 24 binades
                sc = torch.exp2(torch.randint(-6, 7, (p.shape[0], 1, 1, 1), generator=gen).float())
                p.mul_(sc.to(DEV))
        for m in enc.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                sc = torch.exp2(torch.randint(-5, 6, (m.num_features,), generator=gen).float()).to(DEV)
                m.weight.mul_(sc)
                m.running_var.mul_(torch.exp2(torch.randint(-8, 9, (m.num_features,), generator=gen).float()).to(DEV))
    _pass_through_after(enc, probe)
    n = 96
    x = torch.randn((n, 1, 32, 32), generator=gen) * torch.pow(10.0, torch.rand((n, 1, 32, 32), generator=gen) * 9 - 6)
    x[64:80] *= 1e-20                                               # activations around 1e-26 .. 1e-17 from the stem on
    x[80:88] = x[80:88].abs()
    x[88:] = 0.0
    x[88:, :, 5:9, 7:12] = 3.0e-5
    x = x.to(DEV)
    plan = hip.EncoderPlan(enc, True)
    exact = plan.encode(x).double().cpu()
    monkeypatch.setenv("IPSX_PRECISION", "fp32x3")
    got = plan.encode(x).double().cpu()
    assert hip.encoder_kernel_name(plan) == "fused_trunk_x3_kernel"
    monkeypatch.delenv("IPSX_PRECISION")
    truth = _trunk_f64(enc.state_dict(), x)
    assert torch.isfinite(truth).all() and torch.isfinite(got).all()
    # per (patch group, channel) scale: the largest magnitude that channel takes within patches of similar input scale
    worst = []
    for lo, hi in ((0, 64), (64, 80), (80, 88), (88, 96)):
        t = truth[lo:hi]
        scale = t.abs().amax(dim=0).clamp_min(1e-300)
        keep = t.abs().amax(dim=0) > 1e-30                          # channels that are alive (and in range) in this group
        if not bool(keep.any()):
            continue
        e_fp32 = ((exact[lo:hi] - t).abs() / scale)[:, keep].max()
        e_x3 = ((got[lo:hi] - t).abs() / scale)[:, keep].max()
        worst.append((float(e_fp32), float(e_x3)))
        # dense patches: 3x; the nearly blank group, whose outputs are what is left after the BatchNorm shifts cancel the
        # convolution sums (the exact kernel itself is 100x its usual error there): an order of magnitude
        bound = 3 if lo < 88 else 16
        assert float(e_x3) <= bound * float(e_fp32) + 2.4e-7, (probe, lo, float(e_x3), float(e_fp32))
    print("probe %d: max channel-relative error vs float64 per patch group (fp32, fp32x3): %s"
          % (probe, ", ".join("%.2e / %.2e" % w for w in worst)))


def test_encoder_plan_tracks_weight_updates():
    g = Golden("mnist_mini")
    net = g.net(DEV)
    x = g.patches()[0, :8].to(DEV)
    plan = hip.EncoderPlan(net.encoder, True)
    a = plan.encode(x).clone()
    with torch.no_grad():
        net.encoder[0].weight.mul_(1.5)              # what optimizer.step() does: in-place update
        net.encoder[1].running_mean.add_(0.05)       # what a training-mode forward does
    b = plan.encode(x)
    assert not torch.equal(a, b)
    want = orc.Oracle(net.cpu()).encode(x.cpu().numpy())
    assert ulp_diff(b.cpu().numpy(), want) == 0


@pytest.mark.parametrize("case", ["mnist_mini", "mnist_tok1", "cam_b2", "traffic_tiny"])
def test_logits_scores_attn_bit_exact(case):
    g = Golden(case)
    net = g.net(DEV)
    o = orc.Oracle(g.net("cpu"))
    ca = net.transf.crs_attn
    D = g.conf.D
    B, L = 2, 53
    x = rnd((B, L, D), 21)
    qs = ca.scaled_query()
    assert ulp_diff(qs.cpu().numpy(), o.qs) == 0
    pos = rnd((B, L, D), 22)
    vq, R = ca.folded_query(), ca.H * ca.n_token                      # the query folded into the key weights
    got = hip.logits(dev(x), dev(pos), vq, R).cpu().numpy()
    for b in range(B):
        assert ulp_diff(got[b], o.logits(x[b], pos[b])) == 0
    # broadcast positional table (batch stride 0) and no table at all
    got = hip.logits(dev(x), dev(pos[:1]), vq, R).cpu().numpy()
    assert ulp_diff(got[1], o.logits(x[1], pos[0])) == 0
    got = hip.logits(dev(x), None, vq, R).cpu().numpy()
    assert ulp_diff(got[1], o.logits(x[1])) == 0
    # Transformer.get_scores / get_attn through the module API
    with torch.no_grad():
        sc = net.transf.get_scores(dev(x)).cpu().numpy()
        attn = ca.get_attn(dev(x)).cpu().numpy()
    for b in range(B):
        ws, wa = o.scores(x[b], want_attn=True)
        assert ulp_diff(sc[b], ws) == 0
        assert ulp_diff(attn[b], wa) == 0


@pytest.mark.parametrize("L,M", [(128, 64), (48, 16), (512, 256), (600, 100), (2000, 64), (7, 7)])
def test_topm_matches_oracle_and_torch(L, M):
    g = np.random.default_rng(L)
    s = np.stack([g.permutation(L), g.permutation(L)]).astype(np.float32) / L
    top = hip.topm(dev(s), M).cpu().numpy()
    for b in range(2):
        assert np.array_equal(top[b], orc.topm(s[b], M)[0])
        assert np.array_equal(top[b], torch.topk(torch.from_numpy(s[b]), M)[1].numpy())
    # exact ties, canonical rule = earlier position first
    t = np.zeros((1, L), dtype=np.float32)
    t[0, ::3] = 1.0
    assert hip.set_tie_order("canonical") == "torch"          # the default is the reference's order
    try:
        top = hip.topm(dev(t), M).cpu().numpy()[0]
    finally:
        hip.set_tie_order("torch")
    assert np.array_equal(top, orc.topm(t[0], M)[0])


@pytest.mark.parametrize("L,M", [(128, 64), (48, 16), (512, 256), (600, 100), (2000, 64), (7, 7), (1116, 16), (130, 2),
                                 (300, 299), (1024, 1000)])
def test_topm_under_ties_returns_torch_cpu_order(L, M):
    """Equal scores: torch.topk on CPU returns what libstdc++'s nth_element + sort / partial_sort leave behind
    (SURVEY H2); the kernels replay those routines (csrc/ipsx_stdorder.h).  Checked against torch itself (the host
    of the GPU box runs the same ATen CPU code as the reference) and against the oracle's restatement."""
    g = np.random.default_rng(L * 7 + M)
    rows = []
    for distinct in (1, 2, 3, 7, 40):                         # from "everything equal" to "a few repeated values"
        rows.append(g.integers(0, distinct, L).astype(np.float32) * 0.125)
    r = g.permutation(L).astype(np.float32)
    r[g.integers(0, L, L // 3)] = r[0]                        # mostly distinct, one value repeated
    rows.append(r)
    n = g.standard_normal(L).astype(np.float32).round(1)
    n[g.integers(0, L, 5)] = np.nan                           # NaNs rank first and tie with each other
    rows.append(n)
    s = np.stack(rows)
    top = hip.topm(dev(s), M).cpu().numpy()
    for b in range(len(s)):
        want = torch.topk(torch.from_numpy(s[b]), M)[1].numpy()
        assert np.array_equal(top[b], want), (b, top[b][:12], want[:12])
        assert np.array_equal(top[b], orc.topm(s[b], M, aten_ties=True)[0])


@pytest.mark.parametrize("seed", range(6))
def test_wavefront_tie_replay_equals_torch_cpu_on_many_random_rows(seed):
    """The tie replay runs on a whole wavefront (Hoare partition by ballots, introsort leaves sorted by a lane each:
    csrc/scorer.hip torch_topk_wave).  Hundreds of tie-heavy rows per shape - few distinct values, sorted / reversed /
    organ-pipe patterns that reach the depth-limit heap paths, NaNs, negative zeros - against torch.topk on the CPU."""
    g = np.random.default_rng(500 + seed)
    for L, M in ((512, 256), (128, 64), (1000, 333), (1024, 1023), (96, 95), (200, 17), (700, 11)):
        rows = []
        for _ in range(40):
            distinct = int(g.choice([1, 2, 3, 5, 37, 1000]))
            pattern = int(g.integers(0, 4))
            i = np.arange(L)
            if pattern == 0:
                v = g.integers(0, distinct, L).astype(np.float32) * 0.25
            elif pattern == 1:
                v = (i * distinct // L).astype(np.float32)
            elif pattern == 2:
                v = ((L - 1 - i) * distinct // L).astype(np.float32)
            else:
                v = (np.minimum(i, L - 1 - i) % distinct).astype(np.float32)
            if g.integers(0, 9) == 0:
                v[g.integers(0, L, max(1, L // 7))] = np.nan
            neg = g.integers(0, 31, L) == 0
            v[neg] = -v[neg]
            rows.append(v)
        s = np.stack(rows)
        top = hip.topm(dev(s), M).cpu().numpy()
        for b in range(len(s)):
            want = torch.topk(torch.from_numpy(s[b]), M)[1].numpy()
            assert np.array_equal(top[b], want), (L, M, b, top[b][:12], want[:12])


@pytest.mark.parametrize("N,M,I,H,T", [(300, 16, 16, 8, 4), (301, 16, 24, 8, 4), (40, 16, 64, 8, 4),
                                       (1000, 32, 48, 8, 1), (2500, 64, 64, 8, 4), (3000, 256, 256, 8, 1),
                                       (5000, 900, 900, 8, 1)])
def test_scan_matches_oracle(N, M, I, H, T):
    """The persistent scan kernel against the reference-faithful loop of the oracle."""
    B, R = 2, H * T
    lg = rnd((B, N, R), N + M, 3.0)
    mem, sc = hip.scan(dev(lg), M, I, H, T, want_scores=True)
    mem, sc = mem.cpu().numpy(), sc.cpu().numpy()
    L = orc.lib()
    for b in range(B):
        cur = np.arange(M, dtype=np.int64)
        for lo in range(M, N, I):
            cand = np.concatenate([cur, np.arange(lo, min(lo + I, N), dtype=np.int64)])
            s = np.empty(len(cand), dtype=np.float32)
            L.orc_scores_from_logits(orc._f(lg[b][cand])[1], len(cand), H, T, s.ctypes.data_as(orc.f32p), None)
            top, tie = orc.topm(s, M)
            cur, last = cand[top], s[top]
        assert np.array_equal(mem[b], cur)
        assert ulp_diff(sc[b], last) == 0


def _oracle_scan_from_logits(lg, M, I, H, T, aten_ties):
    N = lg.shape[0]
    L = orc.lib()
    cur = np.arange(M, dtype=np.int64)
    for lo in range(M, N, I):
        cand = np.concatenate([cur, np.arange(lo, min(lo + I, N), dtype=np.int64)])
        s = np.empty(len(cand), dtype=np.float32)
        L.orc_scores_from_logits(orc._f(lg[cand])[1], len(cand), H, T, s.ctypes.data_as(orc.f32p), None)
        cur = cand[orc.topm(s, M, aten_ties=aten_ties, rows=lg[cand])[0]]     # (the LOOP's tie rule: orc_topm_loop)
    return cur


@pytest.mark.parametrize("seed", range(12))
def test_scan_random_shapes_with_ties_matches_oracle(seed):
    """Random loop shapes (ragged chunks, odd head / token counts -> the general score path, candidate sets beyond
    1024 -> the bitonic ranking) on QUANTISED logits, so that equal scores occur in most iterations: the default
    tie order must follow torch.topk's CPU order (oracle: orc_topm_aten), the canonical one the position rule; a
    scan cut into resumed ranges must give the same answer."""
    g = np.random.default_rng(1000 + seed)
    H = int(g.choice([1, 2, 3, 8]))
    T = int(g.choice([1, 2, 3, 4]))
    M = int(g.choice([4, 16, 33, 64, 200]))
    I = int(g.choice([5, 16, 64, 100, 900]))
    N = M + int(g.integers(1, 6)) * I + int(g.integers(0, I))
    B, R = 2, H * T
    levels = int(g.choice([2, 3, 5, 1000]))
    lg = (g.integers(0, levels, (B, N, R)).astype(np.float32) - 1.0) * np.float32(0.75)
    if levels <= 5:
        lg[:, :, 1:] = lg[:, :, :1]                       # every row the same across (h, t): whole candidates tie
    for mode, aten in (("torch", True), ("canonical", False)):
        hip.set_tie_order(mode)
        try:
            mem = hip.scan(dev(lg), M, I, H, T).cpu().numpy()
            n_iter = -(-(N - M) // I)
            cut = max(1, n_iter // 2)
            idx = torch.empty((B, M), dtype=torch.int64, device=DEV)
            tie = torch.zeros((B,), dtype=torch.int32, device=DEV)
            hip.scan_range(dev(lg), M, I, H, T, 0, cut, idx, tie)
            hip.scan_range(dev(lg), M, I, H, T, cut, n_iter, idx, tie)
        finally:
            hip.set_tie_order("torch")
        for b in range(B):
            want = _oracle_scan_from_logits(lg[b], M, I, H, T, aten)
            assert np.array_equal(mem[b], want), (mode, H, T, M, I, N, levels)
        assert np.array_equal(idx.cpu().numpy(), mem)


@pytest.mark.parametrize("seed", range(6))
def test_scan_cam_shape_with_ties_matches_oracle(seed):
    """The specialised loop of BASELINE configs[3] (scan_cam_kernel: 8 heads, one token, M = I = 256) on QUANTISED logits
    - equal scores in most iterations: its 32-bit ranking must notice them and hand over to the 64-bit ranking and the
    replay of torch.topk's order - against the oracle, in both tie orders, ragged last chunk, cut into resumed ranges,
    and with 1..4 slides per call."""
    g = np.random.default_rng(4000 + seed)
    M = I = 256
    H, T = 8, 1
    B = int(g.integers(1, 5))
    N = M + int(g.integers(2, 9)) * I + int(g.integers(0, I))
    levels = int(g.choice([2, 3, 5, 17, 1000]))
    lg = (g.integers(0, levels, (B, N, H)).astype(np.float32) - 1.0) * np.float32(0.75)
    if levels <= 5 and seed % 2 == 0:
        lg[:, :, 1:] = lg[:, :, :1]                       # whole candidates tie
    for mode, aten in (("torch", True), ("canonical", False)):
        hip.set_tie_order(mode)
        try:
            mem = hip.scan(dev(lg), M, I, H, T).cpu().numpy()
            n_iter = -(-(N - M) // I)
            cut = max(1, n_iter // 2)
            idx = torch.empty((B, M), dtype=torch.int64, device=DEV)
            tie = torch.zeros((B,), dtype=torch.int32, device=DEV)
            hip.scan_range(dev(lg), M, I, H, T, 0, cut, idx, tie)
            hip.scan_range(dev(lg), M, I, H, T, cut, n_iter, idx, tie)
        finally:
            hip.set_tie_order("torch")
        for b in range(B):
            want = _oracle_scan_from_logits(lg[b], M, I, H, T, aten)
            assert np.array_equal(mem[b], want), (mode, B, N, levels)
        assert np.array_equal(idx.cpu().numpy(), mem)


@pytest.mark.parametrize("N,M,I,H,T,levels", [
    (38000, 5000, 5000, 8, 1, 0),        # the reference's shipped CAMELYON sizes (config/camelyon_config.yml:35-36), ragged end
    (21000, 5000, 5000, 8, 1, 4096),     # quantised logits: equal scores in every iteration -> torch.topk's order replayed
    (20000, 3000, 6000, 8, 4, 0),        # 32 rows per patch, 9,000 candidates
    (40000, 8192, 8192, 2, 1, 0),        # the largest supported candidate set: 16,384
    (30000, 100, 9000, 8, 1, 64),        # 64 * M <= L: torch.topk takes partial_sort (heap select) under ties
    (9000, 4200, 300, 3, 3, 0),          # odd row count, many short chunks on a large memory
    (24000, 5000, 5000, 8, 1, -6),       # a FEW equal pairs among 10,000 candidates (6 duplicated rows): the replay copies the
    (26000, 3000, 7000, 8, 1, -40),      #   tie-free ranges of std::sort from the canonical ranking and follows only the
    (20000, 6000, 4000, 8, 1, -1),       #   ranges on the way to the pairs (-n: n rows duplicated)
])
def test_scan_beyond_the_lds_matches_oracle(N, M, I, H, T, levels):
    """Candidate sets that do not fit one compute unit's LDS (M + I up to 16,384: scan_large_kernel - ranking in LDS,
    the rest through the caller's workspace) against the oracle's loop, torch.topk's tie order included; a run cut into
    resumed ranges gives the same memory."""
    B, R = 2, H * T
    if levels > 0:
        g = np.random.default_rng(N + M)
        lg = (g.integers(0, levels, (B, N, R)).astype(np.float32) - np.float32(levels / 2)) * np.float32(6.0 / levels)
        if levels <= 64:
            lg[:, :, 1:] = lg[:, :, :1]
    else:
        lg = rnd((B, N, R), N + M, 3.0)
        if levels < 0:                                    # duplicated rows: bit-equal scores for those pairs, nothing else
            g = np.random.default_rng(N - M)
            for b in range(B):
                src = g.integers(0, N, -levels)
                dst = g.integers(0, N, -levels)
                lg[b, dst] = lg[b, src]
    assert hip.lib().ipsx_scan_workspace_bytes(B, M, I, H, T) > 0
    mem, sc = hip.scan(dev(lg), M, I, H, T, want_scores=True)
    mem, sc = mem.cpu().numpy(), sc.cpu().numpy()
    n_iter = -(-(N - M) // I)
    idx = torch.empty((B, M), dtype=torch.int64, device=DEV)
    tie = torch.zeros((B,), dtype=torch.int32, device=DEV)
    cut = max(1, n_iter // 2)
    hip.scan_range(dev(lg), M, I, H, T, 0, cut, idx, tie)
    hip.scan_range(dev(lg), M, I, H, T, cut, n_iter, idx, tie)
    L = orc.lib()
    for b in range(B):
        cur = np.arange(M, dtype=np.int64)
        for lo in range(M, N, I):
            cand = np.concatenate([cur, np.arange(lo, min(lo + I, N), dtype=np.int64)])
            s = np.empty(len(cand), dtype=np.float32)
            L.orc_scores_from_logits(orc._f(lg[b][cand])[1], len(cand), H, T, s.ctypes.data_as(orc.f32p), None)
            top = orc.topm(s, M, aten_ties=True, rows=lg[b][cand])[0]          # (the LOOP's tie rule: orc_topm_loop)
            cur, last = cand[top], s[top]
        assert np.array_equal(mem[b], cur), (b, int((mem[b] != cur).sum()), np.nonzero(mem[b] != cur)[0][:8])
        assert ulp_diff(sc[b], last) == 0
    assert np.array_equal(idx.cpu().numpy(), mem)


@pytest.mark.parametrize("L,M", [(10000, 5000), (16384, 8000), (9000, 100), (12345, 12345), (8200, 1)])
def test_topm_beyond_the_lds_is_torch_topk(L, M):
    """ipsx_topm for rows that do not fit the LDS twice over (one key array + workspace) == torch.topk on the host,
    with and without exact ties (both of ATen's branches)."""
    g = np.random.default_rng(L + M)
    rows = [g.standard_normal(L).astype(np.float32),
            (g.integers(0, 50, L)).astype(np.float32),                      # heavy ties
            np.where(g.integers(0, 40, L) == 0, np.float32(np.nan), g.integers(0, 3000, L).astype(np.float32))]
    s = np.stack(rows)
    top = hip.topm(dev(s), M).cpu().numpy()
    for b in range(len(s)):
        want = torch.topk(torch.from_numpy(s[b]), M)[1].numpy()
        assert np.array_equal(top[b], want), (L, M, b, int((top[b] != want).sum()))


def test_gather_rows():
    B, N, M = 3, 50, 7
    src = rnd((B, N, 1, 32, 32), 30)
    idx = np.random.default_rng(31).integers(0, N, (B, M))
    got = hip.gather_rows(dev(src), dev(idx)).cpu().numpy()
    assert np.array_equal(got, np.stack([src[b][idx[b]] for b in range(B)]))
    tab = rnd((1, N, 130), 32)                       # shared table, row not a multiple of 16 bytes
    got = hip.gather_rows(dev(tab).expand(B, -1, -1), dev(idx)).cpu().numpy()
    assert np.array_equal(got, np.stack([tab[0][idx[b]] for b in range(B)]))


@pytest.mark.parametrize("case", ["mnist_mini", "cam_b2", "traffic_tiny"])
def test_aggregate_and_heads_bit_exact(case):
    g = Golden(case)
    net = g.net(DEV)
    o = orc.Oracle(g.net("cpu"))
    B, M, D = 3, g.conf.M, g.conf.D
    x = rnd((B, M, D), 40)
    with torch.no_grad():
        got = net.transf(dev(x)).cpu().numpy()
    want = o.aggregate(x)
    assert ulp_diff(got, want) == 0, "max abs diff %g" % np.abs(got - want).max()
