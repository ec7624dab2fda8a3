"""CPU: the host-side scheduling of ips_amd/selection.py - launch sizes derived from the device geometry (no GPU needed:
the geometry is injected), parts cut at chunk boundaries, and the ABI pieces round 4 added."""
import math

import pytest
import torch

from ips_amd import hip, synth
from ips_amd.architecture import IPSNet
from ips_amd.selection import Selection


class _Dev:
    """stands in for a torch device in the geometry look-up"""
    index = 0


@pytest.fixture
def geometry(monkeypatch):
    def use(cus, xcds):
        monkeypatch.setattr(hip, "device_geometry", lambda dev: hip.Geometry(cus, xcds, cus // xcds))
    return use


def _sel(conf):
    return Selection(IPSNet(torch.device("cpu"), conf))


def test_small_batch_split_follows_the_device(geometry):
    sel = _sel(synth.mnist_conf(N=2500, M=64, I=64))
    geometry(256, 8)                                   # MI355X, SPX
    assert sel.round_patches(_Dev) == 2048 and sel.small_batch_limit(_Dev) == 32768
    edges, its = sel.small_batch_split(1, 2500, _Dev)
    assert edges == [0, 1024, 2016, 2500] and its == [0, 15, 30, 39]          # 1,024 + 992 + 484 patches, 15 + 15 + 9 iterations
    assert sel.small_batch_split(16, 2500, _Dev) is None                      # 19 rounds: the four shrinking parts instead
    geometry(128, 4)                                   # a DPX partition: half the units, the same shape of split
    edges, its = sel.small_batch_split(1, 1250, _Dev)
    assert edges[1] == 512 and edges[2] == 512 + 512 - 16 and edges[-1] == 1250
    assert its[-1] == math.ceil((1250 - 64) / 64)


def test_feature_launches_leave_a_unit_per_loop_on_the_fullest_xcd(geometry):
    sel = _sel(synth.camelyon_conf(N=65536, M=256, I=256))
    geometry(256, 8)
    assert sel.free_units(_Dev, 1) == 248 and sel.free_units(_Dev, 8) == 248 and sel.free_units(_Dev, 9) == 240
    its = sel.feature_parts(1, 65536, _Dev, True)
    assert its[0] == 0 and its[-1] == 255 and all(b > a for a, b in zip(its, its[1:]))
    edges = [0] + [min(65536, 256 + it * 256) for it in its[1:]]
    edges[-1] = 65536
    one = sel.feature_launches(1, 65536, edges, _Dev)
    assert [r1 - r0 for r0, r1, _ in one] == [b - a for a, b in zip(edges, edges[1:])]
    assert all(r1 - r0 <= 248 * 64 for r0, r1, _ in one)
    # several slides: one flat stream of rows cut into full launches that may span a slide's end; every row exactly once,
    # every slide published up to its last row
    flat = sel.feature_launches(4, 65536, edges, _Dev)
    assert flat[0][0] == 0 and flat[-1][1] == 4 * 65536 and all(a[1] == b[0] for a, b in zip(flat, flat[1:]))
    assert all(r1 - r0 <= 224 * 64 for r0, r1, _ in flat)
    last = {}
    for _, _, pubs in flat:
        for slide, rows in pubs:
            assert rows >= last.get(slide, 0)
            last[slide] = rows
    assert last == {0: 65536, 1: 65536, 2: 65536, 3: 65536}
    geometry(32, 1)                                    # a CPX partition: one XCD
    assert sel.free_units(_Dev, 1) == 31


def test_resident_loops_two_for_any_number_of_slides_and_none_on_a_tiny_partition(geometry, monkeypatch):
    sel = _sel(synth.camelyon_conf(N=65536, M=256, I=256))             # the shape whose loop kernel takes slides in turn
    assert [sel.feature_loops(b) for b in (1, 2, 3, 16)] == [1, 2, 2, 2]
    assert [_sel(synth.camelyon_conf(N=8192, M=64, I=64)).feature_loops(b) for b in (1, 2, 3, 16)] == [1, 2, 3, 16]
    monkeypatch.setattr(hip, "persistent_ok", lambda dev: True)        # (the self-test needs the GPU)
    geometry(256, 8)
    assert sel.persistent_allowed(_Dev, 256, 256, 8, 1, 2) and sel.persistent_allowed(_Dev, 64, 64, 8, 1, 16)
    geometry(64, 2)
    assert sel.persistent_allowed(_Dev, 64, 64, 8, 1, 8) and not sel.persistent_allowed(_Dev, 64, 64, 8, 1, 16)
    geometry(4, 1)                                                     # no unit to spare: the per-part launches
    assert not sel.persistent_allowed(_Dev, 256, 256, 8, 1, 1)


def test_abi_major_and_the_round_4_additions_are_declared():
    lib = hip.lib()
    assert lib.ipsx_version() // 100 == hip.ABI_MAJOR
    for name in ("ipsx_aggregate_packed", "ipsx_set_persistent_wait_ms", "ipsx_conv2d_wgrad_nhwc", "ipsx_pack_conv_weight_strided",
                 "ipsx_conv2d_affine_to_nhwc", "ipsx_scan_persistent_on", "ipsx_scan_persistent_groupable",
                 "ipsx_scan_persistent_ws", "ipsx_scan_range_if_ws"):
        assert name in hip._EXPORTS and hasattr(lib, name)
    assert lib.ipsx_scan_persistent_groupable(256, 256, 8, 1) == 1 and lib.ipsx_scan_persistent_groupable(64, 64, 8, 4) == 0
    prev = lib.ipsx_set_persistent_wait_ms(0)          # query
    assert prev == 50
    assert lib.ipsx_set_persistent_wait_ms(80) == 50 and lib.ipsx_set_persistent_wait_ms(50) == 80
    assert lib.ipsx_conv2d_wgrad_nhwc_supported(64, 128, 3, 3, 2, 1) == 1 and lib.ipsx_conv2d_wgrad_nhwc_supported(1, 64, 7, 7, 2, 3) == 1
    assert lib.ipsx_conv2d_wgrad_nhwc_supported(3, 64, 7, 7, 2, 3) == 0 and lib.ipsx_conv2d_wgrad_nhwc_supported(64, 96, 3, 3, 1, 1) == 0
    assert lib.ipsx_conv2d_wgrad_nhwc_workspace_bytes(0, 64, 64, 3, 3) == 0


def test_round_6_additions_team_width_and_the_words_a_call_has_to_clear():
    """ABI 3.02 (no GPU needed: both answers are host arithmetic): the loop of candidate sets beyond the LDS keeps a TEAM of
    compute units per image (csrc/scan_large_team.h) - 16 units of a call at most, its block is part of the workspace - and
    only the head of the projector stream's control words has to be zero when a call starts."""
    lib = hip.lib()
    for name in ("ipsx_scan_workgroups_per_image", "ipsx_projector_stream_ctl_zero_words"):
        assert name in hip._EXPORTS and hasattr(lib, name)
    team = lib.ipsx_scan_workgroups_per_image
    assert [team(b, 5000, 5000, 8, 1) for b in (1, 2, 3, 4, 8, 9, 16)] == [8, 8, 4, 4, 2, 1, 1]
    assert team(1, 256, 256, 8, 1) == 1 and team(1, 64, 64, 8, 4) == 1          # LDS-resident loops
    assert team(1, 3000, 6000, 8, 4) == 1 and team(1, 8192, 8192, 2, 1) == 1    # beyond the LDS, but not 8 heads x one token
    assert team(1, 2000, 2000, 8, 1) == 1                                       # 4,000 candidates
    assert team(0, 5000, 5000, 8, 1) == 0
    ws = lib.ipsx_scan_workspace_bytes
    lp, n2 = 10048, 16384
    assert ws(1, 5000, 5000, 8, 1) >= 8 * lp * 4 + 2 * lp * 4 + 2048 + n2 * 8 + 5000 * 8     # ... + counters, runs, second memory buffer
    assert ws(3, 5000, 5000, 8, 1) == 3 * ws(1, 5000, 5000, 8, 1) and ws(1, 256, 256, 8, 1) == 0
    words, zero = lib.ipsx_projector_stream_ctl_words, lib.ipsx_projector_stream_ctl_zero_words
    for n in (64, 38000, 65536, 16 * 65536):
        assert 0 < zero(n) <= words(n) and zero(n) < n // 32 + 1024             # a flag per 32 rows and a few hundred words
        assert words(n) - zero(n) == words(64) - zero(64)                       # the hand-over accumulators: a fixed block
    assert zero(0) == 0


def test_plan_signature_sees_nested_module_swaps_and_new_entries():
    """EncoderPlan._signature (CPU: it only reads pointers and version counters): in-place writes, replaced tensors, a child
    module exchanged BELOW the top level, a buffer that was None and appears, an entry that is removed - each changes the
    signature (the advisor's round-4 finding: a nested swap after the first ips() kept stale packed weights)."""
    net = IPSNet(torch.device("cpu"), synth.mnist_conf(N=200, M=16, I=16)).eval()
    plan = hip.EncoderPlan(net.encoder, True)
    s0 = plan._signature()
    assert plan._signature() == s0
    with torch.no_grad():
        net.encoder[4][0].conv1.weight.mul_(2.0)                   # in place: the version counter moves
    s1 = plan._signature()
    assert s1 != s0
    old_bn = net.encoder[5][1].bn2
    net.encoder[5][1].bn2 = torch.nn.BatchNorm2d(old_bn.num_features)   # a nested child replaced
    s2 = plan._signature()
    assert s2 != s1 and plan._signature() == s2
    bn = net.encoder[1]
    keep = bn.running_mean
    bn.running_mean = None                                          # an entry goes away (stays in the dict as None) ...
    s3 = plan._signature()
    assert s3 != s2
    bn.running_mean = keep                                          # ... and one that was None appears
    s4 = plan._signature()
    assert s4 != s3
    del net.encoder[4][1].bn1._buffers["num_batches_tracked"]       # a key removed outright: no KeyError, a new signature
    assert plan._signature() != s4


def test_bench_py_parent_starts_the_ranks_and_hands_their_failure_on():
    """``python bench.py --gpus 2`` with no launcher environment starts its own ranks as a child process; here (no GPU) the
    ranks refuse to run, and the parent's exit code says so.  (The same command on a GPU: tests/test_hip_e2e.py.)"""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.is_available():
        pytest.skip("the GPU variant of this test runs the real thing")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    out = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                         cwd=repo, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert out.stderr.count("bench.py needs a GPU") >= 1 and "2-rank child run exited" in out.stderr
