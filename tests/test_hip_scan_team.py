"""The selection loop for candidate sets beyond the LDS as a TEAM of workgroups per image (csrc/scan_large_team.h: the
reference's shipped CAMELYON sizes, config/camelyon_config.yml:35-36; the loop of architecture/ips_net.py:213-241) against
the one-workgroup kernel it replaces - which tests/test_hip_kernels.py::test_scan_beyond_the_lds_matches_oracle holds
against the oracle (and, with the team being the default, holds the team against it too): indices, final scores, tie
flags, resumed ranges, for every team width, with the ranking taken from the runs' top halves and from whole runs."""

import numpy as np
import pytest
import torch

from ips_amd import hip

pytestmark = pytest.mark.gpu


def _logits(B, N, kind, seed):
    g = np.random.default_rng(seed)
    if kind == "blocks":
        # every 8th 64-block of patches scores high: in iteration 0 ONE workgroup of a team of 8 holds the whole top - the
        # ranking from the runs' top halves must notice (the (m + 1)-th score is not above what was left out) and merge all
        lg = g.standard_normal((B, N, 8)).astype(np.float32)
        lg[:, ((np.arange(N) >> 6) & 7) == 0] += np.float32(6.0)
        return lg
    if kind == "sorted":
        # scores fall with the patch index: the memory never changes, every chunk loses as a whole
        return (np.linspace(4.0, -4.0, N, dtype=np.float32)[None, :, None] + g.standard_normal((B, N, 8)).astype(np.float32) * np.float32(1e-3))
    if kind > 0:                                       # quantised: equal scores everywhere -> torch.topk's order replayed
        return (g.integers(0, kind, (B, N, 8)).astype(np.float32) - np.float32(kind / 2)) * np.float32(6.0 / kind)
    lg = (g.standard_normal((B, N, 8)) * 3.0).astype(np.float32)
    for b in range(B):                                 # -kind duplicated rows: a few bit-equal pairs, nothing else
        src, dst = g.integers(0, N, -kind), g.integers(0, N, -kind)
        lg[b, dst] = lg[b, src]
    return lg


def _run(lg, M, I, cut=None):
    B, N = lg.shape[:2]
    n_iter = -(-(N - M) // I)
    if cut is None:
        mem, sc = hip.scan(lg, M, I, 8, 1, want_scores=True)
        return mem.cpu().numpy(), sc.cpu().numpy().view(np.int32), hip.scan.last_tie.cpu().numpy()
    idx = torch.empty((B, M), dtype=torch.int64, device=lg.device)
    tie = torch.zeros((B,), dtype=torch.int32, device=lg.device)
    hip.scan_range(lg, M, I, 8, 1, 0, cut, idx, tie)
    hip.scan_range(lg, M, I, 8, 1, cut, n_iter, idx, tie)
    return idx.cpu().numpy(), None, tie.cpu().numpy()


@pytest.mark.parametrize("B,N,M,I,kind", [
    (1, 38000, 5000, 5000, 0),            # the shipped sizes, ragged end (3,000 candidates in the last chunk)
    (2, 21000, 5000, 5000, 4096),
    (1, 24000, 5000, 5000, -6),
    (3, 40000, 8192, 8192, 0),            # 16,384 candidates: m + 1 too large for the top halves - whole runs
    (1, 30000, 100, 9000, 64),
    (2, 9000, 2100, 2100, 0),             # 8,192 slots: runs of 1,024 for a team of 8
    (1, 17000, 4097, 64, 0),              # many short chunks on a memory just beyond the LDS-resident loop's
    (1, 26000, 5000, 5000, "blocks"),
    (2, 26000, 2500, 7500, "blocks"),
    (1, 23000, 5000, 5000, "sorted"),
])
def test_team_of_workgroups_equals_one_workgroup(B, N, M, I, kind):
    L = hip.lib()
    lg = torch.from_numpy(_logits(B, N, kind, N + M)).cuda()
    n_iter = -(-(N - M) // I)
    try:
        L.ipsx_dbg_scan_team(0)
        assert hip.scan_workgroups_per_image(B, M, I, 8, 1) == 1
        want = _run(lg, M, I)
        for W in (2, 4, 8):
            L.ipsx_dbg_scan_team(W)
            used = hip.scan_workgroups_per_image(B, M, I, 8, 1)
            assert used == W or (B * W > 16 and used < W)                    # (16 units of a call at most)
            for trunc in (1, 0):
                L.ipsx_dbg_scan_team_trunc(trunc)
                got = _run(lg, M, I)
                assert np.array_equal(got[0], want[0]), (W, trunc, int((got[0] != want[0]).sum()))
                assert np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2]), (W, trunc)
                cut = _run(lg, M, I, cut=max(1, n_iter // 2))
                assert np.array_equal(cut[0], want[0]) and np.array_equal(cut[2], want[2]), (W, trunc, "resumed")
    finally:
        L.ipsx_dbg_scan_team(-1)
        L.ipsx_dbg_scan_team_trunc(1)


def test_team_on_random_shapes():
    """tools/scan_team_check.py fuzz, a short stretch of it: random M, I with 4,096 < M + I <= 16,384, ragged ends, one to
    six iterations, every kind of logits above - widths 2 / 4 / 8, top halves and whole runs, resumed ranges."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "scan_team_check.py"), "fuzz", "40000", "150"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "150 failures" not in out.stdout and ": 0 failures" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_team_shapes_and_the_workspace_they_need():
    L = hip.lib()
    assert hip.scan_workgroups_per_image(1, 5000, 5000, 8, 1) == 8          # the shipped CAMELYON sizes
    assert hip.scan_workgroups_per_image(2, 5000, 5000, 8, 1) == 8          # 16 units of a call at most: only the last slide's
    assert hip.scan_workgroups_per_image(4, 5000, 5000, 8, 1) == 4          #   loop is on the critical path
    assert hip.scan_workgroups_per_image(8, 5000, 5000, 8, 1) == 2
    assert hip.scan_workgroups_per_image(9, 5000, 5000, 8, 1) == 1
    assert hip.scan_workgroups_per_image(1, 256, 256, 8, 1) == 1            # LDS-resident loops: no team
    assert hip.scan_workgroups_per_image(1, 3000, 6000, 8, 4) == 1          # other head / token counts: scan_large_kernel
    assert hip.scan_workgroups_per_image(1, 2000, 2000, 8, 1) == 1          # <= 4,096 candidates
    # the team's block (counters + sorted runs) is part of what ipsx_scan_workspace_bytes asks for
    Lp, n2 = 10048, 16384
    assert L.ipsx_scan_workspace_bytes(1, 5000, 5000, 8, 1) >= 8 * Lp * 4 + 2 * Lp * 4 + 2048 + n2 * 8
    assert L.ipsx_scan_workspace_bytes(3, 5000, 5000, 8, 1) == 3 * L.ipsx_scan_workspace_bytes(1, 5000, 5000, 8, 1)
