#!/usr/bin/env python
"""scan_large_team_kernel (a team of workgroups per image, csrc/scan_large_team.h) against scan_large_kernel (one) on
the same logits: indices, scores, tie flags, resumed ranges - and the time per iteration of each.

    python tools/scan_team_check.py [reps]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from ips_amd import hip

SHAPES = [  # B, N, M, I, levels (0: gaussian logits; n > 0: quantised to n levels - ties everywhere; n < 0: -n duplicated rows)
    (1, 38000, 5000, 5000, 0), (2, 21000, 5000, 5000, 4096), (1, 24000, 5000, 5000, -6), (3, 40000, 8192, 8192, 0),
    (1, 30000, 100, 9000, 64), (2, 20000, 6000, 4000, -1), (1, 17000, 4097, 64, 0), (2, 9000, 2100, 2100, 0), (1, 12000, 3000, 1200, -20),
    (1, 26000, 5000, 5000, "blocks"),   # every 8th 64-block of patches scores high: ONE workgroup of a team of 8 holds the whole top
    (2, 26000, 2500, 7500, "blocks"),   #   in iteration 0 - the ranking from the runs' top halves must see that and merge everything
]


def logits(B, N, levels, seed):
    g = np.random.default_rng(seed)
    if levels == "blocks":
        lg = (g.standard_normal((B, N, 8)) * 1.0).astype(np.float32)
        lg[:, ((np.arange(N) >> 6) & 7) == 0] += np.float32(6.0)
        return lg
    if levels > 0:
        lg = (g.integers(0, levels, (B, N, 8)).astype(np.float32) - np.float32(levels / 2)) * np.float32(6.0 / levels)
        if levels <= 64:
            lg[:, :, 1:] = lg[:, :, :1]
        return lg
    lg = (g.standard_normal((B, N, 8)) * 3.0).astype(np.float32)
    for b in range(B):
        src, dst = g.integers(0, N, -levels), g.integers(0, N, -levels)
        lg[b, dst] = lg[b, src]
    return lg


def run(lg, M, I, cut=None):
    B, N = lg.shape[:2]
    n_iter = -(-(N - M) // I)
    if cut is None:
        mem, sc = hip.scan(lg, M, I, 8, 1, want_scores=True)
        return mem.cpu().numpy(), sc.cpu().numpy(), hip.scan.last_tie.cpu().numpy()
    idx = torch.empty((B, M), dtype=torch.int64, device=lg.device)
    tie = torch.zeros((B,), dtype=torch.int32, device=lg.device)
    hip.scan_range(lg, M, I, 8, 1, 0, cut, idx, tie)
    hip.scan_range(lg, M, I, 8, 1, cut, n_iter, idx, tie)
    return idx.cpu().numpy(), None, tie.cpu().numpy()


def fuzz(first, n):
    """Random shapes beyond the LDS (8 heads, one token), random kinds of logits, every team width and the ranking from whole
    runs too, against the one-workgroup kernel:  python tools/scan_team_check.py fuzz [first seed] [seeds]"""
    L = hip.lib()
    bad = 0
    t0 = time.time()
    for seed in range(first, first + n):
        g = np.random.default_rng(9000 + seed)
        while True:
            M = int(g.integers(1, 9000))
            I = int(g.integers(1, 9000))
            if 4096 < M + I <= 16384:
                break
        B = int(g.choice([1, 1, 2, 3]))
        n_iter = int(g.integers(1, 7))
        N = M + (n_iter - 1) * I + int(g.integers(1, I + 1))                 # a ragged last chunk more often than not
        kind = [0, 0, 4096, 256, 64, -1, -6, -40, "blocks"][int(g.integers(0, 9))]
        lg = torch.from_numpy(logits(B, N, kind, seed)).cuda()
        cut = max(1, n_iter // 2) if n_iter > 1 else None
        L.ipsx_dbg_scan_team(0)
        want = run(lg, M, I)
        line = []
        for W in (2, 4, 8):
            for trunc in (1, 0):
                L.ipsx_dbg_scan_team(W)
                L.ipsx_dbg_scan_team_trunc(trunc)
                got = run(lg, M, I)
                ok = np.array_equal(got[0], want[0]) and np.array_equal(got[1].view(np.int32), want[1].view(np.int32)) and np.array_equal(got[2], want[2])
                if ok and cut is not None:
                    got2 = run(lg, M, I, cut=cut)
                    ok = np.array_equal(got2[0], want[0]) and np.array_equal(got2[2], want[2])
                if not ok:
                    line.append("W=%d trunc=%d" % (W, trunc))
        if seed - first < 5:
            L.ipsx_dbg_scan_team(-1)
            print("seed %d: B %d N %d M %d I %d (%d iterations) kind %s, default team %d" % (
                seed, B, N, M, I, n_iter, kind, hip.scan_workgroups_per_image(B, M, I, 8, 1)), flush=True)
        if line:
            bad += 1
            print("seed %d (B %d N %d M %d I %d kind %s): DIFFERENT at %s" % (seed, B, N, M, I, kind, ", ".join(line)), flush=True)
    L.ipsx_dbg_scan_team(-1)
    L.ipsx_dbg_scan_team_trunc(1)
    print("team fuzz, seeds %d..%d: %d failures (%.2f s per seed)" % (first, first + n - 1, bad, (time.time() - t0) / max(n, 1)))
    sys.exit(1 if bad else 0)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "fuzz":
        return fuzz(int(sys.argv[2]) if len(sys.argv) > 2 else 0, int(sys.argv[3]) if len(sys.argv) > 3 else 100)
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    L = hip.lib()
    bad = 0
    for B, N, M, I, levels in SHAPES:
        lg = torch.from_numpy(logits(B, N, levels, N + M)).cuda()
        n_iter = -(-(N - M) // I)
        L.ipsx_dbg_scan_team(0)
        want = run(lg, M, I)
        line = "B=%d N=%d M=%d I=%d levels=%s:" % (B, N, M, I, levels)
        for W in (0, 2, 4, 8):
            L.ipsx_dbg_scan_team(W)
            used = hip.scan_workgroups_per_image(B, M, I, 8, 1)
            got = run(lg, M, I)
            got2 = run(lg, M, I, cut=max(1, n_iter // 2))
            ok = (np.array_equal(got[0], want[0]) and np.array_equal(got[1].view(np.int32), want[1].view(np.int32))
                  and np.array_equal(got[2], want[2]) and np.array_equal(got2[0], want[0]) and np.array_equal(got2[2], want[2]))
            bad += 0 if ok else 1
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                hip.scan(lg, M, I, 8, 1)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            line += "  W=%d(%d) %s %.1f us/it" % (W, used, "ok" if ok else "DIFFERENT", 1e6 * dt / n_iter)
        print(line, flush=True)
    L.ipsx_dbg_scan_team(-1)
    print("team check: %d mismatches" % bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
