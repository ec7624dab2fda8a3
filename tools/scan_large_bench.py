#!/usr/bin/env python
"""Time the selection loop for candidate sets beyond the LDS (scan_large_kernel) on random logits: the reference's
shipped CAMELYON sizes (M = I = 5000, 8 rows per patch) by default, with torch.topk's tie order replayed (the default;
at 10,000 candidates some pair of scores is bit-equal in practically every iteration) and with the canonical order.

    python tools/scan_large_bench.py [N M I H T]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

from ips_amd import hip


def main():
    N, M, I, H, T = (int(v) for v in sys.argv[1:6]) if len(sys.argv) >= 6 else (38000, 5000, 5000, 8, 1)
    B = int(os.environ.get("B", "1"))
    g = np.random.default_rng(5)
    lg = torch.from_numpy((g.standard_normal((B, N, H * T)) * 1.5).astype(np.float32)).cuda()
    n_iter = -(-(N - M) // I)
    for mode in ("torch", "canonical"):
        hip.set_tie_order(mode)
        for _ in range(3):
            hip.scan(lg, M, I, H, T)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            hip.scan(lg, M, I, H, T)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print("%-9s B=%d N=%d M=%d I=%d R=%d: %.3f ms per scan, %.1f us per iteration (%d iterations), boundary tie flag %s"
              % (mode, B, N, M, I, H * T, 1e3 * dt, 1e6 * dt / n_iter, n_iter, hip.scan.last_tie.tolist()))
    hip.set_tie_order("torch")


if __name__ == "__main__":
    main()
