#!/usr/bin/env python
"""Extended fuzzing of ips() + the eval forward against the oracle, bit for bit: the body of
tests/test_hip_e2e.py::test_random_configurations_bit_exact_vs_oracle for many more seeds (random small configurations:
ragged chunks, the M >= N shortcut, 1-3 images, 1 / 2 / 4 query tokens, positional encoding on / off, image and feature
inputs, blank patches with ties).
    python tools/fuzz_e2e.py [first_seed] [n_seeds]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tests.test_hip_e2e as t   # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad, t0 = 0, time.time()
for seed in range(first, first + n):
    try:
        t.test_random_configurations_bit_exact_vs_oracle(seed)
    except AssertionError as e:
        bad += 1
        print("seed", seed, "FAILED", str(e)[:200], flush=True)
print("seeds %d..%d: %d failures (%.2f s per seed)" % (first, first + n - 1, bad, (time.time() - t0) / max(n, 1)))
sys.exit(1 if bad else 0)
