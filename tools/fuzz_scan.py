#!/usr/bin/env python
"""Extended fuzzing of the selection loop against the oracle: the body of
tests/test_hip_kernels.py::test_scan_random_shapes_with_ties_matches_oracle - and of ::test_scan_cam_shape_with_ties_matches_oracle
(the specialised loop of BASELINE configs[3]) - for many more seeds.
    python tools/fuzz_scan.py [first_seed] [n_seeds] [n_seeds of the CAMELYON-shape test]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_hip_kernels as t

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
bad = 0
for seed in range(first, first + n):
    try:
        t.test_scan_random_shapes_with_ties_matches_oracle(seed)
    except AssertionError as e:
        bad += 1
        print("seed", seed, "FAILED", str(e)[:200])
print("seeds %d..%d: %d failures" % (first, first + n - 1, bad), flush=True)
n_cam = int(sys.argv[3]) if len(sys.argv) > 3 else 0
bad_cam = 0
for seed in range(first, first + n_cam):
    try:
        t.test_scan_cam_shape_with_ties_matches_oracle(seed)
    except AssertionError as e:
        bad_cam += 1
        print("CAMELYON shape, seed", seed, "FAILED", str(e)[:200])
if n_cam:
    print("CAMELYON shape, seeds %d..%d: %d failures" % (first, first + n_cam - 1, bad_cam))
bad += bad_cam
sys.exit(1 if bad else 0)
