#!/usr/bin/env python
"""bench.py with every hip.scan_range call bracketed by events: prints the last calls' durations (diagnostic)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ips_amd import hip   # noqa: E402

real = hip.scan_range
marks = []


def timed(*a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = real(*a, **k)
    e1.record()
    marks.append((a[5], a[6], e0, e1))
    return r


hip.scan_range = timed
sys.argv = ["bench.py"] + sys.argv[1:]
import bench   # noqa: E402

try:
    bench.main()
except ZeroDivisionError:
    pass
torch.cuda.synchronize()
for a, b, e0, e1 in marks[:9] + marks[-9:]:
    print("iterations %d-%d %.1f us" % (a, b, e0.elapsed_time(e1) * 1e3), file=sys.stderr)
