#!/usr/bin/env python
"""bench.py with the fused trunk's kernel choice pinned (ipsx_dbg_fused_trunk_pair): 0 rule, 1 one wavefront per patch,
2 two wavefronts per patch.   python tools/bench_mode.py <mode> [bench.py arguments]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ips_amd import hip   # noqa: E402

mode = int(sys.argv[1])
fn = hip.lib().ipsx_dbg_fused_trunk_pair
fn.restype, fn.argtypes = None, [C.c_int]
fn(mode)
sys.argv = ["bench.py"] + sys.argv[2:]
import bench   # noqa: E402

bench.main()
