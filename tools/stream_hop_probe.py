#!/usr/bin/env python
"""Cost of a cross-stream hand-over (wait_stream + a tiny kernel) between the kinds of streams ips() uses: torch's default
stream, a torch side stream, and streams with a compute-unit mask (hip.masked_streams).  Ping-pong of 200 hops, host-timed.
    python tools/stream_hop_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
from ips_amd import hip


def masked_stream(bits):
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            rt = C.CDLL(line.split()[-1])
            break
    words = (C.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    s = C.c_void_p()
    rt.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
    assert rt.hipExtStreamCreateWithCUMask(C.byref(s), 8, words) == 0
    return torch.cuda.ExternalStream(s.value)


dev = torch.device("cuda:0")
x = torch.zeros(64, device=dev)
trunk, loops = masked_stream(((1 << 256) - 1) & ~0xFFFF), masked_stream(0xFFFF)
default = torch.cuda.current_stream(dev)
side = hip.side_stream(dev)
plain = torch.cuda.Stream(device=dev)
plain2 = torch.cuda.Stream(device=dev)


def pingpong(a, b, n=100):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        with torch.cuda.stream(a):
            x.add_(1.0)
        b.wait_stream(a)
        with torch.cuda.stream(b):
            x.add_(1.0)
        a.wait_stream(b)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (2 * n) * 1e6


for name, a, b in (("default <-> side stream", default, side), ("default <-> masked (240 units)", default, trunk),
                   ("default <-> masked (16 units)", default, loops), ("masked <-> masked", trunk, loops),
                   ("plain <-> plain", plain, plain2), ("plain <-> masked (240 units)", plain, trunk),
                   ("same stream (no hop)", default, default)):
    pingpong(a, b, 20)
    print("%-34s %6.1f us per hop" % (name, pingpong(a, b)))
