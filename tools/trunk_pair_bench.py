#!/usr/bin/env python
"""Time the fused 1x32x32 trunk on n patches with either exact fp32 kernel: fused_trunk_kernel (one wavefront per
patch) and fused_trunk_pair_kernel (two; fused_trunk_pair.h), and the product's rule (whole rounds + remainder).

    python tools/trunk_pair_bench.py [n ...]
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ips_amd import hip, synth   # noqa: E402


def main():
    ns = [int(a) for a in sys.argv[1:]] or [128, 256, 452, 512, 600, 1024, 2048, 2500]
    conf, _ = synth.bench_workload("mnist")
    from ips_amd.architecture.ips_net import IPSNet
    dev = torch.device("cuda:0")
    net = synth.fill_weights(IPSNet(dev, conf), 7).to(dev).eval()
    plan = hip.EncoderPlan(net.encoder, True)
    mode = hip.lib().ipsx_dbg_fused_trunk_pair
    mode.restype, mode.argtypes = None, [C.c_int]
    flop = 2 * 18628608
    for n in ns:
        x = torch.randn((n, 1, 32, 32), device="cuda:0")
        line = "n %5d" % n
        for m, name in ((1, "one wave/patch"), (2, "two waves/patch"), (0, "rule")):
            mode(m)
            for _ in range(3):
                plan.encode(x)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                plan.encode(x)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            line += "   %s %.3f ms (%.2f of peak)" % (name, ms, n * flop / (ms * 1e-3) / 157.3e12)
        mode(0)
        print(line, flush=True)


if __name__ == "__main__":
    main()
