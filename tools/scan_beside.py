#!/usr/bin/env python
"""How long do iterations [it0, it1) of the MNIST-shaped selection loop (M = I = 64, 4 tokens x 8 heads, one image) take
ALONE and BESIDE a launch of the fused trunk on another stream?   python tools/scan_beside.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ips_amd import hip, synth   # noqa: E402


def main():
    conf, _ = synth.bench_workload("mnist")
    from ips_amd.architecture.ips_net import IPSNet
    dev = torch.device("cuda:0")
    net = synth.fill_weights(IPSNet(dev, conf), 7).to(dev).eval()
    plan = hip.EncoderPlan(net.encoder, True)
    mode = hip.lib().ipsx_dbg_fused_trunk_pair
    mode.restype, mode.argtypes = None, [C.c_int]
    M = I = 64
    lg = torch.randn((1, 2500, 32), device=dev)
    mem = torch.empty((1, M), dtype=torch.int64, device=dev)
    tie = torch.zeros((1,), dtype=torch.int32, device=dev)
    side = torch.cuda.Stream(device=dev, priority=-1)
    main_s = torch.cuda.current_stream(dev)
    plan.encode(torch.randn((8, 1, 32, 32), device=dev))

    def run(n, m, it0=0, it1=15, delay=False):
        x = torch.randn((max(n, 1), 1, 32, 32), device=dev)
        mode(m)
        ts = []
        for rep in range(6):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if it0:
                hip.scan_range(lg, M, I, 8, 4, 0, it0, mem, tie)
            torch.cuda.synchronize()
            go = torch.cuda.Event()
            go.record(main_s)
            if n:
                plan.encode(x)
            with torch.cuda.stream(side):
                side.wait_event(go)
                if delay:
                    torch.cuda._sleep(100000)      # ~40-50 us: the trunk launch is resident first
                e0.record(side)
                hip.scan_range(lg, M, I, 8, 4, it0, it1, mem, tie)
                e1.record(side)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        mode(0)
        return min(ts[1:]), sorted(ts[1:])[len(ts[1:]) // 2]

    # the small-batch pipeline of IPSNet._select_hip_overlapped, step by step
    vq = net.transf.crs_attn.folded_query()
    x1, x2 = torch.randn((1024, 1, 32, 32), device=dev), torch.randn((992, 1, 32, 32), device=dev)
    flat = torch.cat((x1, x2))
    ix2 = torch.arange(1024, 2016, device=dev, dtype=torch.int32)
    for with_first, with_logits, indexed in ((1, 1, 1), (1, 1, 2), (1, 0, 2), (0, 1, 2)):
        ts = []
        for rep in range(6):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            emb1 = plan.encode(x1) if with_first else torch.randn((1024, 128), device=dev)
            go = torch.cuda.Event()
            go.record(main_s)
            if indexed < 2:
                plan.encode_indexed(flat, ix2) if indexed else plan.encode(x2)
            with torch.cuda.stream(side):
                side.wait_event(go)
                if with_logits:
                    hip.logits(emb1.view(1, 1024, -1), None, vq, 32, out=lg[:, :1024])
                e0.record(side)
                hip.scan_range(lg, M, I, 8, 4, 0, 15, mem, tie)
                e1.record(side)
            if indexed == 2:                       # the side stream's work is enqueued first, as IPSNet does
                plan.encode_indexed(flat, ix2)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        print("pipeline: first part %d, logits %d, indexed %d: iterations 0-15 %6.1f us (median %6.1f)"
              % (with_first, with_logits, indexed, min(ts[1:]), sorted(ts[1:])[2]), flush=True)

    # the real thing: IPSNet on one image, scan_range wrapped in events
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
    for kind in ("synthetic canvas", "random patches"):
        x = synth.make_patches(conf, 1, seed=21).to(dev) if kind == "synthetic canvas" else torch.randn((1, 2500, 1, 32, 32), device=dev)
        with torch.no_grad():
            for rep in range(4):
                del marks[:]
                net.ips(x)
                torch.cuda.synchronize()
        print("IPSNet.ips, %s: %s" % (kind, "  ".join("iterations %d-%d %.1f us" % (a, b, e0.elapsed_time(e1) * 1e3) for a, b, e0, e1 in marks)), flush=True)
    hip.scan_range = real

    for n, m, name in ((0, 1, "alone"), (452, 1, "452 one wave/patch (113 wgs)"), (484, 2, "484 two waves/patch (242 wgs)"),
                       (992, 1, "992 one wave/patch (248 wgs)"), (1024, 1, "1024 one wave/patch (256 wgs)"),
                       (2048, 1, "2048 one wave/patch (512 wgs)")):
        for delay in (False, True):
            a = run(n, m, 0, 15, delay)
            b = run(n, m, 15, 30, delay)
            print("%-34s %s  iterations 0-15: %6.1f us (median %6.1f)   15-30: %6.1f us (median %6.1f)"
                  % (name, "late " if delay else "early", a[0], a[1], b[0], b[1]), flush=True)


if __name__ == "__main__":
    main()
