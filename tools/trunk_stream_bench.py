#!/usr/bin/env python
"""Time ipsx_trunk_stream (one image: trunk + logits as one persistent launch, two patches per pull) against
ipsx_trunk_encode + ipsx_logits on the same patches, both alone.   python tools/trunk_stream_bench.py [patches] [workgroups ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ips_amd import hip, synth   # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
    wgs = [int(a) for a in sys.argv[2:]] or [255, 264]
    conf, _ = synth.bench_workload("mnist")
    from ips_amd.architecture.ips_net import IPSNet
    dev = torch.device("cuda:0")
    net = synth.fill_weights(IPSNet(dev, conf), 7).to(dev).eval()
    plan = hip.EncoderPlan(net.encoder, True)
    ca = net.transf.crs_attn
    vq, R = ca.folded_query(), ca.H * ca.n_token
    x = torch.randn((n, 1, 32, 32), device=dev)
    pos = torch.randn((n, 128), device=dev)
    emb = torch.empty((n, 128), device=dev)
    lg = torch.empty((1, n, R), device=dev)
    flop = n * 2.0 * 18628608

    def layered():
        e = plan.encode(x)
        hip.logits(e.view(1, n, -1), pos.view(1, n, -1), vq, R, out=lg)

    ms = timed(layered)
    print("encode + logits: %.3f ms (%.3f of the fp32 MFMA peak)" % (ms, flop / (ms * 1e-3) / 157.3e12))
    ctl = torch.zeros((plan.image_stream_ctl_words(n),), dtype=torch.int32, device=dev)
    ready = torch.zeros((1,), dtype=torch.int32, device=dev)
    for w in wgs:
        for quads in (0, 1, 2, -1):
            def stream():
                ctl.zero_()
                ready.zero_()
                plan.image_stream(x, pos, vq, R, emb, lg[0], ctl, ready, workgroups=w, quad_pulls=quads)
            ms = timed(stream)
            print("stream, %d workgroups, the first %d pulls four patches (-1: the rule): %.3f ms (%.3f of peak)"
                  % (w, quads, ms, flop / (ms * 1e-3) / 157.3e12), flush=True)


if __name__ == "__main__":
    main()
