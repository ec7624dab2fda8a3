#!/usr/bin/env python
"""Every launch of ipsx_projector_stream timed on its own (HIP events), many launches per configuration: is a slow average
(tools/projector_stream_bench.py showed 5 - 7 ms for single configurations at 255 workgroups) EVERY launch or one launch?
    python tools/stream_outliers.py [launches per configuration]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ips_amd import hip, synth   # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    n = 65536
    conf, _ = synth.bench_workload("cam")
    from ips_amd.architecture.ips_net import IPSNet
    dev = torch.device("cuda:0")
    net = synth.fill_weights(IPSNet(dev, conf), 7).to(dev).eval()
    plan = hip.EncoderPlan(net.encoder, False)
    ca = net.transf.crs_attn
    vq, R = ca.folded_query(), ca.H * ca.n_token
    x = torch.randn((n, conf.n_chan_in), device=dev)
    emb = torch.empty((n, conf.D), device=dev)
    lg = torch.empty((n, R), device=dev)
    ctl = torch.zeros((plan.stream_ctl_words(n),), dtype=torch.int32, device=dev)
    ready = torch.zeros((1,), dtype=torch.int32, device=dev)
    for w in (248, 255, 256):
        for short in (0, w // 2, -11):
            ts = []
            for _ in range(reps):
                ctl.zero_()
                ready.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                plan.stream(x, vq, R, emb, lg, ctl, ready, workgroups=w, short_first=short)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            srt = sorted(ts)
            slow = [(i, round(t, 2)) for i, t in enumerate(ts) if t > 1.5 * srt[len(srt) // 2]]
            print("%3d workgroups, short_first %4d: min %.3f  median %.3f  max %.3f ms; launches over 1.5 x median: %s"
                  % (w, short, srt[0], srt[len(srt) // 2], srt[-1], slow or "none"), flush=True)


if __name__ == "__main__":
    main()
