#!/usr/bin/env python
"""What slows the CAMELYON selection loop (persistent kernel, every row published, its own compute unit) when the rest of
the GPU is busy?  Times the loop alone and beside: the projector stream on other buffers, the fused MNIST trunk (matrix
cores + L2, next to no HBM traffic), a device-to-device copy (HBM only).   python tools/loop_beside.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ips_amd import hip, synth   # noqa: E402


def main():
    dev = torch.device("cuda:0")
    from ips_amd.architecture.ips_net import IPSNet
    conf, _ = synth.bench_workload("cam")
    net = synth.fill_weights(IPSNet(dev, conf), 7).to(dev).eval()
    x = synth.make_patches(conf, 1, seed=21).to(dev)
    net.ips(x)
    lg = net.selection._bufs["features"][1][0].clone()                        # the slide's real logits
    N, M, I, H, T = conf.N, conf.M, conf.I, conf.H, conf.n_token
    mem = torch.empty((1, M), dtype=torch.int64, device=dev)
    tie = torch.zeros((1,), dtype=torch.int32, device=dev)
    words = torch.zeros((2,), dtype=torch.int32, device=dev)
    side = torch.cuda.Stream(device=dev, priority=-1)
    plan = net._plan
    ca = net.transf.crs_attn
    vq, R = ca.folded_query(), ca.H * ca.n_token
    xs = torch.randn((3 * N, conf.n_chan_in), device=dev)
    emb = torch.empty((3 * N, conf.D), device=dev)
    lg2 = torch.empty((3 * N, R), device=dev)
    ctl = torch.zeros((plan.stream_ctl_words(3 * N),), dtype=torch.int32, device=dev)
    rdy = torch.zeros((1,), dtype=torch.int32, device=dev)
    mconf, _ = synth.bench_workload("mnist")
    mnet = synth.fill_weights(IPSNet(dev, mconf), 7).to(dev).eval()
    mplan = hip.EncoderPlan(mnet.encoder, True)
    px = torch.randn((224 * 4 * 6, 1, 32, 32), device=dev)
    mplan.encode(px[:8])
    big = torch.empty((1 << 30,), dtype=torch.uint8, device=dev)
    big2 = torch.empty_like(big)

    def stream_other():
        ctl.zero_(); rdy.zero_()
        plan.stream(xs, vq, R, emb, lg2, ctl, rdy, workgroups=224)

    def trunk_other():
        for _ in range(6):
            mplan.encode(px[:224 * 4])                       # 224 workgroups: one per unit, the loop's stays free

    def copy_other():
        for _ in range(3):
            big2.copy_(big)

    for name, other in (("alone", None), ("beside the projector stream (224 workgroups)", stream_other),
                        ("beside the fused trunk (224 workgroups)", trunk_other), ("beside device-to-device copies", copy_other)):
        ts = []
        for rep in range(4):
            words.zero_(); words[0] = N; tie.zero_()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(side):
                e0.record(side)
                hip.scan_persistent(lg, M, I, H, T, mem, tie, words[:1], words[1:2])
                e1.record(side)
            hip.scan_gate(words[1:2])
            if other is not None:
                other()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print("%-48s %.3f ms  (%.2f us per iteration)" % (name, min(ts), 1e3 * min(ts) / 255), flush=True)


if __name__ == "__main__":
    main()
