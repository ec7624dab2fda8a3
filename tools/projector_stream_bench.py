#!/usr/bin/env python
"""Time ipsx_projector_stream (one persistent launch: moments + Linear + logits per tile, rows published as they
complete) against the launch-by-launch projector (row_stats + GEMM + logits) on one CAMELYON slide, both alone.

    python tools/projector_stream_bench.py [rows] [workgroups ...]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ips_amd import hip, synth   # noqa: E402


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def pmc_mode():
    """The stream as ips() launches it for one slide (255 workgroups, guided tile sizes), alone, five times: what the
    counter passes of tools/collect_profiles.sh profile - counter collection serialises kernels, so the stream cannot be
    measured inside ips() (beside its resident loop), but alone it runs as it is."""
    conf, _ = synth.bench_workload("cam")
    from ips_amd.architecture.ips_net import IPSNet
    dev = torch.device("cuda:0")
    net = synth.fill_weights(IPSNet(dev, conf), 7).to(dev).eval()
    plan = hip.EncoderPlan(net.encoder, False)
    ca = net.transf.crs_attn
    vq, R = ca.folded_query(), ca.H * ca.n_token
    n = conf.N
    x = synth.make_patches(conf, 1, seed=21)[0].to(dev)
    emb = torch.empty((n, conf.D), device=dev)
    lg = torch.empty((n, R), device=dev)
    ctl = torch.zeros((plan.stream_ctl_words(n),), dtype=torch.int32, device=dev)
    ready = torch.zeros((1,), dtype=torch.int32, device=dev)
    for _ in range(5):
        ctl.zero_()
        ready.zero_()
        plan.stream(x, vq, R, emb, lg, ctl, ready, workgroups=hip.device_geometry(dev).cus - 1, short_first=-11)
    torch.cuda.synchronize()
    print("5 launches of the projector stream on %d rows" % n)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "pmc":
        return pmc_mode()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    wgs = [int(a) for a in sys.argv[2:]] or [224, 248, 255]
    conf, _ = synth.bench_workload("cam")
    from ips_amd.architecture.ips_net import IPSNet
    dev = torch.device("cuda:0")
    net = synth.fill_weights(IPSNet(dev, conf), 7).to(dev).eval()
    plan = hip.EncoderPlan(net.encoder, False)
    ca = net.transf.crs_attn
    vq, R = ca.folded_query(), ca.H * ca.n_token
    x = torch.randn((n, conf.n_chan_in), device=dev)
    emb = torch.empty((n, conf.D), device=dev)
    lg = torch.empty((1, n, R), device=dev)
    flop = 2.0 * n * conf.n_chan_in * conf.D

    def layered():
        plan.encode(x, out=emb)
        hip.logits(emb.view(1, n, -1), None, vq, R, out=lg)

    ms = timed(layered)
    print("launch by launch: %.3f ms (%.3f of the fp32 MFMA peak)" % (ms, flop / (ms * 1e-3) / 157.3e12))
    ctl = torch.zeros((plan.stream_ctl_words(n),), dtype=torch.int32, device=dev)
    ready = torch.zeros((1,), dtype=torch.int32, device=dev)
    for w in wgs:
        for short in (0, w // 2, -2, -20, -11):
            def stream():
                ctl.zero_()
                ready.zero_()
                plan.stream(x, vq, R, emb, lg[0], ctl, ready, workgroups=w, short_first=short)
            ms = timed(stream)
            print("stream, %d workgroups, %d short first tiles (-2: every tile 32 rows, -20 | -11: guided): %.3f ms (%.3f of peak on %d units: %.3f)"
                  % (w, short, ms, flop / (ms * 1e-3) / 157.3e12, w, flop / (ms * 1e-3) / 157.3e12 * 256 / w), flush=True)
    stamps(plan, x, vq, R, emb, lg, ctl, ready, n)


def stamps(plan, x, vq, R, emb, lg, ctl, ready, n):
    """Shader cycles per phase of workgroup 0's tiles (ipsx_dbg_projector_stream_stamps)."""
    import ctypes as C
    fn = hip.lib().ipsx_dbg_projector_stream_stamps
    fn.restype, fn.argtypes = None, [C.c_void_p]
    st = torch.zeros((8,), dtype=torch.int64, device=x.device)
    fn(st.data_ptr())
    ctl.zero_(); ready.zero_()
    plan.stream(x, vq, R, emb, lg[0], ctl, ready, workgroups=256, short_first=0)
    torch.cuda.synchronize()
    fn(None)
    v = st.tolist()                                # (v[0] - pull + publication - holds the absolute time of the first stamp)
    names = ("row moments", "GEMM", "epilogue (BatchNorm, ReLU, stores)", "logits")
    tot = sum(v[1:5])
    print("workgroup 0 (STAMP build), %d rows on 256 workgroups: " % n
          + ", ".join("%s %.1f k cycles (%.0f %%)" % (nm, c / 1e3, 100.0 * c / tot) for nm, c in zip(names, v[1:5])))


if __name__ == "__main__":
    main()
