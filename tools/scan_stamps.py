#!/usr/bin/env python
"""Diagnostic: per-phase cycles of the selection-loop kernel (STAMP build), summed over iterations.
    python tools/scan_stamps.py mnist|cam|campipe|large    # scan_fast_kernel alone / inside the CAMELYON pipeline; scan_large_kernel
Also times the un-instrumented kernel (HIP events) and checks that both kernels select the same indices.
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ips_amd import hip

kind = sys.argv[1] if len(sys.argv) > 1 else "mnist"
if kind == "campipe":
    # the persistent loop INSIDE the CAMELYON pipeline (beside the projector): same phase stamps; time spent waiting for
    # rows that are not published yet lands in "weights+scores+keys" (where the next-but-one chunk is requested)
    from ips_amd import synth
    from ips_amd.architecture import IPSNet
    dev = torch.device("cuda:0")
    conf, B = synth.bench_workload("cam")
    net = synth.fill_weights(IPSNet(dev, conf), 7).to(dev).eval()
    x = synth.make_patches(conf, B, seed=21).to(dev)
    L = hip.lib()
    L.ipsx_dbg_scan_stamps.argtypes = [C.c_void_p]
    st = torch.zeros((B * 8 + 2048 + 4 * 256,), dtype=torch.int64, device=dev)     # (+ scan_cam_kernel's per-wave log of iterations 100-103)
    for _ in range(3):
        net.ips(x)
    torch.cuda.synchronize()
    L.ipsx_dbg_scan_stamps(st.data_ptr())
    net.ips(x)
    torch.cuda.synchronize()
    L.ipsx_dbg_scan_stamps(None)
    n_iter = 255
    log = st.cpu().numpy()[B * 8:B * 8 + 4 * n_iter].reshape(n_iter, 4)
    s = st.cpu().numpy()[:8]
    if log[:, 2].any() and not log[:, 0].any():
        # scan_cam_kernel: end of every iteration on the 100 MHz clock + the rows it knew to be published by then
        t = (log[:, 2] - log[0, 2]) / 100.0
        rows = log[:, 3]
        print("cam loop inside ips(): iteration ends, us since the end of iteration 0 | rows known published | us for the last 16 iterations")
        for it in range(0, n_iter, 16):
            print("   it %3d   %8.1f us   %6d rows   %6.1f us / 16 it" % (it, t[it], rows[it], t[it] - t[max(it - 16, 0)]))
        print("   it %3d   %8.1f us   %6d rows" % (n_iter - 1, t[-1], rows[-1]))
        d = t[1:] - t[:-1]
        import numpy as np
        print("   per iteration: median %.2f us, mean %.2f us, 90th percentile %.2f us, longest %.1f us (it %d); iterations over 8 us: %d, their sum %.1f us"
              % (np.median(d), d.mean(), np.percentile(d, 90), d.max(), int(d.argmax()) + 1, int((d > 8).sum()), float(d[d > 8].sum())))
        log = None
    elif not log.any():        # (no per-iteration log from this kernel: the phase totals below are what there is)
        log = None
    print("per iteration (phase 4 incl. waits, phase 5 rank), every 8th:" if log is not None else "(no per-iteration log from this kernel)")
    if log is not None:
        for it in range(0, n_iter, 16):
            print("   it %3d   %6d  %6d" % (it, log[it, 0], log[it, 1]))
        import numpy as np
        for col, nme in ((0, "phase 4"), (1, "phase 5")):
            top = np.argsort(-log[:, col])[:8]
            print("   largest %s: " % nme + ", ".join("it %d: %d" % (i, log[i, col]) for i in sorted(top)))
        raw = st.cpu().numpy()
        print("   last tie replay: re-ranking all candidates %d cycles, replay (un-sort, wavefront routines, write-back) %d cycles" % (
            raw[B * 8 + 2044], raw[B * 8 + 2045]))
        t0 = log[0, 2]
        print("   end of phase 5 in us since iteration 0 (100 MHz clock): " + ", ".join("it %d: %.1f" % (i, (log[i, 2] - t0) / 100.0) for i in (1, 10, 30, 36, 37, 38, 46, 47, 60, 100, 139, 140, 200, 254)))
    names = ["stage chunk+barrier", "row maxima", "exp (new rows)", "row sums", "weights+scores+keys (+waits)", "rank", "gather winners"]
    print("cam pipeline: total %d cycles = %.1f per iteration" % (s[:7].sum(), s[:7].sum() / n_iter))
    for k, nme in enumerate(names):
        print("  %-30s %9.0f cycles/iter" % (nme, s[k] / n_iter))
    sys.exit(0)
if kind == "largepipe":
    # the team loop (scan_large_team_kernel, STAMP build) inside ips() of the shipped CAMELYON sizes: when each iteration's
    # rows were there and when it ended
    from ips_amd import synth
    from ips_amd.architecture import IPSNet
    dev = torch.device("cuda:0")
    conf, B = synth.bench_workload("cam_native")
    net = synth.fill_weights(IPSNet(dev, conf), 7).to(dev).eval()
    x = synth.make_patches(conf, B, seed=21).to(dev)
    L = hip.lib()
    L.ipsx_dbg_scan_stamps.argtypes = [C.c_void_p]
    st = torch.zeros((B * 8 + 128,), dtype=torch.int64, device=dev)
    for _ in range(3):
        net.ips(x)
    torch.cuda.synchronize()
    L.ipsx_dbg_scan_stamps(st.data_ptr())
    net.ips(x)
    torch.cuda.synchronize()
    L.ipsx_dbg_scan_stamps(None)
    n_iter = -(-(x.shape[1] - conf.M) // conf.I)
    raw = st.cpu().numpy()
    log = raw[B * 8:B * 8 + 2 * n_iter].reshape(n_iter, 2)
    t0 = raw[B * 8 + 127]
    print("team loop inside ips() (main workgroup, STAMP build): us since the loop kernel started")
    for it in range(n_iter):
        print("   it %d   rows there %8.1f   end %8.1f   (%.1f us)" % (it, (log[it, 0] - t0) / 100.0, (log[it, 1] - t0) / 100.0,
                                                                      (log[it, 1] - log[it, 0]) / 100.0))
    names = ["rows / hop E of the iteration before + gather + own maxima", "hop A", "exponentials + store + hop B", "row sums + hop C",
             "scores + own run sorted", "run published + hop D + the other runs loaded", "own run ranked among all, ties, new memory", "arrival at hop E"]
    for k, nme in enumerate(names):
        print("  %-56s %9.1f k cycles/iter" % (nme, raw[k] / n_iter / 1e3))
    sys.exit(0)
if kind == "large":
    # scan_large_kernel (candidate sets beyond the LDS): the reference's shipped CAMELYON sizes, or N M I H T on the command line
    N, M, I, H, T = (int(v) for v in sys.argv[2:7]) if len(sys.argv) >= 7 else (38000, 5000, 5000, 8, 1)
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    lg = (torch.randn((1, N, H * T), generator=g) * 1.5).to(dev)
    if len(sys.argv) == 3 and sys.argv[2] == "bench":
        # the logits of the bench workload itself (cam_native: the projector's embeddings of the synthetic slide)
        from ips_amd import synth
        from ips_amd.architecture.ips_net import IPSNet
        conf, _ = synth.bench_workload("cam_native")
        net = synth.fill_weights(IPSNet(dev, conf), 7).to(dev).eval()
        x = synth.make_patches(conf, 1, seed=21).to(dev)
        plan = hip.EncoderPlan(net.encoder, False)
        ca = net.transf.crs_attn
        emb = plan.encode(x[0])
        lg = hip.logits(emb.view(1, x.shape[1], -1), None, ca.folded_query(), ca.H * ca.n_token).contiguous()
        N, M, I, H, T = x.shape[1], conf.M, conf.I, ca.H, ca.n_token
    L = hip.lib()
    L.ipsx_dbg_scan_stamps.argtypes = [C.c_void_p]
    n_iter = -(-(N - M) // I)
    names = ["stage logits (transposed)", "row maxima", "exponentials", "row sums", "scores + keys", "sort",
             "tie check + replay", "new memory"]
    team = int(os.environ.get("TEAM", "-1"))              # TEAM=0: one workgroup (scan_large_kernel); 8 (the default): the team's main workgroup
    L.ipsx_dbg_scan_team(team)
    if hip.scan_workgroups_per_image(1, M, I, H, T) > 1:
        names = ["hop E of the iteration before + gather + own maxima", "hop A: maxima, all to all", "exponentials + store + hop B (all to all)",
                 "row sums of my row + hop C (all to all)", "scores + own run sorted", "run published + hop D + the other runs loaded",
                 "own run ranked among all, ties, new memory", "arrival at hop E"]
    for mode in ("torch", "canonical"):
        hip.set_tie_order(mode)
        st = torch.zeros((1, 8 + 128), dtype=torch.int64, device=dev)      # (+ the team kernel's per-iteration log)
        hip.scan(lg, M, I, H, T)
        L.ipsx_dbg_scan_stamps(st.data_ptr())
        hip.scan(lg, M, I, H, T)
        torch.cuda.synchronize()
        L.ipsx_dbg_scan_stamps(None)
        s = st.cpu().numpy()[0][:8]
        print("large, tie order %s: %d iterations, L=%d, R=%d; %.1f k shader cycles per iteration in the STAMPED build (it spills "
              "registers the product build does not: its passes over the workspace read slower than they are)"
              % (mode, n_iter, M + I, H * T, s.sum() / n_iter / 1e3))
        for k, nme in enumerate(names):
            print("  %-56s %9.1f k cycles/iter" % (nme, s[k] / n_iter / 1e3))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            hip.scan(lg, M, I, H, T)
        b.record()
        torch.cuda.synchronize()
        print("  un-instrumented: %.3f ms per launch = %.1f us per iteration" % (a.elapsed_time(b) / 10, 1e2 * a.elapsed_time(b) / n_iter))
        if mode == "torch":
            # the replay's own phases (thread 0's clock, product build; ipsx_dbg_replay_stamps reads and clears)
            rs = (C.c_ulonglong * 10)()
            L.ipsx_dbg_replay_stamps.argtypes = [C.c_void_p]
            L.ipsx_dbg_replay_stamps(rs)
            hip.scan(lg, M, I, H, T)
            torch.cuda.synchronize()
            L.ipsx_dbg_replay_stamps(rs)
            L.ipsx_dbg_replay_stamps(None)                                 # (off again)
            v = [int(x) for x in rs]
            rn = ["nth_element by the workgroup", "nth_element chain on one wavefront", "sort level 0", "sort level 1", "sort level 2",
                  "sort level 3", "sort deeper levels", "final insertion pass"]
            print("  the replay (%d replays), product build, k cycles each: " % v[8]
                  + ", ".join("%s %.1f" % (nm, c / max(1, v[8]) / 1e3) for nm, c in zip(rn, v[:8])))
    hip.set_tie_order("torch")
    sys.exit(0)
if kind == "camwaves":
    # every wave's clock at the barriers of iterations 100..103 (STAMP build of scan_cam_kernel): who arrives last where
    import numpy as np
    dev = torch.device("cuda:0")
    B, N, M, I, H, T = 1, 65536, 256, 256, 8, 1
    g = torch.Generator(device="cpu").manual_seed(0)
    lg = (torch.randn((B, N, H * T), generator=g) * 3).to(dev)
    L = hip.lib()
    L.ipsx_dbg_scan_stamps.argtypes = [C.c_void_p]
    st = torch.zeros((B * 8 + 2048 + 4 * 256,), dtype=torch.int64, device=dev)
    hip.scan(lg, M, I, H, T)
    L.ipsx_dbg_scan_stamps(st.data_ptr())
    hip.scan(lg, M, I, H, T)
    torch.cuda.synchronize()
    L.ipsx_dbg_scan_stamps(None)
    w = st.cpu().numpy()[B * 8 + 2048:].reshape(4, 16, 16)
    pts = ["top", "B0>", ">B1", "B1>", ">B2", "B2>", ">B4", "B4>", ">B5", "B5>", ">B7", "B7>", "loads", "cnew", "end"]
    for it in (1, 2):
        t0 = w[it, :, 0].min()
        print("iteration %d: cycles since the first wave reached the top (rows: waves 0-3 memory, 4-7 chunk, 8-15 helpers)" % (100 + it))
        print("      " + " ".join("%6s" % p for p in pts))
        for wv in range(16):
            print("  w%-2d " % wv + " ".join("%6d" % (w[it, wv, k] - t0) for k in range(15)))
        nxt = w[it + 1, :, 0].min() - t0
        print("  next iteration's first wave at the top: %d cycles" % nxt)
    sys.exit(0)
B, N, M, I, H, T = (16, 2500, 64, 64, 8, 4) if kind == "mnist" else (1, 65536, 256, 256, 8, 1)
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
lg = (torch.randn((B, N, H * T), generator=g) * 3).to(dev)
L = hip.lib()
L.ipsx_dbg_scan_stamps.argtypes = [C.c_void_p]
st = torch.zeros((B, 8), dtype=torch.int64, device=dev)
hip.scan(lg, M, I, H, T)
L.ipsx_dbg_scan_stamps(st.data_ptr())
hip.scan(lg, M, I, H, T)
torch.cuda.synchronize()
L.ipsx_dbg_scan_stamps(None)
n_iter = -(-(N - M) // I)
names = ["stage chunk+barrier", "row maxima", "exp (new rows)", "row sums", "weights+scores+keys", "rank", "gather winners"]
s = st.cpu().numpy()[0]
tot = s[:7].sum()
print("%s: %d iterations, L=%d, R=%d; total %d cycles = %.1f per iteration" % (kind, n_iter, M + I, H * T, tot, tot / n_iter))
for k, nme in enumerate(names):
    print("  %-22s %9.0f cycles/iter" % (nme, s[k] / n_iter))
if len(names) == 7:
    print("  chunk candidates ranked per iteration (score >= the lowest memory score): %.1f of %d" % (s[7] / n_iter, I))
    s[7] = 0

a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    idx = hip.scan(lg, M, I, H, T)
a.record()
for _ in range(10):
    idx = hip.scan(lg, M, I, H, T)
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / 10
print("  un-instrumented: %.3f ms per launch = %.2f us per iteration" % (ms, 1e3 * ms / n_iter))
