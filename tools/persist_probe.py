#!/usr/bin/env python
"""Does a persistent scan launched first run BESIDE the main stream's kernels?  Launches ipsx_scan_persistent on a side
stream onto an idle GPU, then (after a host sleep) publishes the rows from the main stream, with and without encoder-like
work queued in front of the publish; reports the status word and the wall time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ips_amd import hip

dev = torch.device("cuda:0")
B, N, M, I, H, T = 1, 65536, 256, 256, 8, 1
g = torch.Generator(device="cpu").manual_seed(0)
lg = (torch.randn((B, N, H * T), generator=g) * 3).to(dev)
want = hip.scan(lg, M, I, H, T)
torch.cuda.synchronize()
for mode in sys.argv[1:] or ["default", "side"]:
    for work in (False, True):
        main = torch.cuda.current_stream(dev) if mode == "default" else torch.cuda.Stream(device=dev)
        side = torch.cuda.Stream(device=dev, priority=-1)
        with torch.cuda.stream(main):
            ready = torch.zeros((1,), dtype=torch.int32, device=dev)
            status = torch.zeros((1,), dtype=torch.int32, device=dev)
            mem = torch.empty((B, M), dtype=torch.int64, device=dev)
            tie = torch.zeros((B,), dtype=torch.int32, device=dev)
            big = torch.randn((8192, 8192), device=dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                hip.scan_persistent(lg, M, I, H, T, mem, tie, ready, status)
            time.sleep(0.05)
            if work:
                for _ in range(4):
                    big = big @ big * 1e-4            # ~25 ms of GEMMs on every CU
            hip.publish_rows(ready, N)
            main.wait_stream(side)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print("main=%s work=%s: %.1f ms, status %d, indices %s" % (mode, work, 1e3 * dt, int(status.item()),
                                                                   "ok" if torch.equal(mem, want) else "WRONG"), flush=True)
