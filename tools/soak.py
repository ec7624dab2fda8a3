#!/usr/bin/env python
"""Soak test of the selection pipelines: the legs of the default bench.py run, in sequence, round after round, for a given
number of seconds on one GPU - every leg builds its net afresh each round (as bench.py does), runs a burst of ips() calls
back to back and a burst of device-synchronised calls, and must select the same patches every time.

    python tools/soak.py [seconds, default 150] [calls per burst, default 20]

What it watches (co-resident spinning kernels + gate kernels owe a soak record):
  * results: `last_mem_idx` of every call against the leg's first call (and, where a reference fixture exists, against it);
  * the persistent pipelines: timeouts seen by the host (`hip._PERSIST_STRIKES`), whether they were switched off;
  * time: the slowest synchronised call of every leg against its median (a wait that hits its 50 ms bound shows here);
  * memory: allocator statistics at the end of every round (must stay flat from round 2 on);
  * a watchdog: any call that takes longer than 60 s dumps every thread's stack and ends the run (non-zero exit code).
Every device-side wait is bounded - the loop's wait for rows and the gate's wait for the loop by
`ipsx_set_persistent_wait_ms` (50 ms without progress), the column quarters' hand-over by 50 ms - and a loop that gives up
is redone in the same call (`ipsx_scan_range_if`); the host never waits on a flag, only on streams.
"""
import faulthandler
import gc
import os
import statistics
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from ips_amd import hip, synth                      # noqa: E402
from ips_amd.architecture import IPSNet             # noqa: E402

LEGS = (("mnist", "mnist", None), ("b1", "mnist", 1), ("mnist3000", "mnist3000", None), ("cam", "cam", None),
        ("cam_x16", "cam", 16), ("cam_native", "cam_native", None), ("traffic", "traffic", None), ("native50", "native50", None))


def fixture(name):
    path = os.path.join(REPO, "tests", "golden", "bench_%s.npz" % name)
    return np.load(path)["trace_idx"][:, -1].astype(np.int64) if os.path.exists(path) else None


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 150.0
    burst = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    only = sys.argv[3].split(",") if len(sys.argv) > 3 else None       # a comma list of legs (default: all)
    legs = [l for l in LEGS if only is None or l[0] in only]
    dev = torch.device("cuda:0")
    events = []                                                       # (round, leg, call) of every loop timeout the host saw

    import ctypes as C
    plog = hip.lib().ipsx_dbg_persist_log
    plog.restype, plog.argtypes = C.c_int, [C.c_void_p]

    def gate_log(tag):
        """what the gate kernels and the resident loops saw since the last look (100 MHz ticks -> ms)"""
        buf = (C.c_ulonglong * 8)()
        if plog(buf) != 0:
            return
        v = [int(t) for t in buf]
        late = (v[7] - v[6]) / 1e5 if v[7] else float("nan")
        print("   %s: %d gate launches, longest gate wait %.3f ms (its loop became resident %.3f ms after the gate started), gates "
              "that gave up %d, loops that gave up %d" % (tag, v[0], v[1] / 1e5, late, v[2], v[5]), flush=True)

    hgap = hip.lib().ipsx_dbg_call_host_gap
    hgap.restype, hgap.argtypes = None, [C.c_void_p]
    host = {"max_ms": 0.0, "long": 0, "calls": 0}

    def host_gap(tag=None):
        """the library call's own clock: host time from the loop's launch to its producer's launch having returned"""
        buf = (C.c_ulonglong * 3)()
        hgap(buf)
        host["max_ms"] = max(host["max_ms"], buf[0] / 1e6)
        host["long"] += int(buf[1])
        host["calls"] += int(buf[2])
        if tag:
            print("   %s: host window loop launch -> producer launched, longest %.3f ms over %d library calls (%d longer than 10 ms)" % (
                tag, buf[0] / 1e6, buf[2], buf[1]), flush=True)

    def throttled():
        """the container's CPU quota: (periods throttled, ms throttled) of /sys/fs/cgroup/cpu.stat, None if not readable"""
        for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
            try:
                kv = dict(l.split() for l in open(path).read().splitlines())
            except OSError:
                continue
            t = kv.get("throttled_usec")
            return int(kv.get("nr_throttled", 0)), (int(t) / 1e3 if t is not None else int(kv.get("throttled_time", 0)) / 1e6)
        return None
    thr0 = throttled()

    # torch sizes its CPU pools by the machine (256 hardware threads on the GPU box), the container may run 16 of them at a
    # time (cpu.max): building a net on 128 threads then stalls the whole process for the rest of a 100 ms quota period -
    # anywhere, also between two launches of one ips() call.  IPSX_SOAK_THREADS=0 leaves torch's default (the stalls on record
    # in profiles/r05_soak_long.txt)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else max(1, int(q) // int(per))
    except (OSError, ValueError):
        pass
    want_threads = int(os.environ.get("IPSX_SOAK_THREADS", str(min(quota or 8, os.cpu_count() or 8) // 2 or 1)))
    if want_threads > 0:
        torch.set_num_threads(want_threads)
    print("host: %s CPUs visible, container quota %s, torch intra-op threads %d" % (os.cpu_count(), quota, torch.get_num_threads()), flush=True)

    def note(rounds, leg, call, before):
        if hip._PERSIST_STRIKES != before:
            events.append((rounds, leg, call))
            print("   loop timeout seen by the host: round %d, leg %s, %s" % (rounds, leg, call), flush=True)
            host_gap("at that moment")
        return hip._PERSIST_STRIKES
    inputs, want, stats = {}, {}, {}
    t_start = time.perf_counter()
    rounds = 0
    ok = True
    while time.perf_counter() - t_start < seconds or rounds < 2:
        rounds += 1
        for leg, cfg, batch in legs:
            conf, B = synth.bench_workload(cfg)
            if leg not in inputs:
                x = synth.make_patches(conf, B, seed=21)
                if batch == 1:
                    x = x[:1]
                x = x.to(dev)
                if batch and batch > B:             # more slides than the fixture holds: drawn on the device (as bench.py does)
                    g = torch.Generator(device=dev)
                    g.manual_seed(2100)
                    x = torch.cat((x, torch.relu(torch.randn((batch - B,) + tuple(x.shape[1:]), generator=g, device=dev))), 0)
                inputs[leg] = x
                stats[leg] = {"calls": 0, "lat": [], "mismatch": 0, "fixture_equal": None}
            x = inputs[leg]
            st = stats[leg]
            net = synth.fill_weights(IPSNet(dev, conf), 7).to(dev).eval()
            faulthandler.dump_traceback_later(60, exit=True)
            seen = hip._PERSIST_STRIKES
            net.ips(x)
            seen = note(rounds, leg, "first call of a fresh net", seen)
            idx = net.last_mem_idx.clone()
            if leg not in want:
                want[leg] = idx
                fx = fixture(cfg)
                if fx is not None:
                    got = idx.cpu().numpy()[:fx.shape[0]] if batch != 1 else idx.cpu().numpy()
                    fx = fx[:1] if batch == 1 else fx
                    st["fixture_equal"] = bool((got == fx).all()) if cfg != "cam_native" else float((got == fx).mean())
            for k in range(burst):                  # back to back: the host runs ahead of the device
                net.ips(x)
                seen = note(rounds, leg, "back-to-back call %d" % (k + 1), seen)
                st["mismatch"] += 0 if torch.equal(net.last_mem_idx, want[leg]) else 1
            torch.cuda.synchronize()
            for k in range(burst):                  # synchronised: every call's own time
                c0 = time.perf_counter()
                net.ips(x)
                torch.cuda.synchronize()
                st["lat"].append(1e3 * (time.perf_counter() - c0))
                seen = note(rounds, leg, "synchronised call %d (%.1f ms; the timeout belongs to the call before)" % (k + 1, st["lat"][-1]), seen)
                if st["lat"][-1] > 15.0 and st["lat"][-1] > 4 * statistics.median(st["lat"]):
                    gate_log("round %d, leg %s, synchronised call %d took %.1f ms" % (rounds, leg, k + 1, st["lat"][-1]))
                st["mismatch"] += 0 if torch.equal(net.last_mem_idx, want[leg]) else 1
            faulthandler.cancel_dump_traceback_later()
            st["calls"] += 1 + 2 * burst
            del net
        gc.collect()                                # (net <-> selection <-> plan reference one another: the collector frees them)
        if rounds % 10 == 0:
            gate_log("rounds %d-%d" % (rounds - 9, rounds))
            host_gap("rounds %d-%d" % (rounds - 9, rounds))
        torch.cuda.synchronize()
        print("round %d: %.0f s, allocated %.1f MB, reserved %.1f MB, persistent timeouts so far %d%s" % (
            rounds, time.perf_counter() - t_start, torch.cuda.memory_allocated() / 2**20, torch.cuda.memory_reserved() / 2**20,
            hip._PERSIST_STRIKES, ", persistent pipelines OFF: %s" % hip._PERSIST_OFF if hip._PERSIST_OFF else ""), flush=True)
    print("%-11s %7s %9s %9s %9s %9s  %s" % ("leg", "calls", "median ms", "p99 ms", "max ms", "mismatch", "first call == reference fixture"))
    for leg, _, _ in legs:
        st = stats[leg]
        lat = sorted(st["lat"])
        print("%-11s %7d %9.3f %9.3f %9.3f %9d  %s" % (leg, st["calls"], statistics.median(lat), lat[int(0.99 * (len(lat) - 1))], lat[-1],
                                                       st["mismatch"], st["fixture_equal"]))
        ok = ok and st["mismatch"] == 0
    total = sum(st["calls"] for st in stats.values())
    host_gap()
    thr1 = throttled()
    print("host: longest window loop launch -> producer launched %.3f ms over %d library calls, %d longer than 10 ms; the container's "
          "CPU quota throttled it %s" % (host["max_ms"], host["calls"], host["long"],
                                         "in %d periods, %.0f ms in all" % (thr1[0] - thr0[0], thr1[1] - thr0[1]) if thr0 and thr1 else "(cpu.stat not readable)"))
    print("%d ips() calls in %.0f s over %d rounds; persistent timeouts %d %s; pipelines switched off: %s; %s" % (
        total, time.perf_counter() - t_start, rounds, hip._PERSIST_STRIKES, events, bool(hip._PERSIST_OFF), "OK" if ok else "MISMATCH"))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
