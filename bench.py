#!/usr/bin/env python
"""bench.py - patches scored / second of the IPS no-grad hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus 8 --steps 20 --warmup 5          (starts its own 8 ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one ``IPSNet.ips(patches)`` call (no-grad, eval, eager loading: the patch
tensor is resident in HBM when the timed region starts): embed every patch, score,
run the whole selection loop, gather the M winners.

``value`` follows SURVEY.md 8 d-1: each of the K timed calls is bracketed by barrier + device
sync, a call's time is the max over ranks, ``value`` = patches per call / the MEDIAN call time
(``ms_per_step``).  ``value_pipelined`` is the same K calls enqueued back to back between two
fences (what a training loop sees when the host runs ahead).  The default single-GPU run also
times every other BASELINE configuration that fits one GPU into ``also_measured`` (b1 = d-1's
primary shape, mnist3000 = configs[2], cam / cam_x16 = configs[3], cam_native = the reference's
shipped CAMELYON sizes, traffic = configs[0]'s model on the GPU, native50 = the reference's shipped
50-px MNIST patches), each with its own ``parity`` / ``roofline`` / ``roofline_call``.

Workloads (``ips_amd.synth.BENCH_WORKLOADS``; weights seed 7, patches seed 21):
  every N BASELINE.json configs[1] ("MegaMNIST-1500 @ 1/2/4/8 GPU"): Megapixel-MNIST 1500 - 2500 patches
          of 1x32x32 per image, M = I = 64, 4 query tokens, positional encoding on, at the reference's
          batch size B = 16 (config/mnist_config.yml B_seq).  The SAME 16 x 2500 patches at every N
          (``"scaling": "strong"``): at N > 1 the patch axis is sharded over the N ranks (ips_amd/dist.py),
          value counts the patches all ranks scored, and the line carries ``n1_value_same_workload`` (rank 0
          alone, same process, same inputs) and ``also_measured.mnist3000`` (configs[2], sharded the same way).
          ``--scaling weak`` instead gives every GPU 2500 patches of each image (the image grows with N).
  ``python bench.py --gpus N`` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself
  (a child ``python -m torch.distributed.run``; the parent never touches the GPU) and relays rank 0's line.

Every line carries
  parity        ``net.last_mem_idx`` after the timed loop against the REFERENCE's selection on the same
                inputs (tests/golden/bench_<workload>.npz, recorded by running the imported reference in the
                build container, tools/gen_golden_bench.py) - on every rank;
  roofline      encoder launch(es) timed with HIP events on the launch stream inside the
                timed region; achieved = algorithmic FLOP (37,257,216 per 32-px patch,
                SURVEY.md 8 d-4) / that time, against the 157.3 TFLOP/s fp32 MFMA peak;
  cpu_baseline  oracle/ips_torch.py (the reference's ATen CPU path restated) on this host's
                cores, on a bounded sample of the same workload (N = 1 only).
"""

import argparse
import hashlib
import json
import os
import statistics
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np
import torch
import torch.distributed as dist

# algorithmic encoder FLOP per patch, SURVEY.md section 8 d-4
FLOP_PER_PATCH = {"mnist": 37_257_216, "b1": 37_257_216, "mnist3000": 37_257_216, "native50": 2 * 52_570_176,
                  "traffic": 2 * 441_262_848, "cam": 2 * 1_048_576, "cam_native": 2 * 1_048_576}
FP32_MFMA_PEAK_TFLOPS = 157.3            # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
BF16_MFMA_PEAK_TFLOPS = 2500.0
PATCHES_PER_GPU_WEAK = 2500
LABEL = {
    "mnist": "Megapixel-MNIST 1500: 2500 patches of 1x32x32 per image (BASELINE configs[1])",
    "b1": "Megapixel-MNIST 1500: 2500 patches of 1x32x32, ONE image (image 0 of the headline batch)",
    "mnist3000": "Megapixel-MNIST 3000: 10000 patches of 1x32x32 per image (BASELINE configs[2])",
    "native50": "Megapixel-MNIST reference-native: 900 patches of 1x50x50 per image, M=I=100",
    "traffic": "traffic signs: 192 patches of 3x100x100 per image, ResNet-18 x4 stages, M=16, I=32",
    "cam": "CAMELYON: 65536 x 2048 features per slide, projector, M=I=256 (BASELINE configs[3])",
    "cam_native": "CAMELYON at the reference's shipped M=I=5000 (config/camelyon_config.yml): 38000 x 2048 features per "
                  "slide, 10000 candidates per iteration",
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="images per step (default: the workload's, 16 / 1)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--dedup-blank", action="store_true",
                    help="secondary measurement: exact blank-patch deduplication in front of the encoder "
                         "(IPSX_DEDUP_BLANK=1; the encoder then runs on the ~7 %% non-blank patches only, so the "
                         "roofline object reports launch time but no FLOP rate)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "fp32x3", "bf16"],
                    help="fp32 = the headline (reference parity); bf16 = secondary measurement of BASELINE configs[4]: "
                         "bf16 operands / fp32 accumulate in the residual stages (IPSX_PRECISION=bf16)")
    ap.add_argument("--storage", default="f32", choices=["f32", "bf16", "f16"],
                    help="storage type of the patch tensor (BASELINE configs[4]: half-precision storage; needs "
                         "--precision bf16 or fp32x3 - the exact trunk reads float32)")
    ap.add_argument("--lazy", action="store_true",
                    help="secondary measurement: lazy loading - the patch tensor starts in pinned HOST memory and "
                         "is streamed over PCIe inside every step (the PCIe-inclusive rate; never the headline)")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="do not bracket the encoder launches with HIP events (no `roofline.achieved`): what the events themselves "
                         "cost - at one image per call they move the dispatcher's placement of the loop's workgroup, DESIGN 6")
    ap.add_argument("--config", default=None, choices=sorted(FLOP_PER_PATCH),
                    help="default: mnist at every --gpus N (the headline, BASELINE configs[1], B=16; patch-sharded at N > 1, "
                         "where mnist3000 = configs[2] is also allowed); the others are secondary single-GPU measurements: b1 "
                         "(headline image 0 alone), native50, traffic, cam, cam_native (the reference's shipped M = I = 5000)")
    ap.add_argument("--also", default="all",
                    help="default run only: which other BASELINE configurations are timed into `also_measured` ('all', 'none', "
                         "or a comma list of b1,mnist3000,cam,cam_x16,cam_native,traffic,native50,rank_shard_model; at N > 1 only mnist3000 - "
                         "configs[2], sharded - and the single-rank time of the headline, `n1_value_same_workload`)")
    ap.add_argument("--also-steps", type=int, default=0,
                    help="timed calls per `also_measured` leg, each way (0 = sized per leg: ~0.3 s of calls, 10 to 40)")
    ap.add_argument("--leg-timeout", type=int, default=120, help="seconds an `also_measured` leg may take before the line is "
                    "printed without it")
    ap.add_argument("--watchdog", type=int, default=600, help="seconds the headline measurement may take before every thread's "
                    "stack is dumped to stderr and the run exits")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="strong (default): BASELINE configs[1], the same 16 x 2500 patches at every N, patch axis sharded over "
                         "the ranks; weak: 2500 patches of every image per GPU (the image grows with N; N = 4 is configs[2])")
    return ap.parse_args()


def host_description():
    """CPU model, logical CPUs of the machine, CPUs this process may run on."""
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # the container's CPU quota (cgroup v2 cpu.max, "quota period" in us): threads beyond it are throttled, not run - on
    # the GPU boxes 256 logical CPUs are visible and 16 may run at a time, which is why 16 threads win the table below
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else round(int(q) / int(per), 2)
    except (OSError, ValueError):
        pass
    return {"cpu_model": model, "nproc": os.cpu_count(), "cpus_available": avail, "cpu_quota": quota}


def cpu_baseline(conf, x, budget_s):
    """The reference's CPU path (ATen restatement, oracle/ips_torch.py) on a BOUNDED sample of the same workload.

    Thread count (round 5: the probe on a 2 x 320-patch prefix did not predict the batch - 5.6 k at 64 threads on one
    box, 9.3 k at 32 on the next of the same CPU model; and ONE image alone does not either: 21 k patches/s at 16 threads
    where the batch of 16 reaches 10.6 k): FIXED counts 16 / 32 / 64 (capped by the CPUs this process may use), each timed
    on whole ``ips()`` calls over the WHOLE BATCH with the patch axis cut at a chunk boundary - every iteration then has
    the sample's own shape (B x I patches through the encoder, M + I candidates per image), and a call covers at least
    one full image's worth of patches -, the faster of two calls after a warm-up call; the table goes into the record
    and the best count runs the sample.  The sample: whole calls on the batch itself when one call fits the budget,
    otherwise on a prefix of it - fewer images first, then a shorter patch axis cut at a chunk boundary - repeated until
    the budget is used."""
    from ips_amd import synth
    from ips_amd.architecture import IPSNet
    from oracle import ips_torch

    net = synth.fill_weights(IPSNet(torch.device("cpu"), conf), 7).eval()
    sd = dict(net.state_dict())
    B, N = x.shape[:2]
    host = host_description()
    avail = host["cpus_available"]
    pos_full = net.pos_enc[:, :N] if conf.use_pos else None
    # the table's shape: all B images, M + k I patches each with B (M + k I) >= N (one image's worth) and k >= 8
    k_tab = max(8, -(-(max(N // B, conf.M) - conf.M) // conf.I))
    n_tab = min(N, conf.M + k_tab * conf.I)
    tab_conf = conf if n_tab == N else conf.clone(N=n_tab)
    pos_tab = net.pos_enc[:, :n_tab] if conf.use_pos else None
    table, best = [], None
    for thr in sorted({min(avail, c) for c in (16, 32, 64)}):
        torch.set_num_threads(thr)
        ips_torch.ips(sd, tab_conf, x[:, :n_tab], pos_tab)        # warm-up at this count (thread pool, oneDNN primitives)
        dt = None
        for _ in range(2):
            t0 = time.perf_counter()
            ips_torch.ips(sd, tab_conf, x[:, :n_tab], pos_tab)
            d1 = time.perf_counter() - t0
            dt = d1 if dt is None else min(dt, d1)
        table.append({"threads": thr, "patches_per_s": B * n_tab / dt, "ms_per_call": 1e3 * dt})
        if best is None or dt < best[1]:
            best = (thr, dt)
        if dt > max(budget_s, 1.0):                              # (a host on which this count chokes: no larger ones)
            break
    torch.set_num_threads(best[0])
    per_patch = best[1] / (B * n_tab)                           # seconds per patch at the best count
    # the sample: as much of the batch as one call can take in about a third of the budget
    room = max(1, int(budget_s / 3 / per_patch))                # patches
    b_s = max(1, min(B, room // N))
    n_s = N if b_s * N <= room else max(conf.M + conf.I, min(N, conf.M + (room - conf.M) // conf.I * conf.I))
    sample_conf = conf if n_s == N else conf.clone(N=n_s)
    xs = x[:b_s, :n_s]
    pos_s = pos_full[:, :n_s] if conf.use_pos else None
    reps, t0 = 0, time.perf_counter()
    while True:
        ips_torch.ips(sd, sample_conf, xs, pos_s)
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or reps >= 50:
            break
    whole = b_s == B and n_s == N
    return {"value": b_s * n_s * reps / dt, "unit": "patches/s", "cores": torch.get_num_threads(),
            "threads": torch.get_num_threads(), "nproc": host["nproc"], "cpus_available": avail, "cpu_quota": host["cpu_quota"],
            "cpu_model": host["cpu_model"], "kind": "port",
            "thread_table": table,
            "thread_table_on": "%d image(s) x %d patches (the whole batch, patch axis cut at a chunk boundary: the sample's own "
                               "per-iteration shape), faster of two ips() calls per count" % (B, n_tab),
            "sample": "%d x ips() on %s, %d image(s) x %d patches (oracle/ips_torch.py: the reference's ATen/oneDNN CPU path "
                      "restated, %.1f s; thread count = best of the fixed counts in thread_table)"
                      % (reps, "the bench batch itself" if whole else "a prefix of the bench batch (first images, patch axis cut at a chunk boundary)",
                         b_s, n_s, dt)}


def parity(name, mem_idx, images=None):
    """``mem_idx`` (B, M) against the reference's final selection on the same inputs."""
    path = os.path.join(REPO, "tests", "golden", "bench_%s.npz" % name)
    if not os.path.exists(path):
        return None
    z = np.load(path)
    want = z["trace_idx"][:, -1].astype(np.int64)
    gap = z["rel_gap"]
    if images is not None:
        want, gap = want[images], gap[images]
    got = mem_idx.detach().cpu().numpy()
    if got.shape != want.shape:
        return {"fixture": os.path.relpath(path, REPO), "indices_equal": False, "note": "shape %s vs %s" % (got.shape, want.shape)}
    rows = (got == want).all(1)
    same_set = float(np.mean([len(set(a) & set(b)) / len(a) for a, b in zip(got.tolist(), want.tolist())]))
    ogap = z["order_gap"][images] if images is not None else z["order_gap"]
    return {"fixture": os.path.relpath(path, REPO) + " (the reference's CPU run on these inputs, tools/gen_golden_bench.py)",
            "indices_equal": bool(rows.all()), "images": int(rows.size), "images_equal": int(rows.sum()),
            "selected_in_common": same_set,
            # same patch in the same memory slot; below 1 only where the reference's own neighbouring scores are
            # closer than its noise (min_order_gap: smallest relative step in its sorted top M + 1, last iteration)
            "slots_equal": float((got == want).mean()), "min_order_gap": float(ogap[:, -1].min()),
            "min_rel_gap": float(gap[:, -1].min()), "min_rel_gap_any_iteration": float(gap.min())}


_ALSO = {
    "fp32x3": "IPSX_PRECISION=fp32x3: fp32 operands as 3 exact bf16 terms, 6 bf16 MFMA products, f32 accumulate; "
              "max error vs float64 6.3e-7 (exact-fp32 kernel: 3.7e-7), tests/test_hip_kernels.py",
    "bf16": "IPSX_PRECISION=bf16 (BASELINE configs[4]): operands rounded to bf16, f32 accumulate; reduced precision, "
            "no reference behaviour to match",
}


def measure_precision(net, x, steps, precision, fixture, storage="f32"):
    """The same workload with another trunk arithmetic (opt-in; never the headline `value`), reported next to the
    exact-fp32 headline together with its own parity object.  ``storage``: the patch tensor's storage type - BASELINE
    configs[4] names fp16 storage (``bf16_f16``: the tensor is converted BEFORE the timed region, as a loader that stores
    halves would hand it over; a third of the bytes the trunk reads)."""
    for name in ("encode", "encode_indexed", "stream", "image_stream"):   # drop the event-recording wrappers of the headline run
        net._plan.__dict__.pop(name, None)
    net.ips(x)
    ref_idx = net.last_mem_idx.clone()
    xs = x if storage == "f32" else x.to({"f16": torch.float16, "bf16": torch.bfloat16}[storage])
    os.environ["IPSX_PRECISION"] = precision
    try:
        for _ in range(3):
            net.ips(xs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            net.ips(xs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        got = net.last_mem_idx
        same = bool(torch.equal(got, ref_idx))
        a, b = got.cpu().numpy(), ref_idx.cpu().numpy()
        common = float(np.mean([len(set(u) & set(v)) / len(u) for u, v in zip(a.tolist(), b.tolist())]))
        par = parity(fixture, got)
    finally:
        os.environ["IPSX_PRECISION"] = "fp32"
    rate = x.shape[0] * x.shape[1] * steps / dt
    peak = {"fp32x3": BF16_MFMA_PEAK_TFLOPS / 6, "bf16": BF16_MFMA_PEAK_TFLOPS}[precision]
    # the trunk ALONE on the same patches: ONE launch of the whole batch, no selection loop beside it (HIP events on the launch
    # stream) - inside a call the loops' workgroups take units from it, so the two figures differ
    os.environ["IPSX_PRECISION"] = precision
    try:
        flat = xs.reshape(-1, *xs.shape[2:])
        for _ in range(2):
            net._plan.encode(flat)
        def launches(rows):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                net._plan.encode(rows)
            e1.record()
            torch.cuda.synchronize()
            return rows.shape[0] * 10 / (e0.elapsed_time(e1) * 1e-3)
        alone = launches(flat)
        # ... and on whole rounds of the chip only (2 workgroups per CU, 8 patches each for the bf16 build, 4 for fp32x3's):
        # the batch's 40,000 patches leave 0.77 of a round, which costs 0.8 of one
        cus = torch.cuda.get_device_properties(flat.device).multi_processor_count
        whole = flat.shape[0] // (cus * 16) * (cus * 16)
        alone_whole = launches(flat[:whole]) if whole else alone
    finally:
        os.environ["IPSX_PRECISION"] = "fp32"
    return {"value": rate, "unit": "patches/s", "ms_per_step": 1e3 * dt / steps, "patch_storage": storage,
            "same_indices_as_f32": same, "selected_in_common_with_f32": common, "slots_equal_to_f32": float((a == b).mean()),
            "parity": par,
            "roofline_call": {"bound": "mfma", "achieved": rate * FLOP_PER_PATCH["mnist"] / 1e12, "peak": peak, "unit": "TFLOP/s",
                              "frac": rate * FLOP_PER_PATCH["mnist"] / 1e12 / peak,
                              "what": "whole ips() calls back to back x algorithmic encoder FLOP per patch"},
            "roofline_kernel_alone": {"bound": "mfma", "achieved": alone * FLOP_PER_PATCH["mnist"] / 1e12, "peak": peak, "unit": "TFLOP/s",
                                      "frac": alone * FLOP_PER_PATCH["mnist"] / 1e12 / peak, "patches_per_s": alone,
                                      "whole_rounds": {"patches": whole, "patches_per_s": alone_whole,
                                                       "frac": alone_whole * FLOP_PER_PATCH["mnist"] / 1e12 / peak},
                                      "what": "the trunk kernel alone: the batch's %d patches in ONE launch, 10 launches between two HIP events" % flat.shape[0]},
            "what": _ALSO[precision] + ("" if storage == "f32" else "; patches stored as %s (BASELINE configs[4])" % storage)}


def rank_shard_model(args, ctx, name="mnist", worlds=(2, 4, 8), steps=10):
    """ONE rank's share of the sharded headline on the ONE GPU of this box - a MODEL of a rank,
    not a scaling measurement.  For G in ``worlds``: rank 0's pieces of ``ips_amd.dist.shard_plan`` (its encoder launches,
    its logits, every part's loop iterations on the side stream, the winners' gather), with the two collectives replaced by
    device copies of the same sizes (``dist.LoopbackGroup``: the other ranks' logits are computed beforehand, outside the
    timed region).  What it cannot see: link latency, waiting for the slowest rank.  Next to each G the same rows cut by
    the fixed 50 / 30 / 15 / 5 % shares of rounds 1-5."""
    from ips_amd import hip, synth
    from ips_amd import dist as ipsd
    from ips_amd.architecture import IPSNet
    dev = ctx.dev
    conf, B = synth.bench_workload(name)
    net = synth.fill_weights(IPSNet(dev, conf), 7).to(dev).eval()
    x = synth.make_patches(conf, B, seed=21).to(dev)
    N = conf.N
    ca = net.transf.crs_attn
    R = ca.H * ca.n_token

    def timed_calls(fn):
        for _ in range(3):
            fn()
        lat = []
        for _ in range(steps):
            torch.cuda.synchronize()
            c0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            lat.append(1e3 * (time.perf_counter() - c0))
        return statistics.median(lat)

    n1_ms = timed_calls(lambda: net.ips(x))
    want = net.last_mem_idx.clone()
    emb = net._plan.encode(x.reshape(B * N, *x.shape[2:])).view(B, N, -1)
    full_logits = torch.empty((B, N, R), dtype=torch.float32, device=dev)
    hip.logits(emb, net.pos_enc if net.use_pos else None, ca.folded_query(), R, out=full_logits)
    del emb
    out = {"what": "MODEL, not a scaling measurement: rank 0's own launches of the sharded %s on ONE GPU (encoder pieces, logits, "
                   "loop ranges on the side stream, winners' gather), the all-gather / all-reduce replaced by device copies "
                   "of the same sizes; no link latency, no waiting for the slowest rank" % name,
           "workload": LABEL[name] + ", B=%d" % B, "n1_ms_same_process": n1_ms, "units": hip.device_geometry(dev).cus, "per_world": {}}
    for G in worlds:
        rec = {}
        for label in ("launch_aware", "fixed_shares"):
            plan = ipsd.shard_plan(net, B, N, G, tuple(x.shape[2:]))
            if label == "fixed_shares":
                fixed = ipsd.ShardPlan(N, conf.M, conf.I, G, B)
                fixed.model = plan.model
                plan = fixed
            local = x[:, plan.indices(0).to(dev)].contiguous()
            grp = ipsd.LoopbackGroup(G, 0, full_logits, x)
            timings = []
            ms = timed_calls(lambda: ipsd.ips_sharded(net, local, N, group=grp, plan=plan, timings=timings))
            torch.cuda.synchronize()
            paid, ideal = plan.rounds(0)
            rec[label] = {"ms_per_call": ms, "its": plan.its, "launch_patches": plan.launches(0),
                          "rounds_paid": paid, "rounds_ideal_one_launch": ideal, "modelled_plan_ms": plan.cost_us() / 1e3,
                          "phases": ipsd.phase_ms(timings[-steps:]),
                          "indices_equal_to_single_gpu": bool(torch.equal(net.last_mem_idx, want))}
            del local
        rec["modelled_patches_per_s_if_every_rank_took_this_long"] = B * N / (rec["launch_aware"]["ms_per_call"] * 1e-3)
        rec["modelled_speedup_over_n1"] = n1_ms / rec["launch_aware"]["ms_per_call"]
        out["per_world"][str(G)] = rec
    del net, x, full_logits
    torch.cuda.empty_cache()
    return out


def pmc_traffic(workload, kernel_name, enc_patches, n_launch):
    """HBM bytes from the PMC counters.  They cannot be read from inside this process; they are collected with separate
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of this very command (tools/pmc_traffic.py, gfx950 correction
    applied) and committed under profiles/pmc_traffic.json, one record per workload, each with a hash of the kernel sources
    it was measured on: a figure measured on another version of the kernels is reported as null, not silently reused.
    Returns (bytes, note): bytes per launch of the dominant kernel where the encoder is ONE kernel (the fused trunk),
    bytes per step over all kernels otherwise."""
    try:
        allrec = json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json")))
        pmc = allrec.get(workload) if "kernel" not in allrec else (allrec if workload == "mnist" else None)
        if pmc is None:
            return None, "no PMC record for this workload"
        h = hashlib.sha256()
        for src in pmc["source"].split(","):
            h.update(open(os.path.join(REPO, "ips_amd", "csrc", src), "rb").read())
        if h.hexdigest()[:16] != pmc["source_sha16"]:
            return None, "profiles/pmc_traffic.json[%s] was measured on another version of %s (stale)" % (workload, pmc["source"])
        stamp = "; builder-run rocprofv3 passes on kernel sources %s sha16 %s = the sources of this run" % (pmc["source"], pmc["source_sha16"])
        if "hbm_bytes_per_step" in pmc:
            return pmc["hbm_bytes_per_step"], "bytes per step, all kernels of ips() (PMC, profiles/pmc_traffic.json%s)" % stamp
        if any(k in (kernel_name or "") for k in pmc["kernel"].split("|")) and enc_patches == pmc["patches_per_launch"] * n_launch:
            return pmc["hbm_bytes_per_launch"], "bytes per launch (PMC, profiles/pmc_traffic.json%s)" % stamp
        return None, "profiles/pmc_traffic.json holds %s at %d patches per launch" % (pmc["kernel"], pmc["patches_per_launch"])
    except (OSError, KeyError, ValueError):
        return None, "no PMC record"


class Ctx:
    """What every leg of one bench.py process shares: the process group and the device."""

    def __init__(self, world, rank, dev, share):
        self.world, self.rank, self.dev, self.share = world, rank, dev, share
        self.cpu_job = None


# Driver-run coverage: the default single-GPU invocation times the headline AND every other BASELINE
# configuration that fits one GPU, each with its own parity / roofline objects, as `also_measured.<leg>`:
#   b1          SURVEY d-1's primary shape: one ips() call on ONE image of configs[1]
#   mnist3000   configs[2] on one GPU (16 x 10,000 patches)
#   cam         configs[3], one slide per call (the reference's B_seq = 1)
#   cam_x16     configs[3], 16 slides per call (the reference's B = 16 in one call)
#   cam_native  the reference's shipped CAMELYON sizes (M = I = 5000)
ALSO_LEGS = (("b1", "b1", None), ("mnist3000", "mnist3000", None), ("cam", "cam", None), ("cam_x16", "cam", 16),
             ("cam_native", "cam_native", None), ("traffic", "traffic", None), ("native50", "native50", None))


def make_input(conf, name, B, batch, dev_for_extra):
    """The fixture's batch (host, seeded numpy stream).  More feature slides than the fixture holds (cam, --batch 16) are
    drawn on the device - relu(N(0,1)) like the fixture's, other seeds - because 2 x 10^9 host-side normals would take longer
    than every timed leg together; slide 0 stays the fixture's."""
    from ips_amd import synth
    if (not conf.is_image) and batch > B:
        x0 = synth.make_patches(conf, B, seed=21)
        return x0, batch - B
    return synth.make_patches(conf, max(batch, B), seed=21), 0


def measure(args, ctx, name, batch=None, steps=20, warmup=5, cpu_seconds=0.0, headline=True):
    """One workload: build it, warm up, time it both ways (SURVEY d-1: device sync around every call, median = `value`;
    and the same K calls back to back between two fences = `value_pipelined`), check the selection against the
    reference-recorded fixture, price the encoder launches.  Returns the JSON object on rank 0 (None elsewhere)."""
    world, rank, dev, share = ctx.world, ctx.rank, ctx.dev, ctx.share
    from ips_amd import hip, synth
    from ips_amd import dist as ipsd
    from ips_amd.architecture import IPSNet

    weak = world > 1 and args.scaling == "weak"
    fixture, images = name, None
    if name == "b1":
        conf, B = synth.bench_workload("mnist")
        fixture, images = "mnist", slice(0, 1)
    else:
        conf, B = synth.bench_workload(name)
    if weak:                                                    # the image grows with the node; 4 GPUs = configs[2]
        conf, B = synth.bench_workload("mnist")
        conf = conf.clone(N=PATCHES_PER_GPU_WEAK * world)
        fixture = "mnist3000" if conf.N == 10000 else None
    batch = 1 if name == "b1" else (batch or B)
    n_total = conf.N
    net = synth.fill_weights(IPSNet(dev, conf), 7).to(dev).eval()
    x, n_extra = make_input(conf, name, B, batch, dev)          # the fixture's batch; --batch may take a prefix of it
    if name == "b1":
        x = x[:1]
    elif batch < x.shape[0]:
        x, images = x[:batch], slice(0, batch)
    if batch > B and conf.is_image:
        fixture = None
    x_host = x
    if world > 1:
        shard = ipsd.shard_plan(net, x.shape[0], n_total, world, tuple(x.shape[2:]))     # launch-aware cuts (dist.py)
        mine = shard.indices(rank)
        x = x[:, mine].contiguous()                             # this rank's shard of every image
    n_mine = x.shape[1]
    if args.storage != "f32":
        if args.precision == "fp32":
            print("--storage %s needs --precision bf16 or fp32x3" % args.storage, file=sys.stderr)
            sys.exit(2)
        x = x.to({"bf16": torch.bfloat16, "f16": torch.float16}[args.storage])
    x = x.pin_memory() if args.lazy else x.to(dev)              # headline: resident in HBM before the timed region
    if n_extra:
        g = torch.Generator(device=dev)
        g.manual_seed(2100)
        more = torch.relu(torch.randn((n_extra,) + tuple(x.shape[1:]), generator=g, device=dev, dtype=torch.float32))
        x = torch.cat((x, more), 0)
        del more

    timings = [] if world > 1 else None
    if world == 1:
        def step():
            return net.ips(x)
    else:
        def step():
            return ipsd.ips_sharded(net, x, n_total, timings=timings)

    step()                                                      # builds the encoder plan
    if steps is None:
        # an `also_measured` leg: enough calls for ~0.3 s of timed work each way and ~0.1 s of warm-up (short calls - one
        # image, one slide - otherwise sit on the tail of the clock ramp and the allocator's first-use effects)
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        t1 = max(time.perf_counter() - c0, 1e-5)
        steps = int(min(40, max(10, 0.3 / t1)))
        warmup = int(min(15, max(3, 0.1 / t1)))
    # time the encoder launches with HIP events on the stream they run on (installed BEFORE the warm-up so that the
    # warm-up steps run exactly what the timed steps run, event creation included)
    enc_events = []
    streamed = []

    def timed(fn, count, tag=None):
        def wrapper(t, *a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(t, *a, **kw)
            e1.record()
            enc_events.append((e0, e1, count(t, a)))
            if tag:
                streamed.append(tag)
            return out
        return wrapper

    native_slots = []                                           # (slot, rows) of calls the library enqueued whole (csrc/call.hip)
    if not args.no_kernel_events:
        # calls with a resident loop are enqueued by ONE library call: the producer's launch is bracketed by an event pair the
        # library owns (ipsx_ips_call_run's timing_slot), read back after the timed region
        def hook(rows):
            slot = len(native_slots) % 64
            native_slots.append((slot, rows))
            return slot
        net.selection.timing_hook = hook
        plan = net._plan
        plan.encode = timed(plan.encode, lambda t, a: t.shape[0])
        plan.encode_indexed = timed(plan.encode_indexed, lambda t, a: a[0].numel())     # the overlapped path encodes in parts
        plan.stream = timed(plan.stream, lambda t, a: t.shape[0], 1)                    # features: ONE persistent projector launch
        plan.image_stream = timed(plan.image_stream, lambda t, a: t.shape[0], 2)        # one image: trunk + logits, ONE launch
    for _ in range(max(warmup, 1)):
        step()
    torch.cuda.synchronize()
    enc_events.clear()
    native_slots.clear()
    if timings is not None:
        timings.clear()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # the interpreter's cyclic garbage collector must not pick the timed region for a full collection: with the host
    # several steps ahead of the GPU, a collection that ends up waiting on the device costs tens of milliseconds once
    import gc
    gc.collect()
    gc.freeze()
    fence()
    t0 = time.perf_counter()
    per_step = []
    prof = None
    if os.environ.get("IPSX_BENCH_DEBUG") == "2":
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    for _ in range(steps):
        h0 = time.perf_counter()
        step()
        per_step.append(time.perf_counter() - h0)
    host_enqueue = time.perf_counter() - t0                     # the host's share: how long it took to ENQUEUE the steps
    if prof is not None:
        import pstats
        prof.disable()
        pstats.Stats(prof, stream=sys.stderr).sort_stats("tottime").print_stats(12)
    if os.environ.get("IPSX_BENCH_DEBUG") and rank == 0:
        print("host ms per step:", " ".join("%.2f" % (1e3 * v) for v in per_step), file=sys.stderr)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    enc_ms = sum(a.elapsed_time(b) for a, b, _ in enc_events)
    enc_patches = sum(n for _, _, n in enc_events)
    n_launch = max(len(enc_events), 1)
    if native_slots:                                            # the last <= 64 calls' producer launches (the slots are a ring)
        import ctypes as C
        ms = C.c_float()
        last = native_slots[-64:]
        for slot, rows in last:
            hip._ck(hip.lib().ipsx_ips_call_elapsed(slot, C.byref(ms)), "ipsx_ips_call_elapsed")
            enc_ms += ms.value
            enc_patches += rows
        n_launch = len(enc_events) + len(last)
        streamed.append(2 if name in ("b1",) or conf.is_image else 1)
    net.selection.timing_hook = None
    achieved = enc_patches * FLOP_PER_PATCH[name] / (enc_ms * 1e-3) / 1e12 if enc_ms else 0.0       # (0: --no-kernel-events)
    phases = ipsd.phase_ms(timings) if timings else None
    # (more images than the fixture holds, --batch: its images are the first of the batch - the generator draws image by image)
    par = parity(fixture, net.last_mem_idx[:B] if (world == 1 and batch > B) else net.last_mem_idx, images) if fixture else None

    # SURVEY d-1's protocol - the line's `value`: every one of the K calls bracketed by a fence (barrier + device sync) in
    # front and a device sync behind, the MAX over ranks of each call's time, the median over the K calls
    lat = []
    for _ in range(steps):
        fence()
        c0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        lat.append(1e3 * (time.perf_counter() - c0))
    fence()
    if world > 1:
        t = torch.tensor(lat, dtype=torch.float64, device="cpu" if share else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        lat = [float(v) for v in t.tolist()]
    gc.unfreeze()
    med_ms = statistics.median(lat)

    if world > 1:                                               # every rank's verdict and phase times travel to rank 0
        mine_rec = {"rank": rank, "patches_per_image": n_mine, "phases": phases,
                    "indices_equal": par["indices_equal"] if par else None,
                    "device_index": dev.index, "device": torch.cuda.get_device_name(dev)}
        recs = [None] * world
        dist.all_gather_object(recs, mine_rec)
    kernel_name = hip.encoder_kernel_name(net._plan)
    if streamed and streamed[0] == 2:
        kernel_name = "fused_trunk_stream_kernel (the fused trunk + logits, four / two patches per pull, patches published as they complete)"
    elif streamed:
        kernel_name = "projector_stream_kernel (row moments + Linear + BatchNorm + ReLU + logits per tile, rows published as they complete)"
    traffic, traffic_note = pmc_traffic(name, kernel_name, enc_patches, n_launch)
    if not (args.precision == "fp32" and not args.dedup_blank and not args.lazy and world == 1 and batch == (1 if name == "b1" else B)):
        traffic = None

    out = None
    if rank == 0:
        patches_per_step = batch * n_total
        per_patch_bytes = (conf.n_chan_in * (conf.patch_size[0] * conf.patch_size[1] if conf.is_image else 1) + conf.D) * 4
        out = {
            "metric": "patches scored/sec (no-grad IPS loop)",
            "value": patches_per_step / (med_ms * 1e-3),
            "unit": "patches/s",
            "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": med_ms,
            "higher_is_better": True,
            # the SAME 16 x 2500 patches at every N (BASELINE "MegaMNIST-1500 @ 1/2/4/8 GPU"): the N = 1 line is the first
            # point of a strong-scaling curve; --scaling weak (2500 patches of every image per GPU) says so at every N
            "scaling": "weak" if args.scaling == "weak" else "strong",
            "vs_baseline": None,
            "dtype": {"fp32": "f32", "fp32x3": "f32 as 3 bf16 terms, 6 bf16 MFMA products, f32 accumulate",
                      "bf16": "bf16 operands / f32 accumulate"}[args.precision],
            "data": "synthetic",
            "config": {"workload": "%s, B=%d, M=%d, I=%d, n_token=%d, %s, eager%s"
                                   % (LABEL[name] if not weak else "Megapixel-MNIST %d patches of 1x32x32 per image (2500 per GPU)" % n_total,
                                      batch, conf.M, conf.I, conf.n_token,
                                      "use_pos" if conf.use_pos else "no pos-enc",
                                      "" if world == 1 else ", %d of the %d patches of every image per GPU" % (n_mine, n_total)),
                       "parallelism": "patch-sharded x%d, %d all-gathers of logits, scan overlapped" % (world, len(shard.its) - 1) if world > 1 else "single GPU",
                       "dedup_blank": bool(args.dedup_blank), "lazy_host_patches": bool(args.lazy),
                       "patch_storage": args.storage,
                       # compute units the selection loop of ONE image keeps (1; a team of up to 8 workgroups for candidate
                       # sets beyond the LDS - the shipped CAMELYON M = I = 5000: csrc/scan_large_team.h)
                       "loop_workgroups_per_image": hip.scan_workgroups_per_image(batch, conf.M, conf.I, net.transf.crs_attn.H,
                                                                                   conf.n_token)},
            "timing": "value = patches per call / median over the %d timed calls, each bracketed by barrier + device sync "
                      "(SURVEY d-1; max over ranks per call); value_pipelined = the same %d calls enqueued back to back "
                      "between two fences" % (steps, steps),
            "value_pipelined": patches_per_step * steps / elapsed,
            "ms_per_step_pipelined": 1e3 * elapsed / steps,
            "ms_per_call_median_synced": med_ms,
            "ms_per_call_min_synced": min(lat),
            "host_enqueue_ms_per_step": 1e3 * host_enqueue / steps,
            "parity": par,
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "traffic": traffic,
                         "traffic_unit": traffic_note,
                         "algorithmic_bytes": enc_patches / n_launch * per_patch_bytes,
                         "kernel": kernel_name,
                         "launch_ms": enc_ms / n_launch,
                         "patches_per_launch": enc_patches / n_launch},
        }
        if n_extra:
            out["config"]["workload"] += " (slide 0 = the fixture's, the other %d drawn on the device: relu(N(0,1)))" % n_extra
        if world > 1:
            out["parity_all_ranks"] = all(r["indices_equal"] for r in recs) if par else None
            paid, ideal = shard.rounds(0)
            out["shard_plan"] = {"first_iteration_of_every_part": shard.its, "rank0_launch_patches": shard.launches(0),
                                 "rounds_paid": paid, "rounds_ideal_one_launch": ideal, "model": shard.model.kind}
            out["per_rank"] = recs
            out["distributed"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                                  "devices": [r["device"] for r in recs], "one_gpu_per_rank": not share}
        # the WHOLE call priced against the work it executes: encoder FLOP + the logits' folded-query contraction
        # (2 * D * H * n_token per patch - the reference's per-iteration K projection is not executed here)
        call_flop = FLOP_PER_PATCH[name] + 2 * conf.D * conf.H * conf.n_token
        call_tflops = patches_per_step / (med_ms * 1e-3) * call_flop / 1e12
        call_peak = {"fp32": FP32_MFMA_PEAK_TFLOPS, "fp32x3": BF16_MFMA_PEAK_TFLOPS / 6, "bf16": BF16_MFMA_PEAK_TFLOPS}[args.precision] * world
        if args.dedup_blank:                        # the encoder skipped most patches: no meaningful FLOP rate
            call_tflops = None
        out["roofline_call"] = {"bound": "mfma", "achieved": call_tflops, "peak": call_peak, "unit": "TFLOP/s",
                                "frac": call_tflops / call_peak if call_tflops is not None else None,
                                "what": "patches/s of the whole ips() call (synced median) x executed FLOP per patch (encoder %d + logits %d)"
                                        % (FLOP_PER_PATCH[name], 2 * conf.D * conf.H * conf.n_token),
                                "algorithmic_bytes_per_step": patches_per_step * per_patch_bytes}
        if args.precision == "bf16":    # priced against the dense bf16 MFMA peak
            out["roofline"]["peak"] = BF16_MFMA_PEAK_TFLOPS
            out["roofline"]["frac"] = achieved / BF16_MFMA_PEAK_TFLOPS
        if args.precision == "fp32x3":  # six bf16 products per fp32 product: the bf16 peak / 6 bounds the algorithmic rate
            out["roofline"]["peak"] = BF16_MFMA_PEAK_TFLOPS / 6
            out["roofline"]["frac"] = achieved / (BF16_MFMA_PEAK_TFLOPS / 6)
            out["roofline"]["note"] = "algorithmic fp32 FLOP priced against dense bf16 peak / 6 (6 MFMA products per fp32 product)"
        if args.no_kernel_events:
            out["roofline"].update({"achieved": None, "frac": None, "traffic": None, "launch_ms": None,
                                    "note": "--no-kernel-events: the encoder launches were not timed"})
        elif args.dedup_blank:      # fewer patches are encoded than scored: an algorithmic FLOP rate would be wrong
            out["roofline"].update({"achieved": None, "frac": None, "traffic": None,
                                    "note": "blank-patch dedup: encoder ran on %d of %d patches per launch"
                                            % (int(net._plan.n_encoded.item()), enc_patches // n_launch)})
        if headline and world == 1 and name == "mnist" and args.precision == "fp32" and not (args.dedup_blank or args.lazy) and batch == B:
            out["also_measured"] = {p: measure_precision(net, x, steps, p, fixture) for p in ("fp32x3", "bf16")}
            # BASELINE configs[4] as written: half-precision STORAGE of the patches + the bf16 matrix pipe
            out["also_measured"]["bf16_f16"] = measure_precision(net, x, steps, "bf16", fixture, storage="f16")
        if world == 1 and cpu_seconds > 0:
            ctx.cpu_job = (conf, x_host, cpu_seconds)           # timed by main(), last and under a guard (see there)
        out["host"] = host_description()
    del net, x, x_host
    torch.cuda.empty_cache()
    return out


def slim(rec):
    """What an `also_measured` leg keeps of a full record."""
    keep = ("value", "unit", "ms_per_step", "value_pipelined", "ms_per_step_pipelined", "steps", "warmup", "parity",
            "roofline", "roofline_call", "host_enqueue_ms_per_step")
    out = {k: rec[k] for k in keep if k in rec}
    out["workload"] = rec["config"]["workload"]
    return out


EXIT_WATCHDOG = 3        # a leg (or the CPU baseline) did not come back: the partial line is printed, the exit code says so


def self_launch(args):
    """``python bench.py --gpus N`` (N > 1) without a launcher's environment: start the N ranks as a CHILD
    ``python -m torch.distributed.run`` of this very command and hand its exit code on.  The parent has not touched the GPU
    (no HIP call, no ``torch.cuda.is_available()``) and never does - it only waits; rank 0 of the child prints the JSON
    line on the stdout they share.  (Never an exec: a process that has initialised the GPU must not be replaced, and the
    parent may have been started under a profiler that has.)"""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC: RCCL between processes needs it on this image
    env.setdefault("OMP_NUM_THREADS", "8")
    sys.stdout.flush()
    proc = subprocess.Popen(cmd, env=env)
    try:
        rc = proc.wait()
    except KeyboardInterrupt:
        proc.terminate()
        rc = proc.wait()
    if rc != 0:
        print("bench.py: the %d-rank child run exited with code %d" % (args.gpus, rc), file=sys.stderr)
    sys.exit(rc)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)                                       # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d (the launcher's --nproc-per-node must equal --gpus)" % (args.gpus, world),
                  file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (the HIP path has no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    # IPSX_BENCH_SHARE_GPU=1 (testing aid for 1-GPU boxes): every rank uses cuda:0 and the process group is gloo, so
    # the N > 1 code path of this script can be exercised without N GPUs; RCCL needs one GPU per rank
    share = os.environ.get("IPSX_BENCH_SHARE_GPU") == "1"
    local = 0 if share else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from ips_amd import hip

    hip.lib()                                                   # fail loudly if the extension is missing
    # torch sizes its CPU pools by the machine; a container that may run fewer CPUs than it shows freezes the whole process
    # for the rest of a quota period when the pools burn it (building a net, drawing the inputs) - also in the middle of a
    # timed call (DESIGN 6 "Soak").  The GPU legs need no CPU pool; the CPU baseline sets its own counts
    host = host_description()
    torch.set_num_threads(max(1, min(8, int(host["cpu_quota"] or host["cpus_available"]) // max(world, 1) or 1)))
    if args.dedup_blank:
        os.environ["IPSX_DEDUP_BLANK"] = "1"
    os.environ["IPSX_PRECISION"] = args.precision
    name = args.config or "mnist"                               # ONE workload at every N: BASELINE configs[1]
    if world > 1 and name not in ("mnist", "mnist3000"):
        print("secondary configs are single-GPU measurements", file=sys.stderr)
        sys.exit(2)
    ctx = Ctx(world, rank, dev, share)
    # a run that stops making progress says where (all threads' stacks on stderr) instead of sitting there until the
    # caller's limit: one round-4 run on a freshly leased box hung for 15 minutes without a byte of output and could not be
    # reproduced on the next two boxes
    import faulthandler
    faulthandler.dump_traceback_later(args.watchdog, exit=True)
    out = measure(args, ctx, name, batch=args.batch, steps=args.steps, warmup=args.warmup, cpu_seconds=args.cpu_seconds, headline=True)
    faulthandler.cancel_dump_traceback_later()
    plain = (args.config is None and args.batch is None and args.precision == "fp32"
             and args.storage == "f32" and not (args.dedup_blank or args.lazy or args.no_kernel_events))
    default_run = world == 1 and plain
    if world > 1 and plain and args.scaling == "strong":
        # the first point of the curve, measured in THIS run on THIS box: rank 0 alone, the same 16 x 2500 patches, the
        # single-GPU path (the other ranks wait at the barrier); then configs[2] sharded the same way
        faulthandler.dump_traceback_later(args.watchdog, exit=True)
        if rank == 0:
            solo = measure(args, Ctx(1, 0, dev, share), name, batch=None, steps=min(args.steps, 10), warmup=min(args.warmup, 3),
                           cpu_seconds=0.0, headline=False)
            out["n1_value_same_workload"] = solo["value"]
            out["n1_ms_per_step_same_workload"] = solo["ms_per_step"]
            out["speedup_over_n1_same_run"] = out["value"] / solo["value"]
        dist.barrier()
        if args.also != "none":
            rec = measure(args, ctx, "mnist3000", batch=None, steps=min(args.steps, 10), warmup=min(args.warmup, 3),
                          cpu_seconds=0.0, headline=False)
            if rank == 0:
                leg = slim(rec)
                leg["parity_all_ranks"] = rec.get("parity_all_ranks")
                leg["per_rank"] = rec.get("per_rank")
                out.setdefault("also_measured", {})["mnist3000"] = leg
        faulthandler.cancel_dump_traceback_later()
    if default_run and args.also != "none":
        import threading
        legs = [l for l in ALSO_LEGS if args.also == "all" or l[0] in args.also.split(",")]

        # the headline is measured: a leg that does not come back must not cost it - the line goes out without the rest,
        # and the exit code (EXIT_WATCHDOG) tells the caller that a GPU leg hung
        def bail(leg):
            out["also_measured_incomplete"] = "leg %r did not finish within %d s; the line was printed without it and the legs behind it" % (leg, args.leg_timeout)
            print(json.dumps(out), flush=True)
            faulthandler.dump_traceback(file=sys.stderr)
            os._exit(EXIT_WATCHDOG)
        for leg, cfg, b in legs:
            guard = threading.Timer(args.leg_timeout, bail, kwargs={"leg": leg})
            guard.daemon = True
            guard.start()
            rec = measure(args, ctx, cfg, batch=b, steps=None if args.also_steps <= 0 else min(args.steps, args.also_steps),
                          warmup=min(args.warmup, 3), cpu_seconds=0.0, headline=False)
            guard.cancel()
            out.setdefault("also_measured", {})[leg] = slim(rec)
        if args.also == "all" or "rank_shard_model" in args.also.split(","):
            guard = threading.Timer(args.leg_timeout, bail, kwargs={"leg": "rank_shard_model"})
            guard.daemon = True
            guard.start()
            out["rank_shard_model"] = rank_shard_model(args, ctx)
            guard.cancel()
    if rank == 0 and getattr(ctx, "cpu_job", None) is not None:
        # the CPU baseline comes last and under a guard: it is the one leg whose duration this script does not control (a
        # host that throttles 64 threads down to a few cores can turn its thread table into minutes) - the measured GPU
        # numbers must not wait for it
        import threading
        conf_c, x_c, budget = ctx.cpu_job
        limit = int(6 * budget + 60)

        def bail_cpu():
            out["cpu_baseline"] = None
            out["cpu_baseline_note"] = "the CPU leg did not finish within %d s on this host; line printed without it" % limit
            print(json.dumps(out), flush=True)
            os._exit(EXIT_WATCHDOG)
        guard = threading.Timer(limit, bail_cpu)
        guard.daemon = True
        guard.start()
        out["cpu_baseline"] = cpu_baseline(conf_c, x_c, budget)
        guard.cancel()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
