"""Patch-sharded IPS over the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference is single-device (/root/reference/main.py:19-20); this is the one
multi-GPU mechanism the hot path admits without changing its result (SURVEY.md
section 8 e-2): in eval mode the encoder and a patch's attention logits are pure
per-patch functions, so the patch axis is sharded, the ranks exchange LOGITS
(H*T floats = 128 bytes per patch at the MNIST configuration - latency-bound on
xGMI) and every rank replays the identical selection loop, so all ranks hold the
same ``mem_idx`` as a single-GPU run, bit for bit.

Layout (``partition``): the patch axis is cut into a few PARTS at chunk boundaries of
the selection loop; every part is split evenly over the ranks.  Per part: each rank
encodes and scores its piece; ONE ``all_gather_into_tensor`` assembles the part's
logits and the loop iterations that part makes possible follow it, both on a
high-priority side stream (``ipsx_scan_range`` resumes from the memory indices), while
the encoder is already busy with the next part on the main stream.  Only the last part's iterations are exposed, which
keeps the sequential loop - whose length grows with the image - off the critical path.
The M winning patches are then assembled with one small ``all_reduce`` of zero-filled
owner contributions (exact: x + 0 = x).

On CPU tensors (gloo; the world_size-2 tests) the same partitioning runs with the
ATen path: embeddings are all-gathered instead of logits and the reference loop runs
on them.
"""

import math

import torch
import torch.distributed as dist

from . import hip

PARTS = 4
DEFAULT_UNITS = 256            # compute units of an MI355X in SPX mode: what a plan assumes where no GPU is visible (CPU tests)

# Cumulative share of the loop iterations per part - the cut used where the launch model has nothing to say (encoders whose
# cost is linear in the rows) and one of the candidates everywhere else.  The encoder works through the parts in order while
# the scan of the previous part runs beside it; what stays exposed is the scan of the LAST part, so the parts shrink towards
# the end.
PART_SHARES = (0.5, 0.8, 0.95, 1.0)

# When the loop, not the encoder, is the long pole (feature inputs: a projector row costs less than its share of a
# loop iteration) the order flips: a small first part lets the loop start early, and every later part is ready
# before the loop reaches it.  Measured at the CAMELYON shape (ms per slide): one part 4.18, two halves 3.51, four
# equal 3.15, (0.1, 0.28, 0.6, 1) 3.09, these 3.02, eight equal 3.23 (launch and restart overheads take over).
PART_SHARES_LOOP_BOUND = (0.15, 0.4, 0.7, 1.0)


def part_iterations(n_iter, parts=PARTS, shares=None):
    """First iteration of every part (strictly increasing where possible), plus n_iter at the end."""
    P = max(1, min(parts, n_iter))
    shares = shares if shares is not None else PART_SHARES
    shares = shares if P == len(shares) else [(k + 1) / P for k in range(P)]
    its = [0]
    for k in range(P):
        lo = its[-1] + 1                                  # every part gets at least one iteration
        hi = n_iter - (P - 1 - k)
        its.append(min(max(round(shares[k] * n_iter), lo), hi))
    its[-1] = n_iter
    return its


class LaunchModel:
    """What one rank's launches cost, in microseconds: just enough of a model to place the cuts of the patch axis.

    The fused trunk works in ROUNDS of 8 patches per compute unit (two workgroups of four wavefronts = patches per unit,
    csrc/fused_trunk.hip::fused_launch): a launch of n patches costs ``n // round`` whole rounds plus what its remainder
    costs - up to a quarter round through the two-wavefronts-per-patch kernel (0.28 of a round), up to a half round at one
    wavefront per SIMD (0.54), up to three quarters through the pair kernel at three per SIMD (0.80), a whole round beyond
    (measured, patches -> ms on 256 units: 512 0.15, 1024 0.29, 1536 0.43, 2048 0.54).  A cut of the patch axis that leaves
    a rank 2,688 patches pays two rounds for 1.3 rounds of work; the same rows cut at 2,048 + ... pay for what they are.
    ``round == 0``: an encoder whose cost is linear in its rows (layer-by-layer trunks, the projector).

    Side stream, per part: ``t_xchg`` (logits kernel, all-gather, the copy into the call's logits, the loop's launch) and
    ``t_iter`` per loop iteration - times ``BESIDE`` while the encoder still has launches in flight: a loop workgroup that
    shares its compute unit with fp32 MFMA wavefronts takes 24 us per iteration instead of 6.7 (tools/scan_beside.py).
    The numbers are rough on purpose; the plan they lead to is priced on the GPU by
    ``bench.py``'s ``rank_shard_model`` leg."""

    REST_STEPS = ((0.25, 0.28), (0.5, 0.54), (0.75, 0.80), (1.0, 1.0))
    BESIDE = 3.5

    def __init__(self, round=0, t_round=0.0, t_row=0.0, t_launch=10.0, t_iter=7.0, t_xchg=60.0, kind="rows"):
        self.round, self.t_round, self.t_row = int(round), float(t_round), float(t_row)
        self.t_launch, self.t_iter, self.t_xchg, self.kind = float(t_launch), float(t_iter), float(t_xchg), kind

    def rounds(self, n):
        """Rounds a launch of n rows costs (fractional: the remainder's kernels); 0 for a linear encoder."""
        if n <= 0 or not self.round:
            return 0.0
        full, rest = divmod(int(n), self.round)
        if rest:
            f = rest / self.round
            full += next(c for lim, c in self.REST_STEPS if f <= lim)
        return float(full)

    def encode_us(self, n):
        if n <= 0:
            return 0.0
        return self.t_launch + (self.rounds(n) * self.t_round if self.round else n * self.t_row)

    def key(self):
        return (self.kind, self.round, round(self.t_round, 3), round(self.t_row, 6), self.t_launch, round(self.t_iter, 3), self.t_xchg)


def default_units():
    """Compute units of the GPU this process drives (every rank of a node sees the same kind), DEFAULT_UNITS without one."""
    if torch.cuda.is_available():
        return hip.device_geometry(torch.device("cuda", torch.cuda.current_device())).cus
    return DEFAULT_UNITS


def loop_iteration_us(M, I, H, T):
    """Rough duration of one iteration of the selection loop kernels: ~4 us at the CAMELYON shape and
    6.7 us at the MNIST shape (both 4,096 logits per iteration), 92 us at 10,000 candidates of 8 logits."""
    L, R = M + I, H * T
    if L > 1024:
        return 92.0 * L * R / 80000.0
    if (M, I, H, T) == (256, 256, 8, 1):                  # scan_cam_kernel's shape
        return 4.2
    return 2.5 + L * R / 1000.0


# seconds of fp32 matrix-pipe time per FLOP and compute unit at the rates the kernels reach (157.3 TFLOP/s over 256 units)
_UNIT_FLOPS = 157.3e12 / 256


def _conv_macs(encoder, patch_shape):
    """Multiply-adds of one patch through an image trunk (a meta-device pass over a copy of the modules)."""
    import copy
    macs = [0]

    def hook(m, inp, out):
        macs[0] += out[0].numel() * m.in_channels // m.groups * m.kernel_size[0] * m.kernel_size[1]
    try:
        enc = copy.deepcopy(encoder).to("meta").eval()
        hs = [m.register_forward_hook(hook) for m in enc.modules() if isinstance(m, torch.nn.Conv2d)]
        with torch.no_grad():
            enc(torch.empty((1,) + tuple(patch_shape[-3:]), device="meta"))
        for h in hs:
            h.remove()
    except Exception:                                     # a trunk the meta device cannot run: a mid-sized guess
        return 100e6
    return float(macs[0])


def launch_model(net, patch_shape, units=None, precision=None):
    """The ``LaunchModel`` of ``net``'s encoder on patches of ``patch_shape`` (..., C, h, w) or (..., F)."""
    units = int(units or default_units())
    ca = net.transf.crs_attn
    t_iter = loop_iteration_us(net.M, net.I, ca.H, ca.n_token)
    precision = precision or hip.precision()
    if net.is_image and tuple(patch_shape[-3:]) == (1, 32, 32) and len(net.encoder) == 7:
        # the fused trunk: 8 patches per unit and round.  fp32: 37.26 MFLOP per patch at 0.88 of the unit's pipe; the split
        # trunks on the bf16 pipe: 26 M / 6.8 M patches/s per 256 units
        if precision == "bf16":                   # third build: workgroups of EIGHT patches, two per unit; 36 M patches/s
            return LaunchModel(round=16 * units, t_round=16 * 256 / 36.0, t_iter=t_iter, kind="fused32:bf16")
        t_round = {"fp32x3": 8 * 256 / 6.8}.get(precision, 8 * 37257216 / (_UNIT_FLOPS * 0.88) * 1e6)
        return LaunchModel(round=8 * units, t_round=t_round, t_iter=t_iter, kind="fused32:" + precision)
    if net.is_image:
        key = ("macs", tuple(patch_shape[-3:]))
        cache = net.__dict__.setdefault("_launch_model_cache", {})
        if key not in cache:
            cache[key] = _conv_macs(net.encoder, patch_shape)
        t_row = 2 * cache[key] / (_UNIT_FLOPS * units * 0.8) * 1e6
        n_conv = sum(1 for m in net.encoder.modules() if isinstance(m, torch.nn.Conv2d))
        return LaunchModel(t_row=t_row, t_launch=8.0 * n_conv, t_iter=t_iter, kind="layers")
    F = int(patch_shape[-1])
    return LaunchModel(t_row=2.0 * F * net.D / (_UNIT_FLOPS * units * 0.75) * 1e6, t_launch=15.0, t_iter=t_iter, kind="rows")


def _plan_cost(its, N, M, I, world, B, model):
    """Microseconds from the first launch to the end of the last loop iteration on one rank: the encoder works through the
    parts on the main stream, a part's exchange and loop iterations follow it on the side stream."""
    P = len(its) - 1
    done, main = [], 0.0
    for k in range(P):
        e0 = 0 if k == 0 else min(N, M + its[k] * I)
        e1 = N if k == P - 1 else min(N, M + its[k + 1] * I)
        main += model.encode_us(B * -(-(e1 - e0) // world))
        done.append(main)
    side = 0.0
    for k in range(P):
        side = max(side, done[k]) + model.t_xchg
        n = its[k + 1] - its[k]
        slow = model.t_iter * model.BESIDE
        beside = min(n, int(max(0.0, main - side) // slow))         # iterations that run while the encoder is still busy
        side += beside * slow + (n - beside) * model.t_iter
    return side


_PLAN_CACHE = {}


def plan_iterations(N, M, I, world, B=1, model=None, parts=PARTS):
    """First loop iteration of every part (``its[0] = 0``, ``its[-1] = n_iter``), chosen for the LAUNCHES it gives a rank.

    Without a model (or with a linear one and nothing to gain): the fixed shares.  With the fused trunk's model the cuts are
    searched: from the previous cut, the next one is the last chunk boundary up to which the rank's piece of the part -
    ``B * ceil(rows / world)`` patches - still fits r whole rounds (or r rounds and one of the remainder's steps), for every
    r; the plan whose simulated time (``_plan_cost``) is smallest wins, fewer parts on a tie.  So a part is a whole number
    of rounds on every rank wherever the sizes allow it, the last part takes what is left, and a rank with less than a
    handful of rounds gets two parts or one.  Chunk boundaries are the only cuts: the result does not depend on them."""
    n_iter = math.ceil((N - M) / I)
    if model is None:
        return part_iterations(n_iter, parts)
    key = (N, M, I, world, B, parts, model.key())
    hit = _PLAN_CACHE.get(key)
    if hit is not None:
        return list(hit)
    cands = {tuple(part_iterations(n_iter, p, sh)) for p in range(1, parts + 1)
             for sh in (PART_SHARES, PART_SHARES_LOOP_BOUND, None)}
    if model.round:
        def piece(it0, it1):
            e0 = 0 if it0 == 0 else M + it0 * I
            return B * -(-(min(N, M + it1 * I) - e0) // world)

        def last_fitting(it0, limit):
            """Largest it1 in (it0, n_iter) whose piece stays within ``limit`` patches, or None."""
            e0 = 0 if it0 == 0 else M + it0 * I
            rows = (limit // B) * world                    # B * ceil(rows / world) <= limit  <=>  rows <= (limit // B) * world
            it1 = min((e0 + rows - M) // I, n_iter - 1)
            return it1 if it1 > it0 else None

        total_rounds = max(1.0, piece(0, n_iter) / model.round)

        def grow(prefix, left):
            cands.add(tuple(prefix + [n_iter]))
            if left <= 1:
                return
            it0 = prefix[-1]
            rest_rounds = piece(it0, n_iter) / model.round
            if rest_rounds <= 0.25:
                return
            limits = set()
            rs = range(0, int(rest_rounds) + 1)
            if len(rs) > 10:                               # long axes: a spread of round counts, not every one
                rs = sorted({int(round(rest_rounds * f)) for f in (0.08, 0.15, 0.25, 0.35, 0.5, 0.65, 0.8, 0.9, 0.95)} |
                            {1, 2, int(rest_rounds) - 1, int(rest_rounds)})
            for r in rs:
                for frac, _ in ((0.0, 0.0),) + (LaunchModel.REST_STEPS[:3] if total_rounds < 6 else ()):
                    lim = int((r + frac) * model.round)
                    if lim > 0:
                        limits.add(lim)
            nxt = {last_fitting(it0, lim) for lim in limits}
            for it1 in sorted(x for x in nxt if x is not None):
                grow(prefix + [it1], left - 1)

        grow([0], parts)
    best = min(cands, key=lambda c: (round(_plan_cost(c, N, M, I, world, B, model), 3), len(c), c))
    _PLAN_CACHE[key] = best
    return list(best)


def partition(N, M, I, world, parts=PARTS, B=1, model=None):
    """Cut [0, N) into parts at chunk boundaries and every part into `world` pieces.

    Returns ``(its, edges, piece)``: ``its[k]`` = first loop iteration of part k (``its[-1]`` = n_iter),
    ``edges[k]`` = first patch of part k, ``piece[k]`` = padded piece length of part k; rank r owns
    patches ``[edges[k] + r*piece[k], min(edges[k] + (r+1)*piece[k], edges[k+1]))`` of every part k.
    ``model`` (a ``LaunchModel``) and ``B`` (images per call) make the cuts launch-aware: ``plan_iterations``.
    """
    its = plan_iterations(N, M, I, world, B, model, parts)
    P = len(its) - 1
    edges = [0] + [min(N, M + it * I) for it in its[1:]]
    edges[-1] = N
    piece = [max(1, math.ceil((edges[k + 1] - edges[k]) / world)) for k in range(P)]
    return its, edges, piece


class ShardPlan:
    """The partition of one call shape: which patches every rank holds and in which order, which loop iterations follow
    which part.  Every rank builds the same plan from (net, B, N, world, units); ``ips_sharded`` cross-checks that once per
    shape.  ``indices(rank)`` is what the caller shards the patch axis with."""

    def __init__(self, N, M, I, world, B=1, model=None, parts=PARTS):
        self.N, self.M, self.I, self.world, self.B, self.model = N, M, I, world, B, model
        self.its, self.edges, self.piece = partition(N, M, I, world, parts, B, model)

    def spans(self, rank):
        """Global [lo, hi) ranges a rank owns, in local order (one per part, possibly empty)."""
        out = []
        for k, q in enumerate(self.piece):
            lo = min(self.edges[k] + rank * q, self.edges[k + 1])
            out.append((lo, min(lo + q, self.edges[k + 1])))
        return out

    def indices(self, rank):
        """1-D int64 tensor of the global patch indices a rank holds, in local order."""
        spans = self.spans(rank)
        return torch.cat([torch.arange(lo, hi, dtype=torch.int64) for lo, hi in spans]) if spans else torch.empty(0, dtype=torch.int64)

    def launches(self, rank):
        """Patches of every encoder launch of a rank (B images' pieces of every part)."""
        return [self.B * (hi - lo) for lo, hi in self.spans(rank)]

    def rounds(self, rank=0):
        """(rounds the rank's launches cost under the model, the same rows as ONE ideal launch) - fractional rounds."""
        n = self.launches(rank)
        if self.model is None or not self.model.round:
            return 0.0, 0.0
        return sum(self.model.rounds(v) for v in n), sum(n) / self.model.round

    def cost_us(self):
        return _plan_cost(self.its, self.N, self.M, self.I, self.world, self.B, self.model) if self.model is not None else None

    def signature(self):
        """A number every rank must agree on (cross-checked by ``ips_sharded``)."""
        import zlib
        return zlib.crc32(repr((self.N, self.M, self.I, self.world, self.B, self.its, self.edges, self.piece)).encode())

    def owner_maps(self, device):
        """owner[j] = rank holding global patch j, lpos[j] = its position in that rank's local tensor."""
        owner = torch.empty(self.N, dtype=torch.int64)
        lpos = torch.empty(self.N, dtype=torch.int64)
        for r in range(self.world):
            base = 0
            for lo, hi in self.spans(r):
                owner[lo:hi] = r
                lpos[lo:hi] = torch.arange(base, base + hi - lo)
                base += hi - lo
        return owner.to(device), lpos.to(device)


def shard_plan(net, B, N, world, patch_shape=None, units=None, parts=PARTS, precision=None):
    """The ``ShardPlan`` of a call of ``B`` images of ``N`` patches over ``world`` ranks with ``net``'s encoder.
    ``patch_shape``: (C, h, w) of a patch / (F,) of a feature row - needed for image encoders (the fused trunk takes
    1 x 32 x 32 patches only)."""
    if patch_shape is None:
        if net.is_image:
            raise ValueError("shard_plan: patch_shape=(C, h, w) is needed for an image encoder")
        patch_shape = (net.encoder[1].in_features,)
    model = launch_model(net, tuple(patch_shape), units, precision)
    return ShardPlan(N, net.M, net.I, world, B, model, parts)


def local_spans(N, M, I, rank, world, parts=PARTS):
    """Fixed-share partition (no launch model): global [lo, hi) ranges a rank owns."""
    return ShardPlan(N, M, I, world, parts=parts).spans(rank)


def local_indices(N, M, I, rank, world, parts=PARTS):
    """Fixed-share partition (no launch model): the global patch indices a rank holds, in local order.  Callers of
    ``ips_sharded`` use ``shard_plan(net, B, N, world, patch_shape).indices(rank)`` - or pass ``plan=`` explicitly."""
    return ShardPlan(N, M, I, world, parts=parts).indices(rank)


class LoopbackGroup:
    """A stand-in for a process group on ONE GPU (``bench.py``'s ``rank_shard_model`` leg): this process is rank ``rank`` of
    ``world``; the all-gather of a part's logits is a device copy of the same bytes out of ``full_logits`` (B, N, R) - the
    other ranks' pieces, computed beforehand - and the winners' all-reduce a device copy out of ``full_patches``.  It prices
    a rank's OWN launches (encoder pieces, logits, loop ranges, copies of the collectives' sizes); it is a model of a rank,
    not a measurement of a node: no link latency, no waiting for the slowest rank."""

    def __init__(self, world, rank, full_logits, full_patches):
        self.world, self.rank, self.full_logits, self.full_patches = world, rank, full_logits, full_patches


def _world_rank(group):
    if isinstance(group, LoopbackGroup):
        return group.world, group.rank
    return dist.get_world_size(group), dist.get_rank(group)


@torch.no_grad()
def ips_sharded(net, local_patches, N, group=None, timings=None, plan=None):
    """IPS over ``N`` patches of which this rank holds ``local_patches`` = patches[:, plan.indices(rank)].

    ``net`` is an ``IPSNet`` whose ``conf.N`` (positional table) is the GLOBAL ``N``.  ``plan``: the ``ShardPlan`` the
    caller sharded with; by default ``shard_plan(net, B, N, world, patch shape)`` - the launch-aware partition, the same
    on every rank (cross-checked once per shape).
    Shuffling is the caller's business here (shard after shuffling).  Returns
    ``(mem_patch, mem_pos, mem_idx)`` identical on every rank.

    ``timings`` (GPU path only): a list that receives one dict of HIP events per call - see ``phase_ms`` - so that a
    run can say where a rank's time went (encoder, exchange, loop, what of the loop stayed exposed).
    """
    world, rank = _world_rank(group)
    M, I, D = net.M, net.I, net.D
    if M >= N:
        raise ValueError("sharded IPS needs N > M")
    B = local_patches.shape[0]
    if plan is None:
        pkey = (B, N, world, tuple(local_patches.shape[2:]), hip.precision(), default_units())
        cached = net.__dict__.get("_shard_plan")
        if cached is None or cached[0] != pkey:
            cached = net.__dict__["_shard_plan"] = (pkey, shard_plan(net, B, N, world, tuple(local_patches.shape[2:])))
        plan = cached[1]
    if (plan.N, plan.M, plan.I, plan.world) != (N, M, I, world):
        raise ValueError("ips_sharded: the plan is for (N, M, I, world) = %r, the call for %r"
                         % ((plan.N, plan.M, plan.I, plan.world), (N, M, I, world)))
    its, edges, piece = plan.its, plan.edges, plan.piece
    spans = plan.spans(rank)
    n_local = sum(hi - lo for lo, hi in spans)
    assert local_patches.shape[1] == n_local, "rank %d expects %d patches, got %d" % (rank, n_local, local_patches.shape[1])
    if not isinstance(group, LoopbackGroup) and world > 1 and net.__dict__.get("_shard_plan_checked") != (plan.signature(), world):
        # every rank must have cut the axis the same way (the plan depends on the device's unit count): once per shape
        sig = torch.tensor([plan.signature(), -plan.signature()], dtype=torch.int64,
                           device=local_patches.device if dist.get_backend(group) == "nccl" else "cpu")
        dist.all_reduce(sig, op=dist.ReduceOp.MAX, group=group)
        if int(sig[0]) != plan.signature() or int(sig[1]) != -plan.signature():
            raise RuntimeError("ips_sharded: the ranks disagree on the partition (different GPUs per rank?) - pass the same plan= everywhere")
        net.__dict__["_shard_plan_checked"] = (plan.signature(), world)
    dev = local_patches.device
    was_training = net.training
    if was_training:
        net.encoder.eval(); net.transf.eval()
    try:
        ca = net.transf.crs_attn
        on_gpu = hip.on_device(dev)
        if on_gpu:
            R = ca.H * ca.n_token
            vq = ca.folded_query()
            side, main = hip.side_stream(dev), torch.cuda.current_stream(dev)
            # Per-call device buffers are KEPT between calls of the same shape (like IPSNet._scan_bufs): a block the side
            # stream has used cannot be recycled by the allocator until that stream's work is known to be over, so
            # allocating them afresh in every call piles up one set per un-synchronised call and sends the host into
            # hipMalloc now and then - at 8 GPUs a call is a few milliseconds.  The exchange buffers of every part too.
            bkey = (B, N, M, I, R, world, str(dev), tuple(piece))
            if getattr(net, "_shard_bufs_key", None) != bkey:
                net._shard_bufs = {
                    "logits": torch.empty((B, N, R), dtype=torch.float32, device=dev),
                    "mem_idx": torch.empty((B, M), dtype=torch.int64, device=dev),
                    "tie": torch.zeros((B,), dtype=torch.int32, device=dev),
                    "ws": hip.scan_workspace(B, M, I, ca.H, ca.n_token, dev),
                    "mine": [torch.zeros((B, q, R), dtype=torch.float32, device=dev) for q in piece],
                    "gathered": [torch.empty((world, B, q, R), dtype=torch.float32, device=dev) for q in piece],
                }
                net._shard_bufs_key = bkey
                for t in [net._shard_bufs[k] for k in ("logits", "mem_idx", "tie", "ws")] + \
                        net._shard_bufs["mine"] + net._shard_bufs["gathered"]:
                    if t is not None:
                        t.record_stream(side)
            bufs = net._shard_bufs
            logits, mem_idx_buf, tie, scan_ws = bufs["logits"], bufs["mem_idx"], bufs["tie"], bufs["ws"]
            tie.zero_()
            side.wait_stream(main)            # the buffers are the main stream's; the previous call's readers are done
        else:
            all_emb = torch.empty((B, N, D), dtype=torch.float32, device=dev)

        indexed = False
        if on_gpu and net.is_image and not hip.dedup_blank() and local_patches.is_contiguous():
            if net._plan is None:
                net._plan = hip.EncoderPlan(net.encoder, net.is_image)
            indexed = net._plan.fused(local_patches.shape)                 # encode a column range without copying it
        if indexed:
            flat = local_patches.reshape(B * n_local, *local_patches.shape[2:])
            ikey = (plan.signature(), rank, str(dev))
            if getattr(net, "_shard_index_key", None) != ikey:         # int32 patch indices of every part, cached per plan
                rows = torch.arange(B, device=dev, dtype=torch.int32).unsqueeze(1) * n_local
                net._shard_index, b0 = [], 0
                for lo_, hi_ in spans:
                    net._shard_index.append((rows + torch.arange(b0, b0 + hi_ - lo_, device=dev, dtype=torch.int32)).reshape(-1))
                    b0 += hi_ - lo_
                net._shard_index_key = ikey
        ev = None
        if timings is not None and on_gpu:
            mk = lambda: torch.cuda.Event(enable_timing=True)
            ev = {"start": mk(), "enc": [], "xch": [], "scan": [], "end_select": mk(), "end": mk()}
            ev["start"].record(main)
        base = 0
        for k, (lo, hi) in enumerate(spans):
            n_k, q, part_len = hi - lo, piece[k], edges[k + 1] - edges[k]
            width = R if on_gpu else D
            # (padding rows beyond n_k stay zero from construction: only [:, :n_k] is ever written)
            mine = bufs["mine"][k] if on_gpu else torch.zeros((B, q, width), dtype=torch.float32, device=dev)
            if n_k > 0:
                if indexed:
                    emb = net._plan.encode_indexed(flat, net._shard_index[k]).view(B, n_k, D)
                else:
                    part = local_patches[:, base:base + n_k]
                    emb = net._embed(part.reshape(-1, *local_patches.shape[2:])).view(B, n_k, D)
                if on_gpu:
                    pos = net.pos_enc[:, lo:hi] if net.use_pos else None
                    hip.logits(emb, pos, vq, R, out=mine[:, :n_k])
                else:
                    mine[:, :n_k] = emb
            base += n_k
            if on_gpu:
                # the exchange of this part and its loop iterations run on the side stream: the main stream goes
                # straight on to encoding the next part (neither the gather nor the scan is on its critical path)
                done = torch.cuda.Event(enable_timing=ev is not None)
                done.record(main)
                with torch.cuda.stream(side):
                    side.wait_event(done)
                    if ev is not None:
                        ev["enc"].append(done)
                        x0, x1, s1 = mk(), mk(), mk()
                        x0.record(side)
                    gathered = _all_gather(mine, world, group, out=bufs["gathered"][k], part=(edges[k], edges[k + 1]))
                    logits[:, edges[k]:edges[k + 1]] = gathered.permute(1, 0, 2, 3).reshape(B, world * q, width)[:, :part_len]
                    if ev is not None:
                        x1.record(side)
                    hip.scan_range(logits, M, I, ca.H, ca.n_token, its[k], its[k + 1], mem_idx_buf, tie, scan_ws)
                    if ev is not None:
                        s1.record(side)
                        ev["xch"].append((x0, x1))
                        ev["scan"].append((x1, s1))
            else:
                gathered = _all_gather(mine, world, group)                  # the exchange of this part
                all_emb[:, edges[k]:edges[k + 1]] = gathered.permute(1, 0, 2, 3).reshape(B, world * q, width)[:, :part_len]
        if on_gpu:
            main.wait_stream(side)
            mem_idx = mem_idx_buf.clone()              # the buffer is overwritten by the next call
            hip.scan.last_tie = tie
            if ev is not None:
                ev["end_select"].record(main)
        else:
            mem_idx = _scan_aten(net, all_emb)

        # assemble the winners: every rank contributes the rows it owns, zeros elsewhere
        key = (plan.signature(), str(dev))
        if getattr(net, "_owner_key", None) != key:
            net._owner_maps, net._owner_key = plan.owner_maps(dev), key
        owner, lpos = net._owner_maps
        owned = owner[mem_idx] == rank
        local_idx = torch.where(owned, lpos[mem_idx], torch.zeros_like(mem_idx)).clamp_(0, max(n_local - 1, 0))
        if n_local > 0:
            mem_patch = _take(local_patches, local_idx)
            mem_patch = mem_patch * owned.view(B, M, *(1,) * (mem_patch.dim() - 2)).to(mem_patch.dtype)
        else:
            mem_patch = torch.zeros((B, M) + tuple(local_patches.shape[2:]), dtype=local_patches.dtype, device=dev)
        if isinstance(group, LoopbackGroup):
            # the all-reduce's stand-in: the rows other ranks own arrive by a device copy of the same size
            others = _take(group.full_patches, mem_idx)
            mem_patch = torch.where(owned.view(B, M, *(1,) * (mem_patch.dim() - 2)), mem_patch, others)
        else:
            mem_patch = _all_reduce(mem_patch, group)
        mem_pos = _take(net.pos_enc, mem_idx) if net.use_pos else None
        if on_gpu and ev is not None:
            ev["end"].record(main)
            timings.append(ev)
    finally:
        if was_training:
            net.encoder.train(); net.transf.train()
    net.last_mem_idx = mem_idx
    return mem_patch, mem_pos, mem_idx


def phase_ms(timings):
    """Mean milliseconds per call of the phases of ``ips_sharded`` from the events it recorded (synchronise first):
    ``encode_ms`` (main stream: encoder + logits of all parts), ``exchange_ms`` (side stream: the all-gathers incl.
    waiting for the slowest rank), ``scan_ms`` (side stream: all loop iterations), ``exposed_scan_ms`` (what the main
    stream still waited for after its last part was encoded: exchange + loop of the last part - the serial tail),
    ``winners_ms`` (gather + all-reduce of the M winning patches), ``total_ms``."""
    if not timings:
        return None
    acc = {"encode_ms": 0.0, "exchange_ms": 0.0, "scan_ms": 0.0, "exposed_scan_ms": 0.0, "winners_ms": 0.0, "total_ms": 0.0}
    for ev in timings:
        acc["encode_ms"] += ev["start"].elapsed_time(ev["enc"][-1])
        acc["exchange_ms"] += sum(a.elapsed_time(b) for a, b in ev["xch"])
        acc["scan_ms"] += sum(a.elapsed_time(b) for a, b in ev["scan"])
        acc["exposed_scan_ms"] += ev["enc"][-1].elapsed_time(ev["end_select"])
        acc["winners_ms"] += ev["end_select"].elapsed_time(ev["end"])
        acc["total_ms"] += ev["start"].elapsed_time(ev["end"])
    return {k: v / len(timings) for k, v in acc.items()}


def _all_gather(mine, world, group, out=None, part=None):
    """(world, *mine.shape) from every rank's `mine` (into ``out`` when given).  RCCL moves device buffers directly; any
    other backend (gloo: the CPU tests, and GPU ranks without RCCL between them) goes through host memory.  A
    ``LoopbackGroup`` copies the other ranks' pieces of ``part`` = (first row, end row) out of its precomputed logits."""
    if isinstance(group, LoopbackGroup):
        # what the collective delivers - every rank's piece, (world, B, q, R) - arrives by ONE device copy of that size
        # out of a staging tensor made once per part (the other ranks' logits, computed beforehand); this rank's own
        # piece goes on top
        lo, hi = part
        q = mine.shape[1]
        key = (lo, hi, q, world)
        staged = group.__dict__.setdefault("_staged", {}).get(key)
        if staged is None:
            staged = torch.zeros((world,) + tuple(mine.shape), dtype=mine.dtype, device=mine.device)
            for r in range(world):
                a = min(lo + r * q, hi)
                b = min(a + q, hi)
                if b > a:
                    staged[r, :, :b - a].copy_(group.full_logits[:, a:b])
            group._staged[key] = staged
        gathered = out if out is not None else torch.empty_like(staged)
        gathered.copy_(staged)
        gathered[group.rank].copy_(mine)
        return gathered
    if dist.get_backend(group) == "nccl":
        gathered = out if out is not None else torch.empty((world,) + tuple(mine.shape), dtype=mine.dtype, device=mine.device)
        dist.all_gather_into_tensor(gathered, mine, group=group)
        return gathered
    host = mine.cpu()
    pieces = [torch.empty_like(host) for _ in range(world)]
    dist.all_gather(pieces, host, group=group)
    stacked = torch.stack(pieces, 0)
    if out is not None:
        out.copy_(stacked)
        return out
    return stacked.to(mine.device)


def _all_reduce(t, group):
    if dist.get_backend(group) == "nccl" or not t.is_cuda:
        dist.all_reduce(t, group=group)
        return t
    host = t.cpu()
    dist.all_reduce(host, group=group)
    return host.to(t.device)


def slab_span(N, rank, world):
    """[lo, hi) of the contiguous slab a rank holds in the tournament scheme."""
    per = -(-N // world)
    return min(rank * per, N), min((rank + 1) * per, N)


@torch.no_grad()
def ips_tournament(net, local_patches, N, group=None):
    """The north star's literal multi-GPU scheme, opt-in: rank r runs the WHOLE selection loop
    on its own contiguous slab ``patches[:, slab_span(N, r, world)]`` (local memory of M), ONE all-gather moves the M
    winning embeddings and their global indices of every rank ((B, M, D) floats + (B, M) int64: 33 KB per image at the
    MNIST sizes), and every rank runs one final top-M step over the world*M candidates (rank order).

    This is a tournament, NOT the reference's selection: per-head softmax denominators depend on the candidate set, so
    slab winners are not the global winners in general.  It is checked against its own CPU restatement
    (``oracle.Oracle.tournament``); the result-identical scheme - and the one bench.py measures - is ``ips_sharded``.
    Returns ``(mem_patch, mem_pos, mem_idx)`` identical on every rank."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    M, D = net.M, net.D
    lo, hi = slab_span(N, rank, world)
    B, n_local = local_patches.shape[:2]
    assert n_local == hi - lo, "rank %d expects %d patches, got %d" % (rank, hi - lo, n_local)
    dev = local_patches.device
    was_training = net.training
    if was_training:
        net.encoder.eval(); net.transf.eval()
    try:
        pos = net.pos_enc[:, lo:hi].expand(B, -1, -1) if net.use_pos else None
        k = min(M, n_local)                                      # a slab no larger than the memory keeps everything
        if n_local > M:
            net._emb_parts = net._mem_emb = None
            loc = net._select_hip(local_patches, pos) if hip.on_device(dev) else net._select_aten(local_patches, pos)
            net.last_mem_idx = loc
            emb = net.last_mem_emb                               # (B, M, D) embeddings of the slab's winners
        else:
            loc = torch.arange(n_local, dtype=torch.int64, device=dev).unsqueeze(0).expand(B, -1)
            emb = net._embed(local_patches.reshape(-1, *local_patches.shape[2:])).view(B, n_local, D)
        mine_emb = torch.zeros((B, M, D), dtype=torch.float32, device=dev)
        mine_idx = torch.full((B, M), -1, dtype=torch.int64, device=dev)
        mine_emb[:, :k] = emb
        mine_idx[:, :k] = loc + lo
        all_emb = _all_gather(mine_emb, world, group).permute(1, 0, 2, 3).reshape(B, world * M, D)
        all_idx = _all_gather(mine_idx, world, group).permute(1, 0, 2).reshape(B, world * M)
        valid = all_idx[0] >= 0                                   # slab sizes are the same for every image
        cand_emb, cand_idx = all_emb[:, valid].contiguous(), all_idx[:, valid].contiguous()
        cand_pos = None
        if net.use_pos:
            cand_pos = cand_emb + torch.gather(net.pos_enc.expand(B, -1, -1), 1, cand_idx.unsqueeze(-1).expand(-1, -1, D))
        _, mem_idx = net.score_and_select(cand_emb, cand_pos, M, cand_idx)

        owned = (mem_idx >= lo) & (mem_idx < hi)
        local_idx = torch.where(owned, mem_idx - lo, torch.zeros_like(mem_idx)).clamp_(0, max(n_local - 1, 0))
        mem_patch = _take(local_patches, local_idx)
        mem_patch = mem_patch * owned.view(B, M, *(1,) * (mem_patch.dim() - 2)).to(mem_patch.dtype)
        mem_patch = _all_reduce(mem_patch, group)
        mem_pos = _take(net.pos_enc, mem_idx) if net.use_pos else None
    finally:
        if was_training:
            net.encoder.train(); net.transf.train()
    net.last_mem_idx = mem_idx
    net._emb_parts = net._mem_emb = None
    return mem_patch, mem_pos, mem_idx


def _take(src, idx):
    if hip.on_device(src):
        return hip.gather_rows(src, idx)
    view = idx.view(*idx.shape, *(1,) * (src.dim() - 2)).expand(-1, -1, *src.shape[2:])
    return torch.gather(src.expand(idx.shape[0], *src.shape[1:]), 1, view)


def _scan_aten(net, emb):
    """The reference chunk loop (ips_net.py:213-241) on already-computed embeddings."""
    B, N, D = emb.shape
    M, I = net.M, net.I
    order = torch.arange(N, dtype=torch.int64, device=emb.device).unsqueeze(0).expand(B, -1)
    pos = net.pos_enc.expand(B, -1, -1) if net.use_pos else None
    mem_emb, mem_idx = emb[:, :M], order[:, :M]
    for lo in range(M, N, I):
        hi = min(lo + I, N)
        ce = torch.cat((mem_emb, emb[:, lo:hi]), 1)
        ci = torch.cat((mem_idx, order[:, lo:hi]), 1)
        cp = ce + torch.gather(pos, 1, ci.unsqueeze(-1).expand(-1, -1, D)) if net.use_pos else None
        mem_emb, mem_idx = net.score_and_select(ce, cp, M, ci)
    return mem_idx
