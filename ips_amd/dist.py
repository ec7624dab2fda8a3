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
# Cumulative share of the loop iterations per part.  The encoder works through the parts in order while the scan of
# the previous part runs beside it; what stays exposed is the scan of the LAST part, so the parts shrink towards the
# end (the encoder's workgroup rounds add up the same way whatever the cut).
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


def partition(N, M, I, world, parts=PARTS):
    """Cut [0, N) into parts at chunk boundaries and every part into `world` pieces.

    Returns ``(its, edges, piece)``: ``its[k]`` = first loop iteration of part k (``its[-1]`` = n_iter),
    ``edges[k]`` = first patch of part k, ``piece[k]`` = padded piece length of part k; rank r owns
    patches ``[edges[k] + r*piece[k], min(edges[k] + (r+1)*piece[k], edges[k+1]))`` of every part k.
    """
    n_iter = math.ceil((N - M) / I)
    its = part_iterations(n_iter, parts)
    P = len(its) - 1
    edges = [0] + [min(N, M + it * I) for it in its[1:]]
    edges[-1] = N
    piece = [max(1, math.ceil((edges[k + 1] - edges[k]) / world)) for k in range(P)]
    return its, edges, piece


def local_spans(N, M, I, rank, world, parts=PARTS):
    """Global [lo, hi) ranges this rank owns, in local order (one per part, possibly empty)."""
    _, edges, piece = partition(N, M, I, world, parts)
    out = []
    for k, q in enumerate(piece):
        lo = min(edges[k] + rank * q, edges[k + 1])
        out.append((lo, min(lo + q, edges[k + 1])))
    return out


def local_indices(N, M, I, rank, world, parts=PARTS):
    """1-D int64 tensor of the global patch indices a rank holds, in local order."""
    spans = local_spans(N, M, I, rank, world, parts)
    return torch.cat([torch.arange(lo, hi, dtype=torch.int64) for lo, hi in spans]) if spans else torch.empty(0, dtype=torch.int64)


def _owner_maps(N, M, I, world, device, parts=PARTS):
    """owner[j] = rank holding global patch j, lpos[j] = its position in that rank's local tensor."""
    owner = torch.empty(N, dtype=torch.int64)
    lpos = torch.empty(N, dtype=torch.int64)
    for r in range(world):
        base = 0
        for lo, hi in local_spans(N, M, I, r, world, parts):
            owner[lo:hi] = r
            lpos[lo:hi] = torch.arange(base, base + hi - lo)
            base += hi - lo
    return owner.to(device), lpos.to(device)


@torch.no_grad()
def ips_sharded(net, local_patches, N, group=None, timings=None):
    """IPS over ``N`` patches of which this rank holds ``local_patches`` = patches[:, local_indices(...)].

    ``net`` is an ``IPSNet`` whose ``conf.N`` (positional table) is the GLOBAL ``N``.
    Shuffling is the caller's business here (shard after shuffling).  Returns
    ``(mem_patch, mem_pos, mem_idx)`` identical on every rank.

    ``timings`` (GPU path only): a list that receives one dict of HIP events per call - see ``phase_ms`` - so that a
    run can say where a rank's time went (encoder, exchange, loop, what of the loop stayed exposed).
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    M, I, D = net.M, net.I, net.D
    if M >= N:
        raise ValueError("sharded IPS needs N > M")
    its, edges, piece = partition(N, M, I, world)
    spans = local_spans(N, M, I, rank, world)
    n_local = sum(hi - lo for lo, hi in spans)
    B = local_patches.shape[0]
    assert local_patches.shape[1] == n_local, "rank %d expects %d patches, got %d" % (rank, n_local, local_patches.shape[1])
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
            rows = torch.arange(B, device=dev, dtype=torch.int32).unsqueeze(1) * n_local
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
                    cols = torch.arange(base, base + n_k, device=dev, dtype=torch.int32)
                    emb = net._plan.encode_indexed(flat, (rows + cols).reshape(-1)).view(B, n_k, D)
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
                    gathered = _all_gather(mine, world, group, out=bufs["gathered"][k])
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
        key = (N, M, I, world, str(dev))
        if getattr(net, "_owner_key", None) != key:
            net._owner_maps, net._owner_key = _owner_maps(N, M, I, world, dev), key
        owner, lpos = net._owner_maps
        owned = owner[mem_idx] == rank
        local_idx = torch.where(owned, lpos[mem_idx], torch.zeros_like(mem_idx)).clamp_(0, max(n_local - 1, 0))
        if n_local > 0:
            mem_patch = _take(local_patches, local_idx)
            mem_patch = mem_patch * owned.view(B, M, *(1,) * (mem_patch.dim() - 2)).to(mem_patch.dtype)
        else:
            mem_patch = torch.zeros((B, M) + tuple(local_patches.shape[2:]), dtype=local_patches.dtype, device=dev)
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


def _all_gather(mine, world, group, out=None):
    """(world, *mine.shape) from every rank's `mine` (into ``out`` when given).  RCCL moves device buffers directly; any
    other backend (gloo: the CPU tests, and GPU ranks without RCCL between them) goes through host memory."""
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
