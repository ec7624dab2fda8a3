"""Patch-sharded IPS over the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference is single-device (/root/reference/main.py:19-20); this is the one
multi-GPU mechanism the hot path admits without changing its result (SURVEY.md
section 8 e-2): in eval mode the encoder and a patch's attention logits are pure
per-patch functions, so rank r encodes and scores the contiguous slab
``[r*n_loc, (r+1)*n_loc)`` of the patch axis, ONE ``all_gather`` moves the logits
``(B, n_loc, H*T)`` (128 bytes per patch at the MNIST configuration - latency-bound
on xGMI), and every rank replays the identical scan, so all ranks hold the same
``mem_idx`` as a single-GPU run, bit for bit.  The M winning patches are then
assembled with one small ``all_reduce`` of zero-filled owner contributions (exact:
x + 0 = x).

On CPU tensors (gloo; the world_size-2 tests) the same partitioning runs with the
ATen path: embeddings are all-gathered instead of logits and the reference loop
runs on them.
"""

import torch
import torch.distributed as dist

from . import hip


def slab(N, rank, world):
    """Contiguous slab of the patch axis owned by ``rank``: (lo, hi, n_loc); n_loc is padded."""
    n_loc = (N + world - 1) // world
    lo = min(rank * n_loc, N)
    return lo, min(lo + n_loc, N), n_loc


@torch.no_grad()
def ips_sharded(net, local_patches, N, group=None):
    """IPS over ``N`` patches of which this rank holds ``local_patches`` (B, hi-lo, ...).

    ``net`` is an ``IPSNet`` whose ``conf.N`` (positional table) is the GLOBAL ``N``.
    Shuffling is the caller's business here (shard after shuffling).  Returns
    ``(mem_patch, mem_pos, mem_idx)`` identical on every rank.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi, n_loc = slab(N, rank, world)
    B = local_patches.shape[0]
    assert local_patches.shape[1] == hi - lo, "rank %d expects %d patches, got %d" % (rank, hi - lo, local_patches.shape[1])
    dev = local_patches.device
    M, D = net.M, net.D
    if M >= N:
        raise ValueError("sharded IPS needs N > M")
    was_training = net.training
    if was_training:
        net.encoder.eval(); net.transf.eval()
    try:
        ca = net.transf.crs_attn
        pos = net.pos_enc[:, lo:hi] if net.use_pos else None           # (1, n, D) slab of the table
        emb = net._embed(local_patches.reshape(-1, *local_patches.shape[2:])).view(B, hi - lo, D)

        if hip.on_device(dev):
            R = ca.H * ca.n_token
            mine = torch.zeros((B, n_loc, R), dtype=torch.float32, device=dev)
            if hi > lo:
                hip.logits(emb, pos, hip.pack_linear(ca.k_w.weight), ca.scaled_query(), ca.H, ca.D_k,
                           ca.n_token, out=mine[:, :hi - lo])
            gathered = torch.empty((world, B, n_loc, R), dtype=torch.float32, device=dev)
            dist.all_gather_into_tensor(gathered, mine, group=group)   # the one exchange of the scan
            logits = gathered.permute(1, 0, 2, 3).reshape(B, world * n_loc, R)[:, :N].contiguous()
            mem_idx = hip.scan(logits, M, net.I, ca.H, ca.n_token)
        else:
            mine = torch.zeros((B, n_loc, D), dtype=torch.float32, device=dev)
            mine[:, :hi - lo] = emb
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine, group=group)
            all_emb = torch.stack(parts, 1).reshape(B, world * n_loc, D)[:, :N]
            mem_idx = _scan_aten(net, all_emb)

        # assemble the winners: every rank contributes the rows it owns, zeros elsewhere
        owned = (mem_idx >= lo) & (mem_idx < hi)
        local_idx = (mem_idx - lo).clamp_(0, max(hi - lo - 1, 0))
        mem_patch = _take(local_patches, local_idx)
        mem_patch = mem_patch * owned.view(B, M, *(1,) * (mem_patch.dim() - 2)).to(mem_patch.dtype)
        dist.all_reduce(mem_patch, group=group)
        mem_pos = _take(net.pos_enc, mem_idx) if net.use_pos else None
    finally:
        if was_training:
            net.encoder.train(); net.transf.train()
    net.last_mem_idx = mem_idx
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
