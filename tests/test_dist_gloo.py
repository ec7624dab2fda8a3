"""The N>1 path (ips_amd/dist.py) with world_size 2 on CPU (gloo): the sharded selection
must equal the single-process one - slabs, padding, the gather and the owner all-reduce."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ips_amd import dist as ipsd
from tests.util import Golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, case, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = Golden(case)
        net = g.net("cpu")
        x = g.patches()
        N = x.shape[1]
        mine = ipsd.local_indices(N, net.M, net.I, rank, world)
        mem_patch, mem_pos, mem_idx = ipsd.ips_sharded(net, x[:, mine].contiguous(), N)
        ok = np.array_equal(mem_idx.numpy(), g.mem_idx)
        full_patch, full_pos = net.ips(x)
        ok = ok and torch.equal(mem_patch, full_patch)
        ok = ok and (mem_pos is None or torch.equal(mem_pos, full_pos))
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def _tournament_worker(rank, world, port, case, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.oracle import Oracle
        g = Golden(case)
        net = g.net("cpu")
        x = g.patches()
        N = x.shape[1]
        lo, hi = ipsd.slab_span(N, rank, world)
        mem_patch, mem_pos, mem_idx = ipsd.ips_tournament(net, x[:, lo:hi].contiguous(), N)
        want = Oracle(net).tournament(x.numpy(), net.pos_enc.numpy() if g.conf.use_pos else None, world)
        ok = np.array_equal(mem_idx.numpy(), want)
        ok = ok and torch.equal(mem_patch, torch.stack([x[b][mem_idx[b]] for b in range(x.shape[0])]))
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case,world", [("mnist_ragged", 2), ("cam_b2", 3)])
def test_tournament_mode_equals_its_restatement(case, world):
    """SURVEY 8 e-3, the north star's literal scheme (opt-in): every rank selects on its own slab, one all-gather of
    the M winners' embeddings, one final top-M step - checked against oracle.Oracle.tournament (it is NOT the
    reference's selection, which ips_sharded reproduces)."""
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_tournament_worker, args=(world, _free_port(), case, out), nprocs=world, join=True)
        assert dict(out) == {r: True for r in range(world)}


@pytest.mark.parametrize("case", ["mnist_ragged", "cam_b2"])   # N = 301 (odd, padded slab) and features
def test_sharded_ips_equals_single_process(case):
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), case, out), nprocs=world, join=True)
        assert dict(out) == {0: True, 1: True}


def test_partition_covers_every_patch_once_and_respects_chunk_boundaries():
    for N, M, I in ((301, 16, 24), (2500, 64, 64), (20000, 64, 64), (40, 16, 64), (1000, 32, 48), (65536, 256, 256)):
        for world in (1, 2, 3, 4, 8):
            its, edges, piece = ipsd.partition(N, M, I, world)
            assert its[0] == 0 and edges[0] == 0 and edges[-1] == N
            for k in range(1, len(edges) - 1):
                assert (edges[k] - M) % I == 0 and edges[k] == M + its[k] * I      # parts end where a chunk ends
            got = torch.cat([ipsd.local_indices(N, M, I, r, world) for r in range(world)])
            assert sorted(got.tolist()) == list(range(N))
            # every part's pieces are in rank order: the gathered (rank, piece) layout IS the patch order
            for k in range(len(piece)):
                los = [ipsd.local_spans(N, M, I, r, world)[k] for r in range(world)]
                assert los[0][0] == edges[k] and all(a[1] == b[0] or b[0] == b[1] for a, b in zip(los, los[1:]))
