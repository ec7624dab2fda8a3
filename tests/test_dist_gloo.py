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
        lo, hi, _ = ipsd.slab(N, rank, world)
        mem_patch, mem_pos, mem_idx = ipsd.ips_sharded(net, x[:, lo:hi].contiguous(), N)
        ok = np.array_equal(mem_idx.numpy(), g.mem_idx)
        full_patch, full_pos = net.ips(x)
        ok = ok and torch.equal(mem_patch, full_patch)
        ok = ok and (mem_pos is None or torch.equal(mem_pos, full_pos))
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["mnist_ragged", "cam_b2"])   # N = 301 (odd, padded slab) and features
def test_sharded_ips_equals_single_process(case):
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), case, out), nprocs=world, join=True)
        assert dict(out) == {0: True, 1: True}


def test_slab_partition_covers_everything():
    for N in (1, 7, 64, 301, 2500, 10000):
        for world in (1, 2, 3, 4, 8):
            spans = [ipsd.slab(N, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == N
            for (a, b, n), (c, d, _) in zip(spans, spans[1:]):
                assert b == c and b - a <= n
            assert sum(b - a for a, b, _ in spans) == N
