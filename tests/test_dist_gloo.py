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
        mine = ipsd.shard_plan(net, x.shape[0], N, world, tuple(x.shape[2:])).indices(rank)
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


def _models():
    fused = ipsd.LaunchModel(round=2048, t_round=551.0, t_iter=6.6, kind="fused32:fp32")
    fast = ipsd.LaunchModel(round=2048, t_round=79.0, t_iter=6.6, kind="fused32:bf16")
    rows = ipsd.LaunchModel(t_row=0.018, t_launch=15.0, t_iter=4.2, kind="rows")
    return (None, fused, fast, rows)


def test_partition_covers_every_patch_once_and_respects_chunk_boundaries():
    for N, M, I in ((301, 16, 24), (2500, 64, 64), (20000, 64, 64), (40, 16, 64), (1000, 32, 48), (65536, 256, 256), (65, 64, 64)):
        for world in (1, 2, 3, 4, 8):
            for B in (1, 3, 16):
                for model in _models():
                    plan = ipsd.ShardPlan(N, M, I, world, B, model)
                    its, edges, piece = plan.its, plan.edges, plan.piece
                    n_iter = -(-(N - M) // I)
                    assert its[0] == 0 and its[-1] == n_iter and edges[0] == 0 and edges[-1] == N
                    assert all(a < b for a, b in zip(its, its[1:])) and len(its) - 1 <= ipsd.PARTS
                    for k in range(1, len(edges) - 1):
                        assert (edges[k] - M) % I == 0 and edges[k] == M + its[k] * I      # parts end where a chunk ends
                    got = torch.cat([plan.indices(r) for r in range(world)])
                    assert sorted(got.tolist()) == list(range(N))
                    # every part's pieces are in rank order: the gathered (rank, piece) layout IS the patch order
                    for k in range(len(piece)):
                        los = [plan.spans(r)[k] for r in range(world)]
                        assert los[0][0] == edges[k] and all(a[1] == b[0] or b[0] == b[1] for a, b in zip(los, los[1:]))
                    owner, lpos = plan.owner_maps("cpu")
                    for r in range(world):
                        mine = plan.indices(r)
                        assert torch.equal(owner[mine], torch.full_like(mine, r)) and torch.equal(lpos[mine], torch.arange(mine.numel()))
    # the plain functions are the fixed-share partition
    assert ipsd.local_indices(301, 16, 24, 1, 2).tolist() == ipsd.ShardPlan(301, 16, 24, 2).indices(1).tolist()


def test_partition_gives_every_rank_whole_rounds_of_the_fused_trunk():
    """BASELINE configs[1] at the reference's batch (16 x 2,500 patches of 32 px) on 2 / 4 / 8 ranks of
    256 units.  The fixed 50 / 30 / 15 / 5 % cut gave rank 0 launches of 2,688 / 1,408 / 768 / 144 patches at 8 ranks: 5
    rounds of the fused trunk (8 patches per unit) for 2.45 rounds of work.  The launch-aware plan: every launch but the last
    is a whole number of rounds, the last one's remainder is what the pair kernel / a half round takes, and the rounds paid
    stay within 1.15 of one ideal launch - also by the coarse count (whole rounds, a remainder up to a quarter / half / three
    quarters of a round as 0.25 / 0.5 / 0.75)."""
    import math
    from ips_amd import synth
    from ips_amd.architecture import IPSNet
    for name in ("mnist", "mnist3000"):
        conf, B = synth.bench_workload(name)
        net = IPSNet(torch.device("cpu"), conf)
        for world in (1, 2, 4, 8):
            plan = ipsd.shard_plan(net, B, conf.N, world, (1, 32, 32), units=256, precision="fp32")
            assert plan.model.round == 2048
            n = plan.launches(0)
            paid, ideal = plan.rounds(0)
            coarse = sum(v // 2048 + math.ceil((v % 2048) / 512) / 4 for v in n)
            assert all(v % 2048 == 0 for v in n[:-1]), (name, world, n)
            assert paid <= 1.15 * ideal and coarse <= 1.15 * ideal, (name, world, n, paid, coarse, ideal)
            # the fixed shares on the same model, for the record (and so that the plan is never worse than them)
            old = ipsd.ShardPlan(conf.N, conf.M, conf.I, world, B)
            old.model = plan.model
            assert plan.cost_us() <= old.cost_us() + 1e-6
            if name == "mnist" and world == 8:
                assert sum(math.ceil(v / 2048) for v in old.launches(0)) == 5 and old.rounds(0)[0] > 1.25 * ideal
            # every rank's launches, not only rank 0's
            for r in range(1, world):
                pr, ir = plan.rounds(r)
                assert pr <= 1.15 * max(ir, ideal)
    # a rank with less than a round of work is not cut into four
    conf, _ = synth.bench_workload("mnist")
    net = IPSNet(torch.device("cpu"), conf)
    assert len(ipsd.shard_plan(net, 1, conf.N, 8, (1, 32, 32), units=256, precision="fp32").its) <= 3
    # the plan is a pure function of its arguments (every rank builds the same one)
    a = ipsd.shard_plan(net, 16, conf.N, 4, (1, 32, 32), units=256, precision="fp32")
    b = ipsd.shard_plan(IPSNet(torch.device("cpu"), conf), 16, conf.N, 4, (1, 32, 32), units=256, precision="fp32")
    assert a.signature() == b.signature()
    assert a.signature() != ipsd.shard_plan(net, 16, conf.N, 4, (1, 32, 32), units=304, precision="fp32").signature()


def test_bf16_trunk_parts_are_whole_rounds_of_its_eight_patch_workgroups():
    """The bf16 trunk's third build takes EIGHT patches per workgroup (two workgroups per unit: a round is 16 patches per
    unit).  The single-GPU bf16 path cuts its parts with the sharded path's planner (world = 1): every launch but the last a
    whole number of those rounds - the fixed 50 / 30 / 15 / 5 % cut of the headline batch is 12 rounds' worth of launches
    for 9.8 rounds of work."""
    import math
    from ips_amd import synth
    from ips_amd.architecture import IPSNet
    conf, B = synth.bench_workload("mnist")
    net = IPSNet(torch.device("cpu"), conf)
    model = ipsd.launch_model(net, (1, 32, 32), units=256, precision="bf16")
    assert model.round == 16 * 256
    plan = ipsd.ShardPlan(conf.N, conf.M, conf.I, 1, B, model)
    n = plan.launches(0)
    assert sum(n) == B * conf.N and all(v % model.round == 0 for v in n[:-1])
    fixed = ipsd.ShardPlan(conf.N, conf.M, conf.I, 1, B)
    assert sum(math.ceil(v / model.round) for v in fixed.launches(0)) == 12 > sum(math.ceil(v / model.round) for v in n)
    assert ipsd.launch_model(net, (1, 32, 32), units=256, precision="fp32").round == 8 * 256

