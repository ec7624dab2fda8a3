"""End-to-end parity of IPSNet.ips / IPSNet.forward on the MI355X.

Against (1) the golden fixtures recorded from the reference itself - selected indices
must be IDENTICAL, final outputs within 1e-4 (north_star) - and (2) the CPU oracle on
the same inputs - indices identical, embeddings and outputs bit for bit.
"""

import os

import numpy as np
import pytest
import torch

from ips_amd import hip, synth
from ips_amd.architecture import IPSNet
from oracle import oracle as orc
from tests.util import Golden, GOLDEN_CASES, ulp_diff

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
PRED_TOL = 1e-4      # tolerance north_star states for the final fp32 outputs


@pytest.mark.parametrize("case", [c for c in GOLDEN_CASES if "shuffle_instance" not in c])
def test_ips_and_forward_match_reference_fixture(case):
    g = Golden(case)
    net = g.net(DEV)
    x = g.patches().to(DEV)                              # eager loading: patches resident on the GPU
    torch.manual_seed(g.torch_seed)                      # 'batch' shuffle draws randperm on the CPU generator
    mem_patch, mem_pos = net.ips(x)
    assert mem_patch.is_cuda and mem_patch.shape[:2] == (g.B, g.conf.M)
    assert np.array_equal(net.last_mem_idx.cpu().numpy(), g.mem_idx), "indices differ from the reference"
    s = mem_patch.double().sum(dim=tuple(range(2, mem_patch.dim()))).cpu().numpy()
    assert np.allclose(s, g.mem_patch_sum, rtol=1e-12, atol=1e-9)
    if g.mem_pos_sum is not None:
        # the table is built by the HOST's sin/cos (reference transformer.py:6-18), which differs in
        # the last ulp between CPU models; the selection of rows is exact, the values are host-made
        # (exp() for the frequencies differs by an ulp, multiplied by positions up to N: ~1e-4 absolute)
        assert np.allclose(mem_pos.double().sum(-1).cpu().numpy(), g.mem_pos_sum, atol=2e-2, rtol=0)
        if g.perm is None:
            assert torch.equal(mem_pos, torch.stack([net.pos_enc[0][net.last_mem_idx[b]] for b in range(g.B)]))
    with torch.no_grad():
        preds = net(mem_patch, mem_pos)
    for k, v in g.preds.items():
        assert np.abs(preds[k].cpu().numpy() - v).max() < PRED_TOL, k


@pytest.mark.parametrize("case", ["mnist_ties", "mnist_ties_wide"])
def test_tie_fixtures_need_the_torch_order(case):
    """The tie fixtures (identical blank patches, no positional encoding) are reproduced by the default tie order
    (previous test); the canonical order selects other - equally scored - patches and raises the tie flag."""
    g = Golden(case)
    net = g.net(DEV)
    x = g.patches().to(DEV)
    assert g.min_rel_gap == 0.0
    hip.set_tie_order("canonical")
    try:
        net.ips(x)
        canonical = net.last_mem_idx.cpu().numpy()
        assert int(hip.scan.last_tie.sum().item()) > 0
    finally:
        hip.set_tie_order("torch")
    net.ips(x)
    assert np.array_equal(net.last_mem_idx.cpu().numpy(), g.mem_idx)
    assert not np.array_equal(canonical, g.mem_idx)
    o = orc.Oracle(g.net("cpu"))
    assert np.array_equal(o.ips(g.patches().numpy(), None, aten_ties=True)["mem_idx"], g.mem_idx)


@pytest.mark.parametrize("case", ["mnist_mini", "mnist_ragged", "mnist_onechunk", "mnist_tok1", "cam_b2",
                                  "traffic_tiny", "mnist_full"])
def test_ips_and_forward_bit_exact_vs_oracle(case):
    g = Golden(case)
    net = g.net(DEV)
    o = orc.Oracle(g.net("cpu"))
    x = g.patches()
    pos = g.net("cpu").pos_enc.numpy() if g.conf.use_pos else None
    want = o.ips(x.numpy(), pos)
    mem_patch, mem_pos = net.ips(x.to(DEV))
    assert np.array_equal(net.last_mem_idx.cpu().numpy(), want["mem_idx"])
    assert np.array_equal(mem_patch.cpu().numpy(), want["mem_patch"])
    if pos is not None:
        assert np.array_equal(mem_pos.cpu().numpy(), want["mem_pos"])
    with torch.no_grad():
        preds = net(mem_patch, mem_pos)
    wp = o.forward(want["mem_patch"], want["mem_pos"])
    for k in wp:
        assert ulp_diff(preds[k].cpu().numpy(), wp[k]) == 0, k


@pytest.mark.parametrize("seed", range(10))
def test_random_configurations_bit_exact_vs_oracle(seed):
    """Random small configurations: ragged last chunk, I > N - M, the M >= N shortcut, B > 1, 1 / 2 / 4
    query tokens, with and without positional encoding, image and feature inputs, blank patches with ties - ips() and
    the eval forward against the oracle, bit for bit (the oracle replays torch's tie order like the device does)."""
    g = np.random.default_rng(500 + seed)
    feat = bool(seed % 5 == 4)
    T = int(g.choice([1, 2, 4]))
    tasks = {"task%d" % t: {"id": t, "name": "t%d" % t, "act_fn": "softmax" if t % 2 == 0 else "sigmoid",
                            "metric": "accuracy" if t % 2 == 0 else "multilabel_accuracy"} for t in range(T)}
    M = int(g.choice([4, 8, 16, 24]))
    I = int(g.choice([3, 8, 16, 40]))
    N = int(g.choice([M - 1, M, M + 1, M + I - 1, M + 2 * I + 3, 5 * I + M])) if seed % 3 else int(M + g.integers(1, 60))
    N = max(N, 2)
    use_pos = bool(g.integers(0, 2)) and N % 1 == 0
    B = int(g.choice([1, 2, 3]))
    if feat:
        conf = synth.camelyon_conf(N=N, M=M, I=I, n_chan_in=64, D=32, D_k=8, D_v=8, H=4, D_inner=64, n_token=T,
                                   tasks=tasks, use_pos=use_pos, n_class=3)
    else:
        conf = synth.mnist_conf(N=N, M=M, I=I, n_token=T, tasks=tasks, use_pos=use_pos, n_class=3,
                                blank_frac=float(g.choice([0.0, 0.5, 0.93])))
    cpu = synth.fill_weights(IPSNet(torch.device("cpu"), conf), 40 + seed).eval()
    net = synth.fill_weights(IPSNet(torch.device(DEV), conf), 40 + seed).to(DEV).eval()
    x = synth.make_patches(conf, B, seed=70 + seed)
    o = orc.Oracle(cpu)
    mem_patch, mem_pos = net.ips(x.to(DEV))
    if M >= N:                                                      # shortcut: everything is kept, in order
        assert torch.equal(mem_patch.cpu(), x) and net.last_mem_idx is None
        want_patch, want_pos = x.numpy(), (np.broadcast_to(cpu.pos_enc.numpy(), (B, N, conf.D)) if use_pos else None)
    else:
        want = o.ips(x.numpy(), cpu.pos_enc.numpy() if use_pos else None, aten_ties=True)
        assert np.array_equal(net.last_mem_idx.cpu().numpy(), want["mem_idx"]), (N, M, I, T, use_pos, B, feat)
        assert np.array_equal(mem_patch.cpu().numpy(), want["mem_patch"])
        want_patch, want_pos = want["mem_patch"], want["mem_pos"]
    with torch.no_grad():
        preds = net(mem_patch, mem_pos)
    wp = o.forward(want_patch, want_pos)
    for k in wp:
        assert ulp_diff(preds[k].cpu().numpy(), wp[k]) == 0, k


def test_lazy_loading_equals_eager():
    """patches left on the host (reference ips_net.py:204-206,223,245-247)."""
    g = Golden("mnist_ragged")
    net = g.net(DEV)
    x = g.patches()
    mp_lazy, pos_lazy = net.ips(x)                       # CPU tensor in, device tensors out
    idx_lazy = net.last_mem_idx.clone()
    mp_eager, pos_eager = net.ips(x.to(DEV))
    assert mp_lazy.is_cuda and torch.equal(idx_lazy, net.last_mem_idx)
    assert torch.equal(mp_lazy, mp_eager) and torch.equal(pos_lazy, pos_eager)
    assert np.array_equal(idx_lazy.cpu().numpy(), g.mem_idx)


@pytest.mark.parametrize("pinned,keep_mb", [(False, "16384"), (True, "16384"), (True, "0")])
def test_lazy_slab_pipeline(monkeypatch, pinned, keep_mb):
    """Lazy loading through several slabs (copy stream + events), pinned or pageable host memory, with the
    slabs kept on the device (device-side final gather) or recycled (host-side gather, as the reference)."""
    monkeypatch.setenv("IPSX_LAZY_KEEP_MB", keep_mb)
    g = Golden("mnist_ragged")
    net = g.net(DEV)
    from ips_amd.selection import Selection
    monkeypatch.setattr(Selection, "LAZY_SLAB_BYTES", 2 * 70 * 4096)          # 70 patches per slab -> 5 slabs
    x = g.patches()
    if pinned:
        x = x.pin_memory()
    mp, pos = net.ips(x)
    assert mp.is_cuda and np.array_equal(net.last_mem_idx.cpu().numpy(), g.mem_idx)
    assert torch.equal(mp.cpu(), torch.stack([x[b][net.last_mem_idx[b].cpu()] for b in range(g.B)]))


def test_instance_shuffle_against_oracle():
    """shuffle_style='instance' draws on the device generator; check against the oracle fed the same order."""
    g = Golden("mnist_shuffle_instance")
    net = g.net(DEV)
    x = g.patches().to(DEV)
    seen = {}
    orig = net.do_shuffle

    def spy(patches, pos_enc):
        p, pe = orig(patches, pos_enc)
        seen["p"], seen["pe"] = p, pe
        return p, pe

    net.do_shuffle = spy
    torch.manual_seed(3)
    mem_patch, mem_pos = net.ips(x)
    o = orc.Oracle(g.net("cpu"))
    want = o.ips(seen["p"].cpu().numpy(), seen["pe"].cpu().numpy())
    assert np.array_equal(net.last_mem_idx.cpu().numpy(), want["mem_idx"])
    assert np.array_equal(mem_patch.cpu().numpy(), want["mem_patch"])
    assert np.array_equal(mem_pos.cpu().numpy(), want["mem_pos"])


def test_reference_style_chunk_loop_equals_fused_scan():
    """score_and_select per chunk (the reference's structure, through get_scores + topm kernels)
    selects exactly what the single-launch scan selects."""
    g = Golden("mnist_ragged")
    net = g.net(DEV)
    x = g.patches().to(DEV)
    net.ips(x)
    fused = net.last_mem_idx.clone()
    B, N = x.shape[:2]
    D, M, I = g.conf.D, g.conf.M, g.conf.I
    pos = net.pos_enc.expand(B, -1, -1)
    order = torch.arange(N, device=DEV).unsqueeze(0).expand(B, -1)
    with torch.no_grad():
        emb = net._embed(x.reshape(-1, *x.shape[2:])).view(B, N, D)
        mem_emb, mem_idx = emb[:, :M], order[:, :M]
        for lo in range(M, N, I):
            hi = min(lo + I, N)
            ce = torch.cat((mem_emb, emb[:, lo:hi]), 1)
            ci = torch.cat((mem_idx, order[:, lo:hi]), 1)
            cp = ce + torch.gather(pos, 1, ci.unsqueeze(-1).expand(-1, -1, D))
            mem_emb, mem_idx = net.score_and_select(ce, cp, M, ci)
    assert torch.equal(mem_idx, fused)
    assert np.array_equal(fused.cpu().numpy(), g.mem_idx)


def test_full_size_properties_b16():
    """BASELINE configs[1] at the benchmark batch (16 x 2500 patches): size-independent properties."""
    conf = synth.mnist_conf(N=2500, M=64, I=64)
    net = synth.fill_weights(IPSNet(torch.device(DEV), conf), 7).to(DEV).eval()
    x = synth.make_patches(conf, 16, seed=21).to(DEV)
    mp1, pos1 = net.ips(x)
    idx1 = net.last_mem_idx.clone()
    mp2, _ = net.ips(x)
    assert torch.equal(idx1, net.last_mem_idx) and torch.equal(mp1, mp2)          # deterministic
    idx = idx1.cpu().numpy()
    assert idx.min() >= 0 and idx.max() < 2500
    assert all(len(set(r)) == 64 for r in idx)                                       # a patch is kept once
    assert torch.equal(mp1, torch.stack([x[b][idx1[b]] for b in range(16)]))         # gather is exact
    assert torch.equal(pos1, torch.stack([net.pos_enc[0][idx1[b]] for b in range(16)]))
    assert int(hip.scan.last_tie.sum()) == 0
    # batch independence: image 3 alone selects the same patches
    net.ips(x[3:4])
    assert torch.equal(net.last_mem_idx[0], idx1[3])
    # image 0 of this batch is the mnist_full fixture's input/weights -> the reference's indices
    g = Golden("mnist_full")
    netg = g.net(DEV)
    xg = torch.cat([g.patches().to(DEV), x[:3]], 0)
    netg.ips(xg)
    assert np.array_equal(netg.last_mem_idx[0].cpu().numpy(), g.mem_idx[0])


def test_configs2_size_properties(monkeypatch):
    """BASELINE configs[2] size on one GPU (3000 x 3000 image = 10,000 patches of 32 px, 156 iterations): the
    overlapped path equals the plain one and the reference-structured chunk loop, kept patches are unique, and the
    memory comes out ordered by the scores of the last iteration."""
    conf = synth.mnist_conf(N=10000, M=64, I=64)
    net = synth.fill_weights(IPSNet(torch.device(DEV), conf), 11).to(DEV).eval()
    x = synth.make_patches(conf, 2, seed=5).to(DEV)
    assert net.selection.can_overlap(x)
    net.ips(x)
    idx = net.last_mem_idx.clone()
    monkeypatch.setenv("IPSX_OVERLAP_SCAN", "0")
    net.ips(x)
    monkeypatch.delenv("IPSX_OVERLAP_SCAN")
    assert torch.equal(idx, net.last_mem_idx)
    assert all(len(set(r)) == 64 for r in idx.cpu().numpy())
    # the reference's structure: score_and_select per chunk on embeddings (ipsx_scores + ipsx_topm)
    with torch.no_grad():
        B, N, M, I, D = 2, 10000, 64, 64, conf.D
        emb = net._embed(x.reshape(-1, 1, 32, 32)).view(B, N, D)
        pos = net.pos_enc.expand(B, -1, -1)
        order = torch.arange(N, device=DEV).unsqueeze(0).expand(B, -1)
        mem_emb, mem_idx = emb[:, :M], order[:, :M]
        for lo in range(M, N, I):
            hi = min(lo + I, N)
            ce = torch.cat((mem_emb, emb[:, lo:hi]), 1)
            ci = torch.cat((mem_idx, order[:, lo:hi]), 1)
            cp = ce + torch.gather(pos, 1, ci.unsqueeze(-1).expand(-1, -1, D))
            mem_emb, mem_idx = net.score_and_select(ce, cp, M, ci)
            last_scores = net.transf.get_scores(cp)
            last_ci = ci
        assert torch.equal(mem_idx, idx)
        # ordered by the last iteration's scores, and nothing left out scores higher than the last kept one
        for b in range(B):
            sc = {int(i): float(v) for i, v in zip(last_ci[b].tolist(), last_scores[b].tolist())}
            kept = [sc[int(i)] for i in idx[b].tolist()]
            assert all(a >= c for a, c in zip(kept, kept[1:]))
            assert max(v for i, v in sc.items() if i not in set(idx[b].tolist())) <= kept[-1]


def test_sharded_path_on_gpu_single_rank_rccl():
    """ips_amd.dist.ips_sharded over an RCCL (nccl) group of one rank: exercises the GPU branch
    (logits into a padded slab, all_gather_into_tensor, scan, owner all_reduce) and must equal ips()."""
    import os
    import torch.distributed as dist
    from ips_amd import dist as ipsd
    if dist.is_initialized():
        pytest.skip("a process group already exists")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        g = Golden("mnist_ragged")
        net = g.net(DEV)
        x = g.patches().to(DEV)
        mine = ipsd.shard_plan(net, x.shape[0], x.shape[1], 1, tuple(x.shape[2:])).indices(0).to(DEV)
        mp, pos, idx = ipsd.ips_sharded(net, x[:, mine].contiguous(), x.shape[1])
        full_patch, full_pos = net.ips(x)
        assert torch.equal(idx, net.last_mem_idx) and np.array_equal(idx.cpu().numpy(), g.mem_idx)
        assert torch.equal(mp, full_patch) and torch.equal(pos, full_pos)
    finally:
        dist.destroy_process_group()


def test_sharded_path_world_size_2_on_one_gpu():
    """world_size 2 on the GPU path: two child processes share cuda:0 and exchange through gloo/host memory
    (tools/dist_check.py), so the part-wise indexed encode, the resumable scan on the side stream and the owner
    all-reduce run with more than one rank; every rank must reproduce the reference fixtures and its own ips()."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29547", os.path.join(repo, "tools", "dist_check.py"),
           "--backend", "gloo", "--share-gpu", "--cases", "mnist_ragged,mnist_full,cam_b2"]
    out = subprocess.run(cmd, cwd=repo, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count(" ok") == 6 and "MISMATCH" not in out.stdout
    # the opt-in tournament scheme against its CPU restatement, same transport
    cmd = cmd[:cmd.index("--cases")] + ["--tournament", "--cases", "mnist_ragged,mnist_full,cam_b2"]
    cmd[cmd.index("29547")] = "29548"
    out = subprocess.run(cmd, cwd=repo, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count("tournament ok") == 6 and "MISMATCH" not in out.stdout


@pytest.mark.parametrize("world,precision,storage", [(2, "fp32", "f32"), (2, "bf16", "f16"), (4, "bf16", "f16"), (2, "fp32x3", "bf16")])
def test_sharded_path_at_the_headline_shape_and_at_reduced_precision(world, precision, storage):
    """VERDICT r05 items 1 and 3.  (a) The launch-aware partition (ips_amd.dist.shard_plan: whole rounds of the fused trunk
    per rank, three parts at the headline shape) on the GPU path with more than one rank: the sharded selection of the
    headline batch equals the rank's own single-GPU selection and, at fp32, the reference fixture.  (b) BASELINE configs[4]
    is an 8-GPU configuration: ``IPSX_PRECISION=bf16`` with fp16-STORED patches through ``ips_sharded`` - no reference
    behaviour exists at that precision, so the contract is: sharded == single-GPU at the same precision and storage, bit
    for bit (indices, gathered half-precision patches, positional rows).  Ranks share cuda:0 over gloo (tools/dist_check.py)."""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29551 + world + (7 if precision != "fp32" else 0)),
           os.path.join(repo, "tools", "dist_check.py"), "--backend", "gloo", "--share-gpu", "--bench-shape",
           "--precision", precision, "--storage", storage, "--cases", "mnist_ragged,mnist_full"]
    out = subprocess.run(cmd, cwd=repo, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count(" ok") == 3 * world and "MISMATCH" not in out.stdout
    assert ("%s %s" % (precision, storage)) in out.stdout and "bench_mnist" in out.stdout


def test_sharded_path_over_rccl_one_gpu_per_rank():
    """The N > 1 path as it is deployed: one process per GPU, ``torch.distributed`` backend ``nccl`` (= RCCL over xGMI).
    Runs wherever at least two GPUs are visible (the 8-GPU node of the scaling bench) and skips on a 1-GPU box, where
    ``test_sharded_path_world_size_2_on_one_gpu`` covers the same code over gloo.  Every rank must reproduce the
    reference-recorded fixtures and its own single-GPU ``ips()``; then ``bench.py --gpus N`` itself (configs[2],
    patch-sharded) must report the reference's selection on every rank, and say which backend and devices it ran on."""
    import json
    import os
    import subprocess
    import sys
    n_gpu = torch.cuda.device_count()
    if n_gpu < 2:
        pytest.skip("needs >= 2 GPUs (RCCL needs one GPU per rank); this box has %d" % n_gpu)
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    worlds = [2] + ([n_gpu] if n_gpu >= 4 else [])
    for k, world in enumerate(worlds):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(29551 + k), os.path.join(repo, "tools", "dist_check.py"),
               "--backend", "nccl", "--cases", "mnist_ragged,mnist_full,cam_b2"]
        out = subprocess.run(cmd, cwd=repo, env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        assert out.stdout.count(" ok") == 3 * world and "MISMATCH" not in out.stdout
    # both ways a caller may start it: under the launcher, and bare (bench.py then starts its own ranks)
    for cmd in ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                 "--master-port", "29559", os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                [sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"]):
        clean = {k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
        out = subprocess.run(cmd, cwd=repo, env=clean, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        assert line["n_gpus"] == 2 and line["parity_all_ranks"] is True and line["scaling"] == "strong"
        assert line["distributed"]["backend"] == "nccl" and line["distributed"]["world_size"] == 2
        assert len({r["device_index"] for r in line["per_rank"]}) == 2
        assert line["n1_value_same_workload"] > 0 and "BASELINE configs[1]" in line["config"]["workload"]


def test_bench_py_starts_its_own_ranks():
    """``python bench.py --gpus 2`` with no launcher around it (the shape of the driver's single-GPU command): the parent
    starts two child ranks itself and relays rank 0's line.  On this one-GPU box both ranks share cuda:0 and exchange
    through gloo (IPSX_BENCH_SHARE_GPU=1); the line must be the headline workload - the SAME 16 x 2500 patches as at
    N = 1, labelled strong scaling -, carry every rank's parity verdict against the reference fixture, the single-rank
    time of the same workload measured in the same run, and configs[2] sharded the same way."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["IPSX_BENCH_SHARE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"]
    out = subprocess.run(cmd, cwd=repo, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["steps"] == 3
    assert "BASELINE configs[1]" in line["config"]["workload"]
    assert line["parity"]["indices_equal"] is True and line["parity_all_ranks"] is True
    assert [r["rank"] for r in line["per_rank"]] == [0, 1]
    assert all(r["indices_equal"] is True and r["patches_per_image"] == 1250 for r in line["per_rank"])
    assert line["n1_value_same_workload"] > 0 and line["speedup_over_n1_same_run"] > 0
    leg = line["also_measured"]["mnist3000"]
    assert leg["parity"]["indices_equal"] is True and leg["parity_all_ranks"] is True
    # a failing child is the parent's failure: a secondary config is refused at N > 1 with exit code 2
    bad = subprocess.run(cmd + ["--config", "cam"], cwd=repo, env=env, capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0


def test_training_step_between_ips_calls():
    """What training/iterative.py does (reference :135-163): ips() in train mode, forward with autograd on the
    stock ROCm ops, backward, optimizer.step().  Weights and BatchNorm running statistics have moved, so the
    next ips() must re-pack them: its selection has to equal the oracle's on the UPDATED weights."""
    g = Golden("mnist_mini")
    net = g.net(DEV)
    net.train()
    x = g.patches().to(DEV)
    opt = torch.optim.AdamW(net.parameters(), lr=1e-3, weight_decay=0.1)
    mem_patch, mem_pos = net.ips(x)
    assert np.array_equal(net.last_mem_idx.cpu().numpy(), g.mem_idx)          # eval statistics inside ips()
    before = net.encoder[1].running_mean.clone()
    preds = net(mem_patch, mem_pos)                                            # train mode: batch statistics, dropout
    assert all(p.requires_grad for p in preds.values())
    loss = sum((p ** 2).mean() for p in preds.values())
    opt.zero_grad()
    loss.backward()
    assert net.encoder[0].weight.grad is not None and net.transf.crs_attn.q.grad is not None
    opt.step()
    assert not torch.equal(before, net.encoder[1].running_mean)                # BN buffers moved in forward
    net.ips(x)
    after = net.last_mem_idx.cpu().numpy()
    cpu = IPSNet(torch.device("cpu"), g.conf)
    cpu.load_state_dict({k: v.cpu() for k, v in net.state_dict().items()})
    cpu.eval()
    want = orc.Oracle(cpu).ips(g.patches().numpy(), cpu.pos_enc.numpy())
    assert np.array_equal(after, want["mem_idx"])


def test_resnet50_bottleneck_trunk_bit_exact():
    """enc_type='resnet50' (reference ips_net.py:22-25): Bottleneck blocks through the channels-last kernels."""
    conf = synth.traffic_conf(N=12, M=4, I=4, patch=48, enc_type='resnet50', n_res_blocks=2, D=512)
    net = synth.fill_weights(IPSNet(torch.device(DEV), conf), 3).to(DEV).eval()
    x = synth.make_patches(conf, 1, seed=4)
    cpu = synth.fill_weights(IPSNet(torch.device("cpu"), conf), 3).eval()
    o = orc.Oracle(cpu)
    want = o.encode(x[0].numpy())
    with torch.no_grad():
        got = net._embed(x[0].to(DEV)).cpu().numpy()
    assert got.shape == (12, 512)
    assert ulp_diff(got, want) == 0
    net.ips(x.to(DEV))
    assert np.array_equal(net.last_mem_idx.cpu().numpy(), o.ips(x.numpy())["mem_idx"])


def test_bf16_precision_end_to_end(monkeypatch):
    """ips() + forward with the bf16 trunk: finite, close to the fp32 outputs, selection mostly the same."""
    g = Golden("mnist_full")
    net = g.net(DEV)
    x = g.patches().to(DEV)
    mp32, pos32 = net.ips(x)
    idx32 = net.last_mem_idx.clone()
    with torch.no_grad():
        p32 = net(mp32, pos32)
    monkeypatch.setenv("IPSX_PRECISION", "bf16")
    mp16, pos16 = net.ips(x)
    idx16 = net.last_mem_idx.clone()
    with torch.no_grad():
        p16 = net(mp16, pos16)
    monkeypatch.delenv("IPSX_PRECISION")
    common = len(set(idx32[0].tolist()) & set(idx16[0].tolist()))
    assert common >= 0.85 * g.conf.M, common
    for k in p32:
        assert torch.isfinite(p16[k]).all() and float((p16[k] - p32[k]).abs().max()) < 0.1


@pytest.mark.parametrize("case", ["mnist_mini", "mnist_ragged", "mnist_tok1", "mnist_full"])
def test_fp32x3_precision_selects_the_reference_indices(case, monkeypatch):
    """ips() + forward with the split-bf16 trunk: the reference's indices and its outputs within the north-star
    tolerance (the boundary gaps of the fixtures are far above fp32 rounding differences)."""
    g = Golden(case)
    net = g.net(DEV)
    x = g.patches().to(DEV)
    monkeypatch.setenv("IPSX_PRECISION", "fp32x3")
    mem_patch, mem_pos = net.ips(x)
    assert hip.encoder_kernel_name(net._plan) == "fused_trunk_x3_kernel"
    with torch.no_grad():
        preds = net(mem_patch, mem_pos)
    monkeypatch.delenv("IPSX_PRECISION")
    assert np.array_equal(net.last_mem_idx.cpu().numpy(), g.mem_idx)
    for name, want in g.preds.items():
        np.testing.assert_allclose(preds[name].cpu().numpy(), want, rtol=0, atol=1e-4)


def test_scan_overlapped_with_encoder_equals_plain(monkeypatch):
    """Default path: the image is encoded in 4 parts and the selection loop follows on a side stream
    (ipsx_scan_range); IPSX_OVERLAP_SCAN=0 is the plain encode-all-then-scan path.  Same result."""
    g = Golden("mnist_full")
    net = g.net(DEV)
    x = torch.cat([g.patches(), synth.make_patches(g.conf, 13, seed=99)], 0).to(DEV)    # 35,000 patches: overlapped
    assert net.selection.can_overlap(x) and not net.selection.can_overlap(x[:4])
    mp_a, pos_a = net.ips(x)
    idx_a = net.last_mem_idx.clone()
    monkeypatch.setenv("IPSX_OVERLAP_SCAN", "0")
    assert not net.selection.can_overlap(x)
    mp_b, pos_b = net.ips(x)
    monkeypatch.delenv("IPSX_OVERLAP_SCAN")
    assert torch.equal(idx_a, net.last_mem_idx) and torch.equal(mp_a, mp_b) and torch.equal(pos_a, pos_b)
    assert np.array_equal(idx_a[0].cpu().numpy(), g.mem_idx[0])


@pytest.mark.parametrize("N", [2500, 1000, 2501])
def test_one_image_every_schedule_selects_the_same_patches(N, monkeypatch):
    """One image per call (the reference's eager-sequential mode): by default trunk + logits are ONE persistent launch
    beside a resident loop (ipsx_trunk_stream); IPSX_IMAGE_STREAM=0 cuts the image into parts with the loop beside the
    next part's trunk, IPSX_OVERLAP_SCAN=0 is encode-all-then-scan.  Same indices, patches and positions - with and
    without positional encoding, also when the image is the shuffled one."""
    g = Golden("mnist_full")
    for use_pos, shuffle in ((True, False), (False, False), (True, True)):
        conf = g.conf.clone(N=N, use_pos=use_pos, shuffle=shuffle)
        from ips_amd.architecture import IPSNet
        net = synth.fill_weights(IPSNet(torch.device(DEV), conf), 7).to(DEV).eval()
        x = synth.make_patches(conf, 1, seed=5).to(DEV)
        assert net.selection.can_stream_image(x)
        res = []
        for env in ({}, {"IPSX_NATIVE_CALL": "0"}, {"IPSX_IMAGE_STREAM": "0"}, {"IPSX_OVERLAP_SCAN": "0"}):      # (default: one library call)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            torch.manual_seed(3)                           # (shuffle draws from torch's generator)
            mp, pos = net.ips(x)
            mp, pos = net.ips(x) if not shuffle else (mp, pos)     # a second call: the kept buffers
            res.append((net.last_mem_idx.clone(), mp.clone(), None if pos is None else pos.clone()))
            for k in env:
                monkeypatch.delenv(k)
        for idx, mp, pos in res[1:]:
            assert torch.equal(idx, res[0][0]) and torch.equal(mp, res[0][1])
            assert (pos is None and res[0][2] is None) or torch.equal(pos, res[0][2])


def test_scan_range_resumes_exactly():
    lg = torch.randn((3, 1000, 32), generator=torch.Generator().manual_seed(5)).mul(3).to(DEV)
    want = hip.scan(lg, 32, 48, 8, 4)
    mem = torch.empty((3, 32), dtype=torch.int64, device=DEV)
    tie = torch.zeros((3,), dtype=torch.int32, device=DEV)
    n_iter = -(-(1000 - 32) // 48)
    for a, b in ((0, 1), (1, 7), (7, 8), (8, n_iter)):
        hip.scan_range(lg, 32, 48, 8, 4, a, b, mem, tie)
    assert torch.equal(mem, want)


@pytest.mark.gpu
def test_configs4_half_storage_and_bf16_logits_end_to_end(monkeypatch):
    """BASELINE configs[4] as written: patches stored in float16, bf16 trunk, bf16 logits (IPSX_PRECISION=bf16).  The
    reference has no reduced-precision behaviour; the selection must stay close to the exact path's (>= 80 % of the
    patches in common on the 2500-patch fixture) and ips() must hand back the winners in the storage type."""
    g = Golden("mnist_full")
    net = g.net(DEV)
    x = g.patches().to(DEV)
    net.ips(x)
    exact = net.last_mem_idx.cpu().numpy()
    monkeypatch.setenv("IPSX_PRECISION", "bf16")
    for dtype in (torch.float16, torch.bfloat16):
        xh = x.to(dtype)
        mem_patch, mem_pos = net.ips(xh)
        got = net.last_mem_idx.cpu().numpy()
        assert mem_patch.dtype == dtype and tuple(mem_patch.shape) == (g.B, g.conf.M, 1, 32, 32)
        assert torch.equal(mem_patch, xh[0][net.last_mem_idx[0]].unsqueeze(0))
        common = np.mean([len(set(a) & set(b)) / len(a) for a, b in zip(got.tolist(), exact.tolist())])
        assert common >= 0.80, common
        with torch.no_grad():
            preds = net(mem_patch, mem_pos)
        assert all(torch.isfinite(v).all() for v in preds.values())


@pytest.mark.gpu
def test_persistent_loop_equals_per_part_launches_and_recovers_from_a_timeout(monkeypatch):
    """Feature inputs: the selection loop as ONE persistent launch that follows the projector (ipsx_scan_persistent +
    gate + published row counts) selects exactly what the per-part launches select.  A loop whose rows never arrive ends
    by itself (bounded wait) and sets its status word - from EVERY workgroup - and the conditional launch behind it
    (ipsx_scan_range_if) redoes the loop in the same call; with the status word clear that launch changes nothing."""
    conf = synth.camelyon_conf(N=8192, M=64, I=64)
    net = synth.fill_weights(IPSNet(torch.device(DEV), conf), 5).to(DEV).eval()
    x = synth.make_patches(conf, 1, seed=9).to(DEV)
    monkeypatch.setenv("IPSX_SCAN_PERSIST", "0")
    net.ips(x)
    want = net.last_mem_idx.clone()
    monkeypatch.setenv("IPSX_SCAN_PERSIST", "1")
    for _ in range(3):
        net.ips(x)
        assert torch.equal(net.last_mem_idx, want)
    assert int(net.selection.scan_status.item()) & 1 == 0 and int(net.selection.scan_status.item()) & 2 == 2
    # a loop nobody feeds: negative progress word = cancelled -> it ends at once with the failure bit set ...
    B = 3
    lg = torch.randn((B, 1024, 8), device=DEV)
    plain = hip.scan(lg, 64, 64, 8, 1)
    mem = torch.full((B, 64), -7, dtype=torch.int64, device=DEV)
    tie = torch.zeros((B,), dtype=torch.int32, device=DEV)
    words = torch.tensor([-1, 0], dtype=torch.int32, device=DEV)
    hip.scan_persistent(lg, 64, 64, 8, 1, mem, tie, words[0:1], words[1:2])
    torch.cuda.synchronize()
    assert int(words[1].item()) & 1 == 1
    assert not torch.equal(mem, plain)
    # ... and the conditional launch behind it repairs the result without the host looking at anything
    n_iter = (1024 - 64) // 64
    hip.scan_range_if(lg, 64, 64, 8, 1, 0, n_iter, mem, tie, words[1:2], 1)
    assert torch.equal(mem, plain)
    mem.fill_(-7)
    words.zero_()
    hip.scan_range_if(lg, 64, 64, 8, 1, 0, n_iter, mem, tie, words[1:2], 1)        # status clear: every workgroup leaves
    assert int((mem != -7).sum().item()) == 0
    # the host learns of a timeout from the mirrored status word, one call later: ONE event re-runs the device's self-test
    # and - it passes - leaves the persistent pipelines on (a stalled host is not a broken device); the third event of
    # the process (IPSX_PERSIST_STRIKES) switches them off (the per-part launches take over) - results valid throughout
    monkeypatch.setattr(hip, "_PERSIST_OFF", None)              # (restored when this test ends)
    monkeypatch.setattr(hip, "_PERSIST_STRIKES", 0)
    monkeypatch.setattr(hip, "_PERSIST_RECENT", [])
    for event in (1, 2):
        net.selection.scan_status_host.fill_(1)
        with pytest.warns(UserWarning, match="self-test passes"):
            net.ips(x)
        assert torch.equal(net.last_mem_idx, want) and hip.persistent_ok(DEV) and hip._PERSIST_STRIKES == event
        assert net.selection.scan_status is not None and int(net.selection.scan_status.item()) & 3 == 2
    net.selection.scan_status_host.fill_(1)
    with pytest.warns(UserWarning, match="switched off"):
        net.ips(x)
    assert torch.equal(net.last_mem_idx, want) and not hip.persistent_ok(DEV)
    net.selection.scan_status = None
    net.ips(x)
    assert net.selection.scan_status is None and torch.equal(net.last_mem_idx, want)
    monkeypatch.setattr(hip, "_PERSIST_OFF", None)
    monkeypatch.setattr(hip, "_PERSIST_STRIKES", 0)
    monkeypatch.setattr(hip, "_PERSIST_RECENT", [])
    assert hip.persistent_ok(DEV)
    # ... and events further apart than IPSX_PERSIST_WINDOW calls do not add up
    monkeypatch.setenv("IPSX_PERSIST_WINDOW", "2")
    for _ in range(4):
        net.selection.scan_status_host.fill_(1)
        with pytest.warns(UserWarning, match="self-test passes"):
            net.ips(x)
        net.ips(x)
        net.ips(x)
        assert hip.persistent_ok(DEV) and torch.equal(net.last_mem_idx, want)
    monkeypatch.setattr(hip, "_PERSIST_RECENT", [])


@pytest.mark.gpu
def test_persistent_loop_for_candidate_sets_beyond_the_lds(monkeypatch):
    """ipsx_scan_persistent_ws / ipsx_scan_range_if_ws: the loop of a candidate set beyond the LDS (scan_large_kernel, the
    reference's shipped CAMELYON shape in small) follows the projector stream as one resident launch and selects what the
    per-part launches select; a cancelled loop sets its status word and the conditional launch with the workspace redoes it."""
    conf = synth.camelyon_conf(N=11000, M=2100, I=2100)
    net = synth.fill_weights(IPSNet(torch.device(DEV), conf), 5).to(DEV).eval()
    x = synth.make_patches(conf, 2, seed=9).to(DEV)
    ca = net.transf.crs_attn
    assert hip.scan_persistent_large(conf.M, conf.I, ca.H, ca.n_token) and not hip.scan_persistent_supported(conf.M, conf.I, ca.H, ca.n_token)
    monkeypatch.setenv("IPSX_LARGE_PERSIST", "0")
    want = []
    for xb in (x[:1], x):
        net.selection.scan_status = None
        net.ips(xb)
        assert net.selection.scan_status is None
        want.append(net.last_mem_idx.clone())
    monkeypatch.setenv("IPSX_LARGE_PERSIST", "1")
    for xb, w in zip((x[:1], x), want):
        for _ in range(2):
            net.selection.scan_status = None
            net.ips(xb)
            assert net.selection.scan_status is not None and torch.equal(net.last_mem_idx, w)
    assert int(net.selection.scan_status.item()) & 3 == 2
    # many calls on fresh slides of ragged sizes (the hand-over between the stream's tiles and the loop's waits)
    g = torch.Generator(device="cpu").manual_seed(11)
    cases = []
    for k in range(10):
        B = 1 + k % 2
        N = 32 * int(torch.randint(140, 420, (1,), generator=g)) if B > 1 else int(torch.randint(4300, 14000, (1,), generator=g))
        cases.append(torch.randn((B, N, conf.n_chan_in), generator=g).to(DEV))
    monkeypatch.setenv("IPSX_LARGE_PERSIST", "0")
    wants = []
    for xc in cases:
        net.ips(xc)
        wants.append(net.last_mem_idx.clone())
    monkeypatch.setenv("IPSX_LARGE_PERSIST", "1")
    for xc, w in zip(cases, wants):
        net.selection.scan_status = None
        net.ips(xc)
        assert torch.equal(net.last_mem_idx, w), tuple(xc.shape)
        assert (net.selection.scan_status is not None) == (net.selection.n_iter(xc.shape[1]) >= 3)   # (short loops: one scan)
    torch.cuda.synchronize()
    assert int(net.selection.scan_status_host.item()) & 1 == 0
    # a loop nobody feeds (negative progress words) gives up at once; the conditional launch repairs the result
    B, N, M, I, H = 2, 9000, 2100, 2100, 8
    lg = torch.randn((B, N, H), device=DEV)
    plain = hip.scan(lg, M, I, H, 1)
    ws = hip.scan_workspace(B, M, I, H, 1, lg.device)
    mem = torch.full((B, M), -7, dtype=torch.int64, device=DEV)
    tie = torch.zeros((B,), dtype=torch.int32, device=DEV)
    words = torch.tensor([-1, -1, 0], dtype=torch.int32, device=DEV)
    hip.scan_persistent(lg, M, I, H, 1, mem, tie, words[:2], words[2:], workspace=ws)
    torch.cuda.synchronize()
    assert int(words[2].item()) & 1 == 1 and not torch.equal(mem, plain)
    n_iter = -(-(N - M) // I)
    hip.scan_range_if(lg, M, I, H, 1, 0, n_iter, mem, tie, words[2:], 1, workspace=ws)
    assert torch.equal(mem, plain)
    mem.fill_(-7)
    words.zero_()
    hip.scan_range_if(lg, M, I, H, 1, 0, n_iter, mem, tie, words[2:], 1, workspace=ws)      # status clear: nothing runs
    assert int((mem != -7).sum().item()) == 0


@pytest.mark.gpu
def test_stream_hand_over_stress(monkeypatch):
    """The hand-over between the persistent producers and the resident loops (flags, cursor, progress words: relaxed
    atomics behind one release fence, an acquire per wait), many times over: 40 calls with fresh slides of ragged sizes,
    1 / 2 / 5 slides per call (5: two loop workgroups take the slides in turn), each held to the per-part launches."""
    conf = synth.camelyon_conf(N=4096, M=256, I=256)
    net = synth.fill_weights(IPSNet(torch.device(DEV), conf), 5).to(DEV).eval()
    g = torch.Generator(device="cpu").manual_seed(3)
    cases = []
    for k in range(40):
        B = (1, 2, 5)[k % 3]
        N = 32 * int(torch.randint(40, 400, (1,), generator=g)) if B > 1 else int(torch.randint(1300, 12000, (1,), generator=g))
        cases.append(torch.randn((B, N, conf.n_chan_in), generator=g).to(DEV))
    monkeypatch.setenv("IPSX_SCAN_PERSIST", "0")
    want = []
    for x in cases:
        net.ips(x)
        want.append(net.last_mem_idx.clone())
    monkeypatch.setenv("IPSX_SCAN_PERSIST", "1")
    used = 0
    for rep in range(2):
        for x, w in zip(cases, want):
            net.selection.scan_status = None
            net.ips(x)
            used += net.selection.scan_status is not None
            assert torch.equal(net.last_mem_idx, w), "slides %s, call %d" % (tuple(x.shape), rep)
    torch.cuda.synchronize()
    assert used == 2 * len(cases)                       # every call went through the persistent pipeline
    assert int(net.selection.scan_status_host.item()) & 1 == 0


@pytest.mark.gpu
def test_library_calls_from_two_threads_keep_their_hand_overs_apart():
    """ipsx_ips_call_run shares one pair of hand-over events per device and enqueues under a per-device lock: two host
    threads, each with a net and a stream of its own (one image on the fused trunk, one slide through the projector), 40
    calls each at the same time - every call selects what the same net selects alone."""
    import threading
    dev = torch.device(DEV)
    confs = [synth.mnist_conf(N=2500, M=64, I=64), synth.camelyon_conf(N=8192, M=256, I=256)]
    nets = [synth.fill_weights(IPSNet(dev, c), 3 + k).to(DEV).eval() for k, c in enumerate(confs)]
    xs = [synth.make_patches(c, 1, seed=4 + k).to(DEV) for k, c in enumerate(confs)]
    want = []
    for net, x in zip(nets, xs):
        net.ips(x)
        assert net.selection.scan_status is not None          # the resident-loop pipeline, through the library call
        want.append(net.last_mem_idx.clone())
    torch.cuda.synchronize()
    wrong, errors = [0, 0], []

    def work(k):
        try:
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                for _ in range(40):
                    nets[k].ips(xs[k])
                    wrong[k] += 0 if torch.equal(nets[k].last_mem_idx, want[k]) else 1
                st.synchronize()
        except Exception as e:                                 # noqa: BLE001 - reported by the assert below
            errors.append(repr(e))
    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not any(t.is_alive() for t in threads) and not errors, errors
    assert wrong == [0, 0]


@pytest.mark.gpu
@pytest.mark.parametrize("groups", [1, 2, 3, 5])
def test_persistent_loops_of_several_slides_on_fewer_workgroups(groups):
    """ipsx_scan_persistent_on: the loops of 5 slides (the CAMELYON shape, M = I = 256) on 1 / 2 / 3 / 5 resident
    workgroups, each taking its slides one after the other while the rows are published slide by slide from another
    stream, select what the plain loop selects (ragged last chunk; a per-slide progress word)."""
    B, N, M, I, H = 5, 256 + 7 * 256 + 100, 256, 256, 8
    assert hip.scan_persistent_groupable(M, I, H, 1) and not hip.scan_persistent_groupable(64, 64, H, 1)
    lg = torch.randn((B, N, H), device=DEV)
    plain = hip.scan(lg, M, I, H, 1)
    mem = torch.full((B, M), -7, dtype=torch.int64, device=DEV)
    tie = torch.zeros((B,), dtype=torch.int32, device=DEV)
    words = torch.zeros((B + 1,), dtype=torch.int32, device=DEV)
    side = hip.side_stream(DEV)                     # (THE side stream: a fresh one may share the main stream's hardware queue,
    side.wait_stream(torch.cuda.current_stream())   #  and a resident loop then keeps its own producers from running)
    with torch.cuda.stream(side):
        hip.scan_persistent(lg, M, I, H, 1, mem, tie, words[:B], words[B:], workgroups=groups)
    hip.scan_gate(words[B:])
    for b in range(B):                              # the producer's order: slide by slide, a few steps each
        for rows in (M + I, N // 2, N):
            hip.publish_rows(words[b:b + 1], rows)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert int(words[B].item()) & 1 == 0
    assert torch.equal(mem, plain)
    # a cancelled call (negative progress words): every workgroup gives up on its FIRST slide, the slides behind it are
    # never started - and the conditional launch behind the loop redoes all of them
    mem.fill_(-7)
    words.fill_(-1)
    words[B] = 0
    with torch.cuda.stream(side):
        hip.scan_persistent(lg, M, I, H, 1, mem, tie, words[:B], words[B:], workgroups=groups)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert int(words[B].item()) & 1 == 1 and not torch.equal(mem, plain)
    hip.scan_range_if(lg, M, I, H, 1, 0, -(-(N - M) // I), mem, tie, words[B:], 1)
    assert torch.equal(mem, plain)
    if groups < B:                                  # shapes of the generic LDS loop: one workgroup per image only
        lg2 = torch.randn((3, 1024, H), device=DEV)
        with pytest.raises(RuntimeError, match="fewer workgroups"):
            hip.scan_persistent(lg2, 64, 64, H, 1, mem[:3, :64].contiguous(), tie[:3], words[:3], words[B:], workgroups=2)


@pytest.mark.gpu
def test_persistent_self_test_sees_serialised_kernels():
    """Whether a persistent loop can run beside its producers is established by doing it once per device
    (hip.persistent_ok), not by looking for profiler / debug variables: on this box it can; with the runtime told to
    serialise every kernel (AMD_SERIALIZE_KERNEL=3, what counter collection does too) the self-test's loop times out after
    the bounded wait and the persistent pipelines stay off - ips() then selects the same patches with per-part launches."""
    import subprocess
    import sys
    assert hip.persistent_ok(DEV)
    code = ("import torch\n"
            "from ips_amd import hip, synth\n"
            "from ips_amd.architecture import IPSNet\n"
            "dev = torch.device('cuda:0')\n"
            "ok = hip.persistent_ok(dev)\n"
            "conf = synth.camelyon_conf(N=4096, M=64, I=64)\n"
            "net = synth.fill_weights(IPSNet(dev, conf), 5).to(dev).eval()\n"
            "net.ips(synth.make_patches(conf, 1, seed=9).to(dev))\n"
            "print('persistent_ok', ok, 'used', net.selection.scan_status is not None, 'sum', int(net.last_mem_idx.sum()))\n")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for extra in ({}, {"AMD_SERIALIZE_KERNEL": "3"}):
        env = dict(os.environ, PYTHONPATH=repo, **extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines()[-1].split())
    assert outs[0][1] == "True" and outs[0][3] == "True", outs
    assert outs[1][1] == "False" and outs[1][3] == "False", outs
    assert outs[0][5] == outs[1][5], outs                      # the same selection either way


@pytest.mark.gpu
@pytest.mark.parametrize("B,N", [(1, 8192), (2, 8192), (9, 4096), (1, 70000)])
def test_every_variant_of_the_feature_pipeline_selects_the_same_patches(B, N, monkeypatch):
    """The CAMELYON path has several schedules of the same arithmetic - persistent loop or per-part launches (also what
    more than IPSX_PERSIST_MAX_B slides get), loop beside the projector or after it, the projector of one slide as ONE
    persistent launch that publishes tile by tile (ipsx_projector_stream: the default for one slide) or launch by launch,
    equal parts or the latency-shaped layout with half-tile launches: the selected indices are the same in all of them (the un-overlapped one is held
    against the oracle and the reference's recordings elsewhere)."""
    import os
    from ips_amd import synth
    from ips_amd.architecture import IPSNet
    dev = torch.device("cuda:0")
    conf = synth.camelyon_conf(N=N, M=256, I=256)
    net = synth.fill_weights(IPSNet(dev, conf), 7).to(dev).eval()
    x = synth.make_patches(conf, B, seed=3).to(dev)
    res = {}
    for name, env in (("default", {}), ("the entry points one by one instead of ONE library call", {"IPSX_NATIVE_CALL": "0"}),
                      ("per-part launches", {"IPSX_SCAN_PERSIST": "0"}), ("after", {"IPSX_OVERLAP_SCAN": "0"}),
                      ("launch by launch beside the persistent loop", {"IPSX_CAM_STREAM": "0"}),
                      ("latency-shaped parts", {"IPSX_CAM_PARTS": "latency", "IPSX_CAM_STREAM": "0"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        net.ips(x)
        net.ips(x)                                  # (a second call: cached buffers, the status mirror of the first)
        res[name] = net.last_mem_idx.clone()
        for k in env:
            monkeypatch.delenv(k)
    for name, idx in res.items():
        assert torch.equal(idx, res["after"]), name


@pytest.mark.gpu
@pytest.mark.parametrize("conf_fn,patch,n", [(synth.mnist_conf, 50, 2100), (synth.traffic_conf, 100, 1100)])
def test_layered_trunk_on_two_streams_gives_the_same_bits(conf_fn, patch, n, monkeypatch):
    """EncoderPlan.encode_plain sends the batch through the layer-by-layer trunk in two halves on two streams (default)
    or in one piece (IPSX_LAYERED_STREAMS=1): the same kernels on the same patches, so the embeddings are bit-identical -
    also across repeated calls (the halves share one workspace buffer)."""
    dev = torch.device("cuda:0")
    conf = conf_fn(N=64, M=8, I=8, patch=patch)
    net = synth.fill_weights(IPSNet(dev, conf), 11).to(dev).eval()
    g = torch.Generator(device="cpu").manual_seed(4)
    x = torch.rand((n, conf.n_chan_in, patch, patch), generator=g).to(dev)
    plan = hip.EncoderPlan(net.encoder, True)
    assert not plan.fused(x.shape)
    two = [plan.encode(x).clone() for _ in range(3)]
    monkeypatch.setenv("IPSX_LAYERED_STREAMS", "1")
    one = plan.encode(x)
    torch.cuda.synchronize()
    for t in two:
        assert torch.equal(t, one)
