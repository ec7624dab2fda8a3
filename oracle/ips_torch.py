"""ATen restatement of the reference's CPU path - the `cpu_baseline` of bench.py.

TEST INFRASTRUCTURE (see oracle/ips_oracle.cpp header): imported only by tests/ and
by the cpu_baseline leg of bench.py, never by ips_amd/.

The reference (/root/reference, Python) cannot travel to the GPU box, and its speed
on a CPU is the speed of the ATen/oneDNN/MKL kernels it dispatches.  This module
restates `IPSNet.ips` + `IPSNet.forward` (architecture/ips_net.py:169-283,
architecture/transformer.py:29-152) as plain functions over a state-dict, calling
the same ATen ops in the same order on the same chunks, so that timing it IS timing
the reference's CPU path ("kind": "port").  tests/test_ips_torch.py checks it against
the golden fixtures (identical indices; identical bits when run on the machine that
generated them).
"""

import math

import torch
import torch.nn.functional as F


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"],
                        sd[p + ".bias"], False, 0.1, 1e-5)


def _basic_block(x, sd, p, stride):
    idt = x
    y = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"], None, stride, 1), sd, p + ".bn1"))
    y = _bn(F.conv2d(y, sd[p + ".conv2.weight"], None, 1, 1), sd, p + ".bn2")
    if p + ".downsample.0.weight" in sd:
        idt = _bn(F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride, 0), sd, p + ".downsample.1")
    return F.relu(y + idt)


def encode(sd, x, is_image):
    """IPSNet.encoder in eval mode (ips_net.py:17-60)."""
    if not is_image:
        y = F.layer_norm(x, (x.shape[-1],), None, None, 1e-5)
        y = F.linear(y, sd["encoder.1.weight"], sd["encoder.1.bias"])
        return F.relu(_bn(y, sd, "encoder.2"))
    y = F.relu(_bn(F.conv2d(x, sd["encoder.0.weight"], None, 2, 3), sd, "encoder.1"))
    y = F.max_pool2d(y, 3, 2, 1)
    stage = 4
    while "encoder.%d.0.conv1.weight" % stage in sd:
        for blk in (0, 1):
            p = "encoder.%d.%d" % (stage, blk)
            y = _basic_block(y, sd, p, 2 if (blk == 0 and stage > 4) else 1)
        stage += 1
    return F.adaptive_avg_pool2d(y, 1).flatten(1)


def get_scores(sd, x, H, Dk, T):
    """Transformer.get_scores (transformer.py:71-83, 29-34, 143-148), eval mode."""
    B, L = x.shape[:2]
    q = F.linear(sd["transf.crs_attn.q"], sd["transf.crs_attn.q_w.weight"]).view(1, T, H, Dk).transpose(1, 2)
    k = F.linear(x, sd["transf.crs_attn.k_w.weight"]).view(B, L, H, Dk).transpose(1, 2)
    attn = torch.softmax(torch.matmul(q / Dk ** 0.5, k.transpose(2, 3)), dim=-1)
    return attn.mean(dim=1).transpose(1, 2).mean(-1)


@torch.no_grad()
def ips(sd, conf, patches, pos_enc=None, trace=None):
    """IPSNet.ips (ips_net.py:169-262) without the shuffle; returns (mem_patch, mem_pos, mem_idx)."""
    M, I, D = conf.M, conf.I, conf.D
    B, N = patches.shape[:2]
    shape = patches.shape
    pos = pos_enc.expand(B, -1, -1) if conf.use_pos else None
    mem_emb = encode(sd, patches[:, :M].reshape(-1, *shape[2:]), conf.is_image).view(B, M, -1)
    idx = torch.arange(N, dtype=torch.int64).unsqueeze(0).expand(B, -1)
    mem_idx = idx[:, :M]
    for i in range(math.ceil((N - M) / I)):
        lo = i * I + M
        hi = min(lo + I, N)
        it_emb = encode(sd, patches[:, lo:hi].reshape(-1, *shape[2:]), conf.is_image).view(B, -1, D)
        all_emb = torch.cat((mem_emb, it_emb), dim=1)
        all_idx = torch.cat((mem_idx, idx[:, lo:hi]), dim=1)
        scored = all_emb
        if conf.use_pos:
            scored = all_emb + torch.gather(pos, 1, all_idx.view(B, -1, 1).expand(-1, -1, D))
        top = torch.topk(get_scores(sd, scored, conf.H, conf.D_k, conf.n_token), M, dim=-1)[1]
        mem_emb = torch.gather(all_emb, 1, top.unsqueeze(-1).expand(-1, -1, D))
        mem_idx = torch.gather(all_idx, 1, top)
        if trace is not None:
            trace.append(mem_idx.clone())
    take = mem_idx.view(B, -1, *(1,) * (len(shape) - 2)).expand(-1, -1, *shape[2:])
    mem_patch = torch.gather(patches, 1, take)
    mem_pos = torch.gather(pos, 1, mem_idx.unsqueeze(-1).expand(-1, -1, D)) if conf.use_pos else None
    return mem_patch, mem_pos, mem_idx


@torch.no_grad()
def forward(sd, conf, mem_patch, mem_pos=None):
    """IPSNet.forward in eval mode (ips_net.py:264-283)."""
    B, M = mem_patch.shape[:2]
    H, Dk, Dv, T = conf.H, conf.D_k, conf.D_v, conf.n_token
    x = encode(sd, mem_patch.reshape(-1, *mem_patch.shape[2:]), conf.is_image).view(B, M, -1)
    if mem_pos is not None:
        x = x + mem_pos
    p = "transf.crs_attn."
    q = F.linear(sd[p + "q"], sd[p + "q_w.weight"]).view(1, T, H, Dk).transpose(1, 2)
    k = F.linear(x, sd[p + "k_w.weight"]).view(B, M, H, Dk).transpose(1, 2)
    v = F.linear(x, sd[p + "v_w.weight"]).view(B, M, H, Dv).transpose(1, 2)
    attn = torch.softmax(torch.matmul(q / Dk ** 0.5, k.transpose(2, 3)), dim=-1)
    y = torch.matmul(attn, v).transpose(1, 2).contiguous().view(B, T, -1)
    y = F.linear(y, sd[p + "fc.weight"]) + sd[p + "q"]
    y = F.layer_norm(y, (conf.D,), sd[p + "layer_norm.weight"], sd[p + "layer_norm.bias"], 1e-6)
    m = "transf.mlp."
    z = F.linear(torch.relu(F.linear(y, sd[m + "w_1.weight"], sd[m + "w_1.bias"])), sd[m + "w_2.weight"],
                 sd[m + "w_2.bias"]) + y
    z = F.layer_norm(z, (conf.D,), sd[m + "layer_norm.weight"], sd[m + "layer_norm.bias"], 1e-6)
    preds = {}
    for task in conf.tasks.values():
        o = "output_layers.%s.0." % task['name']
        logit = F.linear(z[:, task['id']], sd[o + "weight"], sd[o + "bias"])
        preds[task['name']] = torch.softmax(logit, -1) if task['act_fn'] == 'softmax' else torch.sigmoid(logit)
    return preds
