#!/usr/bin/env python
"""How far from sorted is the memory the selection loop carries into its next iteration?  (CPU, float32 numpy replay of the
loop on a bench workload's logits.)  The review of round 3 proposed to verify the carried order and repair it locally
instead of re-sorting: this prints, per iteration, the adjacent inversions and the largest displacement of the carried
memory under the NEXT iteration's scores, and the chunk candidates that reach the lowest memory score.
    python tools/order_stats.py cam [rows] | mnist"""
import sys, math, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ips_amd import synth
from ips_amd.architecture import IPSNet
name = sys.argv[1] if len(sys.argv) > 1 else "cam"
conf, B = synth.bench_workload(name)
if name == "cam": conf = conf.clone(N=int(sys.argv[2]) if len(sys.argv) > 2 else 32768)
net = synth.fill_weights(IPSNet(torch.device("cpu"), conf), 7).eval()
x = synth.make_patches(conf, 1, seed=21)
torch.set_num_threads(8)
with torch.no_grad():
    N = x.shape[1]
    emb = torch.cat([net.encoder(x[0, i:i+4096].reshape(-1, *x.shape[2:])).flatten(1) for i in range(0, N, 4096)])
    if conf.use_pos: emb = emb + net.pos_enc[0, :N]
    ca = net.transf.crs_attn
    q = ca.q_w(ca.q) / ca.attention.temperature      # (1,T,H*Dk)
    k = ca.k_w(emb)                                   # (N,H*Dk)
    H, T, Dk = ca.H, ca.n_token, ca.D_k
    qh = q.view(T, H, Dk); kh = k.view(N, H, Dk)
    lg = torch.einsum("thd,nhd->nht", qh, kh).numpy().astype(np.float32)   # (N,H,T)
M, I = conf.M, conf.I
mem = np.arange(M)
n_iter = math.ceil((N - M) / I)
inv_adj, maxdisp, surv, sortedok, ninv = [], [], [], 0, []
heads_same, prev_md = [], None   # round 5's review, item 5: heads whose (max, denominator) did not move
for it in range(n_iter):
    lo = M + it * I; hi = min(N, lo + I)
    cand = np.concatenate([mem, np.arange(lo, hi)])
    xl = lg[cand]                                   # (L,H,T)
    e = np.exp(xl - xl.max(0, keepdims=True)); w = e / e.sum(0, keepdims=True)
    sc = w.mean(1).mean(1)
    ms = sc[:M]
    md = np.stack([xl.max(0), e.sum(0)])            # (2,H,T): what a memory row's weight depends on beside its own logit
    if prev_md is not None:
        heads_same.append(int((md == prev_md).all(0).sum()))
    prev_md = md
    if it > 0:
        adj = int((ms[:-1] < ms[1:]).sum()); inv_adj.append(adj)
        order = np.argsort(-ms, kind="stable"); rank = np.empty(M, int); rank[order] = np.arange(M)
        maxdisp.append(int(np.abs(rank - np.arange(M)).max()))
        sortedok += adj == 0
    surv.append(int((sc[M:] >= ms.min()).sum()))
    top = np.argsort(-sc, kind="stable")[:M]
    mem = cand[top]
print(name, "N", N, "iters", n_iter)
for nm, a in (("adjacent inversions in carried memory", inv_adj), ("max displacement", maxdisp), ("survivors", surv)):
    a = np.array(a); print("%-40s mean %.1f median %.0f p90 %.0f max %d" % (nm, a.mean(), np.median(a), np.percentile(a, 90), a.max()))
a = np.array(heads_same); print("(head, token) rows whose (max, denominator) is bit-equal to the previous iteration's: mean %.2f max %d of %d rows, in %d of %d iterations any"
                              % (a.mean(), a.max(), H * T, int((a > 0).sum()), len(a)))
print("iterations with carried memory exactly sorted:", sortedok, "of", n_iter - 1)
