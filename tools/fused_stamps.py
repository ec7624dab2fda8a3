#!/usr/bin/env python
"""Diagnostic: per-phase cycle breakdown of fused_trunk_kernel from in-kernel s_memtime stamps.

    python tools/fused_stamps.py [n_patches] [fp32|fp32x3|bf16]

Runs the STAMP build (ipsx_dbg_fused_trunk_stamps) on synthetic patches and prints, per phase,
the median wave-cycles and the matrix-pipe cycles the phase's MFMAs need alone (64 cycles each).
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ips_amd import hip, synth
from ips_amd.architecture import IPSNet

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
os.environ["IPSX_PRECISION"] = prec
dev = torch.device("cuda:0")
conf = synth.mnist_conf(N=2500)
net = synth.fill_weights(IPSNet(dev, conf), 7).to(dev).eval()
x = synth.make_patches(conf, (n + 2499) // 2500, seed=21).reshape(-1, 1, 32, 32)[:n].contiguous().to(dev)
plan = hip.EncoderPlan(net.encoder, True)
ref = plan.encode(x)
L = hip.lib()
fn = L.ipsx_dbg_fused_trunk_stamps
fn.restype = C.c_int
fn.argtypes = [C.POINTER(hip.Trunk), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
n_wave = ((n + 7) // 8) * 8
st = torch.zeros((n_wave, 16), dtype=torch.int64, device=dev)
out = torch.empty_like(ref)
for _ in range(3):
    rc = fn(C.byref(plan.trunk), x.data_ptr(), n, out.data_ptr(), st.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
torch.cuda.synchronize()
assert torch.equal(out, ref)
s = st.cpu().numpy().astype(np.int64)
if prec == "bf16" and os.environ.get("IPSX_BF16_BUILD", "3") == "3":
    # the third build: eight patches per workgroup, two quads one after the other - the SECOND quad's rows (4 .. 7 of every 8)
    # carry its own phases and, right behind them, the 4x4 stage over all eight (per PATCH: halve those phases)
    s = s[: (n // 8) * 8].reshape(-1, 8, 16)[:, 4:, :].reshape(-1, 16)
    print("third build: rows of the second quad of every workgroup; the 4x4 stage's phases (l2.*) are for EIGHT patches")
names = ["load", "stem+pool", "l1.0.c1", "l1.0.c1 epi", "l1.0.c2", "l1.0.c2 epi", "l1.1.c1", "l1.1.c1 epi",
         "l1.1.c2", "l1.1.c2 epi", "l2.0.c1+down", "l2.0 epi+c2", "l2.1.c1", "l2.1.c2", "avgpool"]
mfma = [0, 400, 1152, 0, 1152, 0, 1152, 0, 1152, 0, 640, 1152, 1152, 1152, 0]
if prec == "fp32x3":      # bf16 MFMAs of 32 cycles, 6 per 8 fp32 MFMAs of 64 -> in units of 64 cycles: x 6/16 (stem: 4 K-steps x 6 x 2 n-tiles x 8 tiles)
    mfma = [(8 * 2 * 4 * 6 / 2) if k == 1 else m * 6 / 16 for k, m in enumerate(mfma)]
if prec == "bf16":        # one bf16 MFMA of 32 cycles per 8 fp32 MFMAs of 64: x 1/16 (stem: 4 K-steps x 2 n-tiles x 8 tiles)
    mfma = [(8 * 2 * 4 / 2) if k == 1 else m / 16 for k, m in enumerate(mfma)]
# stamps: 0 start,1 loaded,2 stem,3 c1,4 epi,5 c2,6 epi(+barrier),7..10 block 1,11 l2.0 c1+down,12 cv5,13 cv6,14 cv7,15 end
life = s[:, 15] - s[:, 0]
print("waves %d  median life %d cycles  (matrix-pipe cycles alone: %d)" % (len(s), np.median(life), sum(mfma) * 64))
for k in range(15):
    d = s[:, k + 1] - s[:, k]
    print("%-14s median %8d  p10 %8d  p90 %8d   mfma-alone %7d  ratio %.2f" % (
        names[k], np.median(d), np.percentile(d, 10), np.percentile(d, 90), mfma[k] * 64,
        (np.median(d) / (mfma[k] * 64)) if mfma[k] else float("nan")))
span = s[:, 15].max() - s[:, 0].min()
print("kernel span %d cycles; sum of MFMA cycles per SIMD %d -> pipe utilisation %.3f" % (
    span, int(len(s) * sum(mfma) * 64 / 1024), len(s) * sum(mfma) * 64 / 1024 / span))
