#!/usr/bin/env python
"""Diagnostic: where a workgroup of fused_stage64_kernel (layer1 of the 50-px trunk, LDS-resident) spends its cycles.
    python tools/stage_stamps.py [patches]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ips_amd import hip, synth
from ips_amd.architecture import IPSNet

n = int(sys.argv[1]) if len(sys.argv) > 1 else 7200
dev = torch.device("cuda:0")
conf = synth.mnist_conf(N=900, M=100, I=100, patch=50)
net = synth.fill_weights(IPSNet(dev, conf), 7).to(dev).eval()
x = torch.rand((n, 1, 50, 50), device=dev)
plan = hip.EncoderPlan(net.encoder, True)
os.environ["IPSX_LAYERED_STREAMS"] = "1"
plan.encode(x)
L = hip.lib()
L.ipsx_dbg_fused_stage_stamps.argtypes = [C.c_void_p]
st = torch.zeros((16,), dtype=torch.int64, device=dev)
L.ipsx_dbg_fused_stage_stamps(st.data_ptr())
plan.encode(x)
torch.cuda.synchronize()
L.ipsx_dbg_fused_stage_stamps(None)
s = st.cpu().numpy()
names = ["load input + masks", "conv b0.c1", "epilogue", "conv b0.c2", "epilogue (+identity)", "conv b1.c1", "epilogue",
         "conv b1.c2", "epilogue (+identity)"]
tot = s[9] - s[0]
print("workgroup 0, wave 0: %d cycles; a convolution's MFMAs alone are 72 stages x 32 x 64 = 147456" % tot)
for k, nme in enumerate(names):
    print("  %-24s %8d" % (nme, s[k + 1] - s[k]))
