#!/usr/bin/env python
"""Can the bf16 trunk and the selection loops be kept apart by compute-unit masks?  (beside a trunk launch that
fills the chip the loop workgroups of a call wait for a whole free unit.)  Streams made by hipExtStreamCreateWithCUMask: the
bf16 trunk alone on a stream with 256 / 240 / 128 of the mask's bits set - the rate follows the bits.  What became of it:
profiles/r06_bf16_cu_masks.txt (a pipeline on such streams was built, measured at 26.7 M patches/s against 27.7 and taken out).
    python tools/cu_mask_probe.py"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["IPSX_PRECISION"] = "bf16"
from ips_amd import hip, synth
from ips_amd.architecture import IPSNet


def hip_runtime():
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            return C.CDLL(line.split()[-1])
    return C.CDLL("libamdhip64.so")


def masked_stream(rt, bits):
    words = (C.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    s = C.c_void_p()
    rc = rt.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask: %d" % rc)
    return torch.cuda.ExternalStream(s.value)


def main():
    dev = torch.device("cuda:0")
    conf = synth.mnist_conf(N=2500)
    net = synth.fill_weights(IPSNet(dev, conf), 7).to(dev).eval()
    x = synth.make_patches(conf, 20, seed=21).reshape(-1, 1, 32, 32).contiguous().to(dev)
    plan = hip.EncoderPlan(net.encoder, True)
    rt = hip_runtime()
    rt.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
    torch.cuda.synchronize()
    full = (1 << 256) - 1
    masks = [("all 256 bits", full), ("bits 0..239", (1 << 240) - 1), ("bits 16..255", full & ~0xFFFF), ("bits 0..127", (1 << 128) - 1),
             ("every bit but 8 k, 8 k + 1 (k < 8) of each 32", sum(((1 << 32) - 1 - 3) << (32 * i) for i in range(8)))]
    for name, bits in masks:
        st = masked_stream(rt, bits)
        for n in (40960, 38400, 40000):
            xs = x[:n]
            with torch.cuda.stream(st):
                for _ in range(3):
                    plan.encode(xs)
                st.synchronize()
                t0 = time.perf_counter()
                for _ in range(10):
                    plan.encode(xs)
                st.synchronize()
            dt = (time.perf_counter() - t0) / 10
            print("%-50s %6d patches  %7.1f us  %5.2f M patches/s" % (name, n, dt * 1e6, n / dt / 1e6), flush=True)


if __name__ == "__main__":
    main()
