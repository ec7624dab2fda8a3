#!/usr/bin/env python
"""Fuzzing of the SCHEDULES around the kernels: random slide counts and ragged slide lengths through the feature
pipeline, random patch counts through the one-image pipeline - every schedule (one library call with a resident loop,
the entry points one by one, per-part launches, loop after the encoder) must select the same patches, call after call.
What tests/test_hip_e2e.py::test_every_variant_of_the_feature_pipeline... and ::test_one_image_every_schedule... do for a
few shapes, for many.
    python tools/fuzz_pipelines.py [first_seed] [n_seeds]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ips_amd import hip, synth                      # noqa: E402
from ips_amd.architecture import IPSNet             # noqa: E402

FEATURE_ENVS = (("default", {}), ("entry points one by one", {"IPSX_NATIVE_CALL": "0"}), ("per-part launches", {"IPSX_SCAN_PERSIST": "0"}),
                ("after", {"IPSX_OVERLAP_SCAN": "0"}), ("launch by launch beside the loop", {"IPSX_CAM_STREAM": "0"}))
IMAGE_ENVS = (("default", {}), ("entry points one by one", {"IPSX_NATIVE_CALL": "0"}), ("per-part launches", {"IPSX_SCAN_PERSIST": "0"}),
              ("after", {"IPSX_OVERLAP_SCAN": "0"}))


def run(net, x, envs):
    res = {}
    for name, env in envs:
        os.environ.update(env)
        try:
            net.ips(x)
            net.ips(x)                               # (cached buffers, the status mirror of the first call)
            res[name] = net.last_mem_idx.clone()
        finally:
            for k in env:
                del os.environ[k]
    return res


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    dev = torch.device("cuda:0")
    bad, t0 = 0, time.time()
    for seed in range(first, first + n):
        g = np.random.default_rng(7000 + seed)
        if seed % 3 == 2:                            # one image on the fused trunk (32-px patches)
            N = int(g.integers(66, 6000))
            M = I = 64
            conf = synth.mnist_conf(N=N, M=M, I=I, use_pos=bool(g.integers(0, 2)))
            B, envs, what = 1, IMAGE_ENVS, "image"
        elif seed % 7 == 6 or os.environ.get("LARGE") == "1":
            # feature slides with a candidate set beyond the LDS (the reference's shipped M = I = 5000 and neighbours): the
            # loop is a TEAM of workgroups per slide (csrc/scan_large_team.h) in every schedule that has resident loops
            M = int(g.choice([5000, 4200, 2300]))
            I = int(g.choice([M, 5000, 3000]))
            B = int(g.choice([1, 1, 2, 3]))
            N = int(g.integers(M + 1, 45000 if B == 1 else 20000))
            conf = synth.camelyon_conf(N=N, M=M, I=I)
            envs, what = FEATURE_ENVS, "features (beyond the LDS)"
        else:                                        # feature slides: the CAMELYON shape or a generic one
            cam = bool(g.integers(0, 2))
            M = I = 256 if cam else int(g.choice([32, 64, 128]))
            B = int(g.choice([1, 1, 2, 3, 5, 9, 16]))
            N = int(g.integers(M + 1, 20000 if B > 2 else 70000))
            conf = synth.camelyon_conf(N=N, M=M, I=I)
            envs, what = FEATURE_ENVS, "features"
        net = synth.fill_weights(IPSNet(dev, conf), 7 + seed).to(dev).eval()
        x = synth.make_patches(conf, B, seed=3 + seed).to(dev)
        res = run(net, x, envs)
        ok = all(torch.equal(v, res["after"]) for v in res.values())
        if not ok:
            bad += 1
            print("seed %d (%s, B %d, N %d, M %d): DIFFER: %s" % (seed, what, B, N, M, [k for k, v in res.items() if not torch.equal(v, res["after"])]),
                  flush=True)
        del net, x
    torch.cuda.synchronize()
    print("seeds %d..%d: %d failures (%.2f s per seed); persistent timeouts seen by the host: %d" % (
        first, first + n - 1, bad, (time.time() - t0) / max(n, 1), hip._PERSIST_STRIKES))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
