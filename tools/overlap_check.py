import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ips_amd import synth
from ips_amd.architecture import IPSNet
dev = torch.device("cuda:0")
for B, N in ((4, 20000), (16, 5000)):
    conf = synth.mnist_conf(N=N, M=64, I=64)
    net = synth.fill_weights(IPSNet(dev, conf), 7).to(dev).eval()
    x = synth.make_patches(conf, B, seed=3).to(dev)
    for mode in ("1", "0", "1", "0"):
        os.environ["IPSX_OVERLAP_SCAN"] = mode
        for _ in range(3): net.ips(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): net.ips(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print("B=%d N=%d overlap=%s: %.3f ms/step  %.3f M patches/s" % (B, N, mode, dt * 1e3, B * N / dt / 1e6))
