import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ips_amd import synth, hip
from ips_amd.architecture import IPSNet
dev = torch.device("cuda:0")
conf = synth.mnist_conf(N=2500, M=64, I=64, B=16, B_seq=16, n_epoch=10, n_epoch_warmup=1, lr=1e-3, wd=0.1)
net = synth.fill_weights(IPSNet(dev, conf), 7).to(dev)
x = synth.make_patches(conf, 16, seed=3).to(dev)
print("x", x.shape, x.dtype, x.is_contiguous(), "shuffle", net.shuffle, "nonzero frac", float((x != 0).float().mean()))
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
for mode in ("eval", "train", "eval"):
    getattr(net, mode)()
    print(mode, "ips %.2f ms" % timed(lambda: net.ips(x)))
net.shuffle = False
for mode in ("eval", "train"):
    getattr(net, mode)()
    print("no shuffle", mode, "ips %.2f ms" % timed(lambda: net.ips(x)))
