#!/usr/bin/env python
"""tests/golden/mnist_disk/: a tiny Megapixel-MNIST dataset in the reference's on-disk format (written by
ips_amd.data.megapixel_mnist.write_synthetic) and what the REFERENCE's dataset class
(/root/reference/data/megapixel_mnist/mnist_dataset.py, imported from where it lies) returns for it:
patch tensors and labels per item, for two patch geometries (non-overlapping and 50 % overlap).

    python tools/gen_golden_data.py
"""
import os
import sys

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np

from ips_amd import synth
from ips_amd.data import megapixel_mnist as mm

OUT = os.path.join(REPO, "tests", "golden", "mnist_disk")
GEOMS = {"p32s32": ([32, 32], [32, 32]), "p50s25": ([50, 50], [25, 25])}


def main():
    mm.write_synthetic(OUT, n_train=3, n_test=2, width=200, height=150, n_noise=6, seed=4)
    sys.path.insert(0, "/root/reference")
    from data.megapixel_mnist.mnist_dataset import MegapixelMNIST as Ref
    out = {}
    for tag, (ps, st) in GEOMS.items():
        conf = synth.mnist_conf(data_dir=OUT, patch_size=ps, patch_stride=st)
        for split, train in (("train", True), ("test", False)):
            ds = Ref(conf, train=train)
            for i in range(len(ds)):
                item = ds[i]
                key = "%s_%s_%d" % (tag, split, i)
                out[key + "_input"] = item['input'].numpy()
                for t in conf.tasks.values():
                    out[key + "_" + t['name']] = np.asarray(item[t['name']])
    np.savez_compressed(os.path.join(OUT, "expected.npz"), **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items() if k.endswith("0_input")})


if __name__ == "__main__":
    main()
