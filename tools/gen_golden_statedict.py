#!/usr/bin/env python
"""Generate tests/golden/state_dict_layout.json by constructing the REFERENCE's IPSNet (imported from
/root/reference) for its three shipped configurations and recording, per configuration,

  * the full ``state_dict()`` key -> (shape, dtype) list in the reference's order, and
  * per-tensor checksums (sum, sum of |x|) of the freshly constructed module under ``torch.manual_seed(1234)``:
    ``transf.*`` and ``output_layers.*`` are built by the reference's own code from torch's global RNG AFTER the
    encoder, so equal checksums also pin that the encoder construction here consumes the RNG stream exactly as
    the reference's does (torchvision's trunk incl. its discarded 1000-way classifier).

The encoder topology under the torchvision boundary is this repo's stand-in on both sides (torchvision is not in the
image); it is pinned independently by tests/test_resnet_topology.py against torchvision's published
parameter counts and key counts.  Data only; see tests/test_aten_path_golden.py.

    python tools/gen_golden_statedict.py
"""

import json
import os
import sys

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import torch

from ips_amd import synth
from tools.refimport import import_reference

SEED = 1234
CONFS = {
    "mnist": synth.mnist_conf(N=900, M=100, I=100, patch=50),
    "traffic": synth.traffic_conf(N=192, M=10, I=32, patch=100),
    "camelyon": synth.camelyon_conf(N=4096, M=256, I=256),
}


def main():
    ref_ips, _, _ = import_reference()
    out = {"seed": SEED, "configs": {}}
    for name, conf in CONFS.items():
        torch.manual_seed(SEED)
        net = ref_ips.IPSNet(torch.device("cpu"), conf)
        entries = []
        for k, v in net.state_dict().items():
            e = {"key": k, "shape": list(v.shape), "dtype": str(v.dtype).replace("torch.", "")}
            if v.is_floating_point():
                e["sum"] = float(v.double().sum())
                e["abs_sum"] = float(v.double().abs().sum())
            entries.append(e)
        out["configs"][name] = {"conf": conf.__dict__, "n_param": sum(p.numel() for p in net.parameters()),
                                "entries": entries}
        print(name, len(entries), "entries,", out["configs"][name]["n_param"], "parameters")
    with open(os.path.join(REPO, "tests", "golden", "state_dict_layout.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
