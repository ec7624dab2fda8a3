"""oracle/ips_torch.py (the ATen restatement timed as cpu_baseline) against the fixtures."""

import numpy as np
import pytest
import torch

from oracle import ips_torch
from tests.util import Golden

CASES = ["mnist_mini", "mnist_ragged", "mnist_onechunk", "mnist_tok1", "traffic_tiny", "cam_b2"]


@pytest.mark.parametrize("case", CASES)
def test_torch_restatement_matches_reference_fixture(case):
    g = Golden(case)
    net = g.net("cpu")
    sd = {k: v for k, v in net.state_dict().items()}
    trace = []
    mem_patch, mem_pos, mem_idx = ips_torch.ips(sd, g.conf, g.patches(), net.pos_enc, trace)
    assert np.array_equal(torch.stack(trace, 1).numpy(), g.trace_idx)
    preds = ips_torch.forward(sd, g.conf, mem_patch, mem_pos)
    for k, v in g.preds.items():
        assert np.abs(preds[k].numpy() - v).max() < 1e-5
