"""`architecture.ips_net` (the name the reference's main.py imports) resolves to ips_amd through the shim."""

import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_import_names_resolve_to_ips_amd(tmp_path):
    code = ("from architecture.ips_net import IPSNet; from architecture.transformer import Transformer, pos_enc_1d;"
            "import ips_amd.architecture as a; assert IPSNet is a.IPSNet and Transformer is a.Transformer;"
            "from utils.utils import Logger, Struct, adjust_learning_rate, shuffle_batch, shuffle_instance;"
            "from training.iterative import train_one_epoch, evaluate, init_batch, fill_batch, shrink_batch, compute_loss;"
            "import ips_amd.training.iterative as t; assert evaluate is t.evaluate; print('ok')")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(REPO, "ips_amd", "compat"), REPO]))
    out = subprocess.run([sys.executable, "-c", code], cwd=tmp_path, env=env, capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr
