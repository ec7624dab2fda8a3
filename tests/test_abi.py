"""The C-ABI library loads and exports every symbol include/ipsx.h declares (no compute)."""

import ctypes
import os
import re

import pytest

from ips_amd import hip

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "ipsx.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ipsx_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for must in ("ipsx_trunk_encode", "ipsx_projector", "ipsx_logits", "ipsx_scan", "ipsx_scores",
                 "ipsx_topm", "ipsx_gather_rows", "ipsx_aggregate", "ipsx_head", "ipsx_pack_conv_weight"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    path = hip.library_path()
    assert os.path.exists(path), "libipsx.so not built - run __graft_entry__.build()"
    lib = ctypes.CDLL(path)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    assert lib.ipsx_version() // 100 == 3          # include/ipsx.h: major = incompatible signature changes


def test_python_binding_covers_the_header():
    assert sorted(hip._EXPORTS) == declared_symbols()
    hip.lib()          # argtypes/restypes attach without error


def test_hip_backend_refuses_to_run_without_library(monkeypatch, tmp_path):
    monkeypatch.setattr(hip, "_SO", str(tmp_path / "nope.so"))
    monkeypatch.setattr(hip, "_LIB", None)
    with pytest.raises(RuntimeError, match="no fallback"):
        hip.lib()


def test_header_is_plain_c_and_a_c_program_links(tmp_path):
    """include/ipsx.h through a C compiler (gcc -std=c11 -pedantic), linked against libipsx.so: what a cgo / JNI / FFI
    binding of the reference's host language would do.  Only host-side entry points are called (no GPU here)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "abi.c"
    src.write_text(r'''
#include <stdio.h>
#include "ipsx.h"
int main(void) {
    ipsx_conv cv; ipsx_trunk tr; ipsx_transf tf;
    (void)cv; (void)tr; (void)tf;
    if (ipsx_version() / 100 != IPSX_VERSION / 100) return 1;
    if (ipsx_packed_conv_weight_elems(64, 64, 3, 3) != (size_t)2 * 72 * 64 * 4) return 2;
    if (ipsx_patchify_count(1500, 1500, 32, 32, 32, 32) != 46 * 46) return 3;
    if (ipsx_patchify_count(10, 10, 32, 32, 32, 32) != 0) return 4;
    if (ipsx_folded_query_elems(8, 4, 128) != (size_t)1 * 16 * 64 * 4) return 5;
    if (ipsx_packed_stem_weight_split_bytes(64, 3) != (size_t)2 * 4 * 3 * 1024) return 6;
    if (ipsx_set_tie_order(-1) != 1) return 7;                /* query: the default is the reference's order */
    printf("abi ok %d\n", ipsx_version());
    return 0;
}
''')
    exe = tmp_path / "abi"
    libdir = os.path.dirname(hip.library_path())
    cmd = ["gcc", "-std=c11", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(root, "include"), str(src), "-o", str(exe),
           "-L", libdir, "-lipsx", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.startswith("abi ok"), (out.returncode, out.stdout, out.stderr)


def test_bn_train_workspace_bound_matches_the_binding():
    """hip.bn_train_forward/backward allocate the workspace from a constant instead of asking the library per call:
    the library's own figure never exceeds it, and unsupported channel counts are reported as such (no GPU needed)."""
    from ips_amd import hip
    L = hip.lib()
    for rows in (1, 63, 4096, 262144, 10 ** 7):
        for c in (4, 64, 128, 512, 1024):
            assert L.ipsx_bn_train_supported(rows, c) == 1
            assert 0 < L.ipsx_bn_train_workspace_floats(rows, c) <= 2 * hip._BN_MAX_SLABS * c
    for c in (0, 3, 6, 48, 2048):
        assert L.ipsx_bn_train_supported(16, c) == 0
        assert L.ipsx_bn_train_workspace_floats(16, c) == 0


def test_fused_training_path_knows_which_encoders_it_covers():
    """training/fused_encoder.supported(): BasicBlock trunks (ResNet-18, 2 or 4 stages) yes; Bottleneck trunks and the
    feature projector no - those keep the stock ops (no GPU needed: the check reads the module tree and asks the
    library about channel counts)."""
    import torch
    from ips_amd import synth
    from ips_amd.architecture import IPSNet
    from ips_amd.training import fused_encoder
    cpu = torch.device("cpu")
    assert fused_encoder.supported(IPSNet(cpu, synth.mnist_conf(N=64, M=8, I=8)).encoder)
    assert fused_encoder.supported(IPSNet(cpu, synth.traffic_conf(N=64, M=8, I=8)).encoder)
    assert not fused_encoder.supported(IPSNet(cpu, synth.traffic_conf(N=64, M=8, I=8, enc_type='resnet50', D=2048, D_k=256, D_v=256)).encoder)
    assert not fused_encoder.supported(IPSNet(cpu, synth.camelyon_conf(N=64, M=8, I=8)).encoder)
    # on the CPU the weights keep their default layout (channels-last storage is for the GPU training path only)
    w = IPSNet(cpu, synth.mnist_conf(N=64, M=8, I=8)).encoder[4][0].conv1.weight
    assert w.is_contiguous()
