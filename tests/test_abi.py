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
    assert lib.ipsx_version() // 100 == 1


def test_python_binding_covers_the_header():
    assert sorted(hip._EXPORTS) == declared_symbols()
    hip.lib()          # argtypes/restypes attach without error


def test_hip_backend_refuses_to_run_without_library(monkeypatch, tmp_path):
    monkeypatch.setattr(hip, "_SO", str(tmp_path / "nope.so"))
    monkeypatch.setattr(hip, "_LIB", None)
    with pytest.raises(RuntimeError, match="no fallback"):
        hip.lib()
