"""Import the read-only reference (/root/reference) as a parity oracle.

TEST/FIXTURE INFRASTRUCTURE ONLY - nothing under ips_amd/ imports this.

The reference's architecture/ips_net.py:6 imports four names from
``torchvision.models``; torchvision is not installed in this image
(SURVEY.md section 8 c-1), so a 4-name stand-in module is registered whose
``resnet18``/``resnet50`` return this repo's own torchvision-shaped trunk
(ips_amd/architecture/resnet.py).  Everything else the reference executes is
its own code, run from where it lies; nothing is copied and no bytecode is
written into the reference tree.
"""

import importlib
import os
import sys
import types

REF_ROOT = os.environ.get("IPS_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isfile(os.path.join(REF_ROOT, "architecture", "ips_net.py"))


def _install_torchvision_stub():
    if "torchvision.models" in sys.modules and not getattr(
            sys.modules["torchvision.models"], "_ipsx_stub", False):
        return  # a real torchvision is present: use it
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if repo not in sys.path:
        sys.path.insert(0, repo)
    from ips_amd.architecture.resnet import resnet18_trunk, resnet50_trunk

    class _Weights:
        IMAGENET1K_V1 = "IMAGENET1K_V1"

    def _mk(fn):
        def ctor(weights=None, **_kw):
            if weights is not None:
                raise RuntimeError("pretrained weights cannot be fetched offline; use pretrained=False")
            return fn(3)
        return ctor

    tv = types.ModuleType("torchvision")
    models = types.ModuleType("torchvision.models")
    models._ipsx_stub = True
    models.resnet18 = _mk(resnet18_trunk)
    models.resnet50 = _mk(resnet50_trunk)
    models.ResNet18_Weights = _Weights
    models.ResNet50_Weights = _Weights
    tv.models = models
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = models


def import_reference():
    """Returns (ips_net_module, transformer_module, utils_module) of the reference."""
    if not reference_available():
        raise FileNotFoundError(REF_ROOT)
    sys.dont_write_bytecode = True
    _install_torchvision_stub()
    # the reference uses top-level package names (architecture, utils, training)
    for name in list(sys.modules):
        if name.split(".")[0] in ("architecture", "utils", "training") and \
                not (getattr(sys.modules[name], "__file__", None) or REF_ROOT).startswith(REF_ROOT):
            del sys.modules[name]
    sys.path.insert(0, REF_ROOT)
    try:
        ips_net = importlib.import_module("architecture.ips_net")
        transformer = importlib.import_module("architecture.transformer")
        utils = importlib.import_module("utils.utils")
    finally:
        sys.path.remove(REF_ROOT)
    return ips_net, transformer, utils


def import_reference_training():
    """The reference's training/iterative.py (imports its own utils.utils)."""
    import_reference()
    sys.path.insert(0, REF_ROOT)
    try:
        return importlib.import_module("training.iterative")
    finally:
        sys.path.remove(REF_ROOT)
