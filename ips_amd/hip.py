"""ctypes binding of libipsx.so (include/ipsx.h) for torch tensors on a ROCm device.

PyTorch is plumbing here: it owns device memory and the stream; every function
below hands raw device pointers to the C ABI, which launches the hand-written
gfx950 kernels of ips_amd/csrc on ``torch.cuda.current_stream()``.

There is NO fallback in this module: if the shared library cannot be loaded, or a
call returns an error code, a RuntimeError is raised.  ``IPSX_BACKEND=aten``
(explicit opt-out, e.g. to time stock PyTorch-ROCm ops on the same GPU) makes
``on_device`` report False so callers keep to ATen; the default is ``hip``.
"""

import collections
import ctypes as C
import os
import subprocess

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_PKG, "lib", "libipsx.so")
_LIB = None

f32p = C.c_void_p  # device pointers travel as integers
ABI_MAJOR = 3       # include/ipsx.h IPSX_VERSION / 100: the signatures in _EXPORTS below are those of this major version


def backend():
    b = os.environ.get("IPSX_BACKEND", "hip").lower()
    if b not in ("hip", "aten"):
        raise ValueError("IPSX_BACKEND must be 'hip' or 'aten', got {!r}".format(b))
    return b


def precision():
    """'fp32' (default: exact, reference parity), 'fp32x3' (every fp32 operand of the residual stages split exactly
    into three bf16 terms, six products on the bf16 matrix pipe, fp32 accumulation: fp32-grade accuracy but not
    bit-identical to the fp32 kernel) or 'bf16' (bf16 operands with fp32 accumulation; no reference behaviour
    to match).  The last two exist for the fused 1x32x32 trunk."""
    p = os.environ.get("IPSX_PRECISION", "fp32").lower()
    if p not in ("fp32", "fp32x3", "bf16"):
        raise ValueError("IPSX_PRECISION must be 'fp32', 'fp32x3' or 'bf16', got {!r}".format(p))
    return p


_TIE_MODES = {"canonical": 0, "torch": 1, "torch_all": 2}


def tie_order():
    """How equal scores are ordered (``IPSX_TIE_ORDER``).
    'torch' (default): the order torch.topk returns on CPU (the reference's), replayed on the device - in the selection
    loops only where the tie is between bit-IDENTICAL candidates (duplicated patches: they tie in the reference's own
    arithmetic too); two different candidates whose scores collide in the last bit of this arithmetic keep the canonical
    order (in the reference's arithmetic they are an ulp apart: a replay reproduces nothing of it, and costs 100+ us per
    iteration at 10,000 candidates).  ``ipsx_topm`` - scores only - replays on every tie.
    'torch_all': replay wherever scores tie (rounds 1-4).  'canonical': earlier candidate position first, never a replay."""
    t = os.environ.get("IPSX_TIE_ORDER", "torch").lower()
    if t not in _TIE_MODES:
        raise ValueError("IPSX_TIE_ORDER must be 'torch', 'torch_all' or 'canonical', got {!r}".format(t))
    return t


def set_tie_order(mode):
    """Switch the tie order at run time ('torch' | 'torch_all' | 'canonical'); returns the previous one."""
    prev = lib().ipsx_set_tie_order(_TIE_MODES[mode])
    return {v: k for k, v in _TIE_MODES.items()}[prev]


def dedup_blank():
    """Opt-in exact blank-patch deduplication in front of the fused encoder (IPSX_DEDUP_BLANK=1)."""
    return os.environ.get("IPSX_DEDUP_BLANK", "0") == "1"


# Graph replays (training/graphed.py) update parameters and BatchNorm statistics without bumping tensor._version or
# changing storage pointers, so every cache keyed on (data_ptr, _version) also carries this counter; whoever mutates
# weights behind autograd's back calls weights_changed().
_WEIGHTS_GENERATION = 0


def weights_generation():
    return _WEIGHTS_GENERATION


def weights_changed():
    """Invalidate every packed-weight cache (EncoderPlan, folded query): call after a HIP-graph replay or any other
    update of parameters / buffers that bypasses ``_version``."""
    global _WEIGHTS_GENERATION
    _WEIGHTS_GENERATION += 1


def _optimizer_stepped(optimizer, args, kwargs):
    weights_changed()


# Not every in-place update bumps ``_version``: torch's FUSED optimizers (AdamW(fused=True), ...) move the parameters
# without touching it (checked: version 0 -> 0 across a step), so every optimizer step invalidates the packed weights.
# The hook is process-global in torch (there is no per-module form), so it is installed when the first IPSNet is BUILT
# (``IPSNet.__init__`` calls ``install_optimizer_hook``), not when this module is imported: a process that merely imports
# the package keeps its optimizers untouched.  (Manual updates through ``.data`` or raw pointers still need an explicit
# ``weights_changed()``.)
_HOOK_INSTALLED = False


def install_optimizer_hook():
    global _HOOK_INSTALLED
    if _HOOK_INSTALLED:
        return True
    try:
        from torch.optim.optimizer import register_optimizer_step_post_hook
    except ImportError:      # older torch: per-optimizer hooks only; ips_amd.training registers them itself
        return False
    register_optimizer_step_post_hook(_optimizer_stepped)
    _HOOK_INSTALLED = True
    return True


def on_device(x):
    """True when ``x`` (tensor / device / str) is a GPU and the HIP backend is selected."""
    if torch.is_tensor(x):
        dev = x.device
    else:
        dev = torch.device(x)
    return dev.type == "cuda" and backend() == "hip"


# ------------------------------------------------------------------ C structs
class Conv(C.Structure):
    _fields_ = [("c_in", C.c_int), ("c_out", C.c_int), ("kh", C.c_int), ("kw", C.c_int),
                ("stride", C.c_int), ("pad", C.c_int),
                ("w_packed", C.c_void_p), ("alpha", C.c_void_p), ("shift", C.c_void_p),
                ("w_packed_bf16", C.c_void_p), ("colsum", C.c_void_p)]


class Block(C.Structure):
    _fields_ = [("n_conv", C.c_int), ("conv", Conv * 3), ("has_down", C.c_int), ("down", Conv)]


class Trunk(C.Structure):
    _fields_ = [("c_in", C.c_int), ("h", C.c_int), ("w", C.c_int), ("stem", Conv),
                ("n_block", C.c_int), ("blocks", C.POINTER(Block)), ("precision", C.c_int), ("patch_dtype", C.c_int)]


class IpsCall(C.Structure):
    """include/ipsx.h ``ipsx_ips_call``: one ips() call with a resident loop, enqueued by ONE library call."""
    _fields_ = [("b", C.c_int), ("n", C.c_int64), ("m", C.c_int), ("i", C.c_int), ("h", C.c_int), ("n_token", C.c_int),
                ("logits", C.c_void_p), ("mem_idx", C.c_void_p), ("words", C.c_void_p), ("words_total", C.c_int64),
                ("loops", C.c_int), ("scan_workspace", C.c_void_p), ("scan_workspace_bytes", C.c_size_t),
                ("trunk", C.POINTER(Trunk)), ("pos", C.c_void_p), ("quad_pulls", C.c_int),
                ("lin", C.POINTER(Conv)), ("ln_eps", C.c_float), ("short_first", C.c_int),
                ("x", C.c_void_p), ("emb", C.c_void_p), ("v_packed", C.c_void_p), ("r", C.c_int), ("workgroups", C.c_int),
                ("src", C.c_void_p), ("src_row_bytes", C.c_int64), ("src_bstride_rows", C.c_int64),
                ("pos_table", C.c_void_p), ("pos_row_bytes", C.c_int64), ("pos_bstride_rows", C.c_int64),
                ("mem_patch", C.c_void_p), ("mem_pos", C.c_void_p), ("mem_idx_out", C.c_void_p), ("status_host", C.c_void_p),
                ("timing_slot", C.c_int), ("stream", C.c_void_p), ("side_stream", C.c_void_p)]


class Transf(C.Structure):
    _fields_ = [("n_token", C.c_int), ("h", C.c_int), ("d", C.c_int), ("dk", C.c_int),
                ("dv", C.c_int), ("d_inner", C.c_int)] + \
               [(n, C.c_void_p) for n in ("q", "wq", "wk", "wv", "fc", "ln1_g", "ln1_b", "w1", "b1",
                                          "w2", "b2", "ln2_g", "ln2_b")] + \
               [("temperature", C.c_float), ("ln_eps", C.c_float)]


_EXPORTS = {
    # name: (restype, argtypes)
    "ipsx_version": (C.c_int, []),
    "ipsx_last_error": (C.c_char_p, []),
    "ipsx_device_count": (C.c_int, []),
    "ipsx_device_is_gfx950": (C.c_int, [C.c_int]),
    "ipsx_packed_conv_weight_elems": (C.c_size_t, [C.c_int] * 4),
    "ipsx_pack_conv_weight": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "ipsx_pack_conv_weight_strided": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                                C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "ipsx_pack_conv_weights_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "ipsx_packed_conv_weight_bf16_bytes": (C.c_size_t, [C.c_int] * 4),
    "ipsx_pack_conv_weight_bf16": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "ipsx_packed_conv_weight_x3_bytes": (C.c_size_t, [C.c_int] * 4),
    "ipsx_pack_conv_weight_x3": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "ipsx_packed_stem_weight_split_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "ipsx_pack_stem_weight_split": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "ipsx_bn_affine": (C.c_int, [C.c_void_p] * 5 + [C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ipsx_conv2d_affine": (C.c_int, [C.POINTER(Conv), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                     C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ipsx_conv2d_affine_to_nhwc": (C.c_int, [C.POINTER(Conv), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                             C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ipsx_conv2d_affine_nhwc": (C.c_int, [C.POINTER(Conv), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                          C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ipsx_conv2d_lds_nhwc_supported": (C.c_int, [C.c_int] * 7),
    "ipsx_conv2d_lds_nhwc": (C.c_int, [C.POINTER(Conv), C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "ipsx_conv2d_lds_nhwc_stats_slabs": (C.c_int64, [C.c_int64]),
    "ipsx_conv2d_lds_nhwc_stats": (C.c_int, [C.POINTER(Conv), C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p,
                                             C.c_void_p, C.c_void_p]),
    "ipsx_bn_train_forward_partials": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p,
                                                 C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "ipsx_conv2d_dgrad_s2_lds_nhwc_supported": (C.c_int, [C.c_int] * 7),
    "ipsx_conv2d_dgrad_s2_lds_nhwc": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "ipsx_stem7x7s2_nhwc_supported": (C.c_int, [C.c_int] * 8),
    "ipsx_stem7x7s2_nhwc": (C.c_int, [C.POINTER(Conv), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ipsx_conv2d_wgrad_nhwc_supported": (C.c_int, [C.c_int] * 6),
    "ipsx_conv2d_wgrad_nhwc_workspace_bytes": (C.c_size_t, [C.c_int64] + [C.c_int] * 4),
    "ipsx_conv2d_wgrad_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64] + [C.c_int] * 8 + [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ipsx_maxpool_3x3s2_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ipsx_maxpool_3x3s2_bwd_nhwc_supported": (C.c_int, [C.c_int] * 3),
    "ipsx_maxpool_3x3s2_bwd_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ipsx_avgpool_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "ipsx_maxpool_3x3s2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "ipsx_avgpool": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "ipsx_trunk_workspace_bytes": (C.c_size_t, [C.POINTER(Trunk), C.c_int64]),
    "ipsx_trunk_kernel": (C.c_char_p, [C.POINTER(Trunk)]),
    "ipsx_trunk_encode": (C.c_int, [C.POINTER(Trunk), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                    C.c_size_t, C.c_void_p]),
    "ipsx_trunk_dedup_workspace_bytes": (C.c_size_t, [C.POINTER(Trunk), C.c_int64]),
    "ipsx_trunk_encode_dedup": (C.c_int, [C.POINTER(Trunk), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                          C.c_size_t, C.c_void_p, C.c_void_p]),
    "ipsx_trunk_encode_dedup_flagged": (C.c_int, [C.POINTER(Trunk), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "ipsx_patchify_count": (C.c_int64, [C.c_int] * 6),
    "ipsx_patchify": (C.c_int, [C.c_void_p] + [C.c_int] * 8 + [C.c_void_p, C.c_void_p]),
    "ipsx_patchify_sparse": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64] + [C.c_int] * 8 +
                             [C.c_void_p, C.c_void_p, C.c_void_p]),
    "ipsx_projector": (C.c_int, [C.POINTER(Conv), C.c_void_p, C.c_int64, C.c_float, C.c_void_p,
                                 C.c_void_p, C.c_size_t, C.c_void_p]),
    "ipsx_projector_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "ipsx_weight_colsum": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "ipsx_projector_stats": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "ipsx_projector_apply": (C.c_int, [C.POINTER(Conv), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ipsx_projector_apply_publish": (C.c_int, [C.POINTER(Conv), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_int32, C.c_void_p]),
    "ipsx_trunk_stream_ctl_words": (C.c_size_t, [C.c_int64]),
    "ipsx_trunk_stream_supported": (C.c_int, [C.POINTER(Trunk), C.c_int, C.c_int]),
    "ipsx_trunk_stream": (C.c_int, [C.POINTER(Trunk), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ipsx_projector_stream_ctl_words": (C.c_size_t, [C.c_int64]),
    "ipsx_projector_stream_ctl_zero_words": (C.c_size_t, [C.c_int64]),
    "ipsx_projector_stream_supported": (C.c_int, [C.POINTER(Conv), C.c_int64, C.c_int]),
    "ipsx_projector_stream": (C.c_int, [C.POINTER(Conv), C.c_void_p, C.c_int64, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "ipsx_logits_stats": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                    C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int64,
                                    C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "ipsx_query_proj": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.c_int,
                                  C.c_void_p, C.c_void_p]),
    "ipsx_folded_query_elems": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "ipsx_fold_query": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "ipsx_logits": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                              C.c_int, C.c_int64, C.c_int, C.c_int,
                              C.c_void_p, C.c_int64, C.c_void_p]),
    "ipsx_folded_query_bf16_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "ipsx_fold_query_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "ipsx_logits_bf16": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                   C.c_int, C.c_int64, C.c_int, C.c_int,
                                   C.c_void_p, C.c_int64, C.c_void_p]),
    "ipsx_set_tie_order": (C.c_int, [C.c_int]),
    "ipsx_scan": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ipsx_scan_workspace_bytes": (C.c_size_t, [C.c_int] * 5),
    "ipsx_scan_workgroups_per_image": (C.c_int, [C.c_int] * 5),
    "ipsx_scan_range": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ipsx_scan_range_if": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "ipsx_scan_persistent_supported": (C.c_int, [C.c_int] * 4),
    "ipsx_scan_persistent_groupable": (C.c_int, [C.c_int] * 4),
    "ipsx_scan_persistent_on": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int,
                                          C.c_void_p]),
    "ipsx_scan_persistent_ws": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int,
                                          C.c_void_p, C.c_size_t, C.c_void_p]),
    "ipsx_scan_range_if_ws": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t,
                                        C.c_void_p]),
    "ipsx_scan_persistent": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "ipsx_publish_rows": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "ipsx_set_persistent_wait_ms": (C.c_int, [C.c_int]),
    "ipsx_scan_gate": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ipsx_trunk_encode_indexed": (C.c_int, [C.POINTER(Trunk), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "ipsx_scores_workspace_bytes": (C.c_size_t, [C.c_int] * 5),
    "ipsx_scores": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 6 +
                    [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ipsx_topm": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "ipsx_topm_workspace_bytes": (C.c_size_t, [C.c_int] * 3),
    "ipsx_gather_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int,
                                   C.c_int64, C.c_int64, C.c_void_p]),
    "ipsx_ips_call_run": (C.c_int, [C.POINTER(IpsCall)]),
    "ipsx_ips_call_elapsed": (C.c_int, [C.c_int, C.POINTER(C.c_float)]),
    "ipsx_ips_finish": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int,
                                  C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ipsx_aggregate_workspace_bytes": (C.c_size_t, [C.POINTER(Transf), C.c_int, C.c_int]),
    "ipsx_aggregate": (C.c_int, [C.POINTER(Transf), C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                 C.c_void_p, C.c_size_t, C.c_void_p]),
    "ipsx_aggregate_packed": (C.c_int, [C.POINTER(Transf), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_size_t, C.c_void_p]),
    "ipsx_head": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                            C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "ipsx_bn_train_supported": (C.c_int, [C.c_int64, C.c_int]),
    "ipsx_bn_train_workspace_floats": (C.c_size_t, [C.c_int64, C.c_int]),
    "ipsx_bn_train_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ipsx_bn_train_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p]),
}


def library_path():
    return _SO


def build(verbose=False):
    """Compile ips_amd/csrc for gfx950 into ips_amd/lib/libipsx.so (in-tree)."""
    cmd = ["make", "-C", os.path.join(_PKG, "csrc")] + ([] if verbose else ["-s"])
    subprocess.check_call(cmd)
    return _SO


def lib():
    """The loaded library.  Raises (never falls back) when it is missing."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(_SO):
            raise RuntimeError(
                "libipsx.so not found at {} - build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C ips_amd/csrc`; the HIP backend has no fallback".format(_SO))
        L = C.CDLL(_SO)
        for name, (res, args) in _EXPORTS.items():
            fn = getattr(L, name)          # AttributeError here = header/library mismatch
            fn.restype, fn.argtypes = res, args
        if L.ipsx_version() // 100 != ABI_MAJOR:
            raise RuntimeError("libipsx.so ABI version {} != {}.x (include/ipsx.h: the major number changes with every "
                               "incompatible change of an exported signature)".format(L.ipsx_version(), ABI_MAJOR))
        L.ipsx_set_tie_order(_TIE_MODES[tie_order()])
        if os.environ.get("IPSX_CAM_HEAD"):        # diagnostic: units a lone slide's projector stream hands out as column quarters first
            L.ipsx_dbg_stream_head.restype, L.ipsx_dbg_stream_head.argtypes = None, [C.c_int]
            L.ipsx_dbg_stream_head(int(os.environ["IPSX_CAM_HEAD"]))
        if os.environ.get("IPSX_SCAN_R8", "1") == "0":      # diagnostic: the 8-row loop shapes through scan_fast_kernel
            L.ipsx_dbg_scan_r8.argtypes = [C.c_int]
            L.ipsx_dbg_scan_r8(0)
        _LIB = L
    return _LIB


def _ck(rc, what):
    if rc != 0:
        raise RuntimeError("{} failed ({}): {}".format(what, rc, lib().ipsx_last_error().decode()))


def _stream():
    """The current stream's handle.  (The raw getter: ``torch.cuda.current_stream()`` builds a Stream object through three
    Python layers - ~4 us, ten times per ips() call, most of them between the first and the last launch of the call.)"""
    return C.c_void_p(_raw_stream(_cur_device()))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)
if _raw_stream is None or _cur_device is None:          # (another torch build: the documented way)
    def _stream():                                      # noqa: F811
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _f32(t):
    if t.dtype != torch.float32:
        raise TypeError("expected float32, got {}".format(t.dtype))
    return t if t.is_contiguous() else t.contiguous()


_PATCH_DTYPES = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


def _patches(t):
    """Patch tensors may be stored in bfloat16 / float16 (BASELINE configs[4]); only the reduced-precision fused trunks
    read those (the exact path's contract is float32 in)."""
    if t.dtype not in _PATCH_DTYPES:
        raise TypeError("patches must be float32, bfloat16 or float16, got {}".format(t.dtype))
    if t.dtype != torch.float32 and precision() == "fp32":
        raise TypeError("{} patch storage needs IPSX_PRECISION=bf16 or fp32x3 (the exact trunk reads float32)".format(t.dtype))
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------ scorer
def query_proj(q, wq, temperature):
    """(T, D) queries, (H*Dk, D) weight -> (T, H*Dk) = (q @ wq.T) / temperature."""
    q, wq = _f32(q.detach()), _f32(wq.detach())
    out = torch.empty((q.shape[0], wq.shape[0]), dtype=torch.float32, device=q.device)
    _ck(lib().ipsx_query_proj(_p(q), _p(wq), C.c_float(temperature), q.shape[0], q.shape[1],
                              wq.shape[0], _p(out), _stream()), "ipsx_query_proj")
    return out


def pack_linear(weight):
    """nn.Linear weight (out, in) -> MFMA B-operand stream (a 1x1 convolution)."""
    w = weight.detach()
    return _pack_conv(w.reshape(w.shape[0], w.shape[1], 1, 1))


def fold_query(qs, wk_weight, H, Dk, T):
    """Scaled query (T, H*Dk) folded into the key weights k_w.weight (H*Dk, D) -> packed (H*T, D) operand of
    ``logits`` (ipsx_fold_query): logit = x . V[h*T + t],  V[h*T + t] = sum_j qs[t][h, j] k_w[h*Dk + j]."""
    D = wk_weight.shape[1]
    out = torch.empty(lib().ipsx_folded_query_elems(H, T, D), dtype=torch.float32, device=qs.device)
    _ck(lib().ipsx_fold_query(_p(_f32(qs)), _p(pack_linear(wk_weight)), H, Dk, T, D, _p(out), _stream()), "ipsx_fold_query")
    return out


def fold_query_bf16(qs, wk_weight, H, Dk, T):
    """The folded query rounded to bfloat16 in the operand layout of ``ipsx_logits_bf16`` (a uint8 tensor: that dtype is
    how ``logits`` tells the two apart)."""
    D = wk_weight.shape[1]
    out = torch.empty(lib().ipsx_folded_query_bf16_bytes(H, T, D), dtype=torch.uint8, device=qs.device)
    _ck(lib().ipsx_fold_query_bf16(_p(_f32(qs)), _p(pack_linear(wk_weight)), H, Dk, T, D, _p(out), _stream()),
        "ipsx_fold_query_bf16")
    return out


def logits(emb, pos, vq, R, out=None):
    """Per-patch attention logits (B, n, R = H*T) from the folded query ``vq``; ``out`` may be a column slice of
    (B, N, R).  A bfloat16 folded query (``fold_query_bf16``) selects the bf16 matrix pipe."""
    B, n, D = emb.shape
    emb = _f32(emb)
    if out is None:
        out = torch.empty((B, n, R), dtype=torch.float32, device=emb.device)
    if out.stride(2) != 1 or out.stride(1) != R:
        raise ValueError("logits output must be row-contiguous")
    pos_bs = 0
    if pos is not None:
        if pos.stride(2) != 1 or pos.stride(1) != D:
            pos = pos.contiguous()
        pos_bs = pos.stride(0) if pos.shape[0] > 1 else 0
    if vq.dtype == torch.uint8:
        _ck(lib().ipsx_logits_bf16(_p(emb), n * D, _p(pos), pos_bs, _p(vq), B, n, D, R, _p(out), out.stride(0), _stream()),
            "ipsx_logits_bf16")
        return out
    _ck(lib().ipsx_logits(_p(emb), n * D, _p(pos), pos_bs, _p(vq), B, n, D, R, _p(out), out.stride(0), _stream()),
        "ipsx_logits")
    return out


def logits_stats(emb, pos, vq, R, out, stats_x, stats_out, ln_eps):
    """``logits(emb, pos, vq, R, out)`` and, in the same launch, the LayerNorm row moments of ``stats_x`` (P, F) into
    ``stats_out`` (P, 2).  fp32 folded query only."""
    B, n, D = emb.shape
    emb = _f32(emb)
    if out.stride(2) != 1 or out.stride(1) != R:
        raise ValueError("logits output must be row-contiguous")
    pos_bs = 0
    if pos is not None:
        if pos.stride(2) != 1 or pos.stride(1) != D:
            pos = pos.contiguous()
        pos_bs = pos.stride(0) if pos.shape[0] > 1 else 0
    stats_x = _f32(stats_x)
    sn, sf = stats_x.shape
    if tuple(stats_out.shape) != (sn, 2) or stats_out.dtype != torch.float32 or not stats_out.is_contiguous():
        raise ValueError("stats_out must be a contiguous (P, 2) float32 tensor")
    _ck(lib().ipsx_logits_stats(_p(emb), n * D, _p(pos), pos_bs, _p(vq), B, n, D, R, _p(out), out.stride(0),
                                _p(stats_x), sn, sf, C.c_float(ln_eps), _p(stats_out), _stream()), "ipsx_logits_stats")
    return out


def scan_workspace(B, M, I, H, T, device):
    """Device workspace of the selection loop for candidate sets beyond the LDS (M + I above ~4,000: the reference's
    shipped CAMELYON configuration is M = I = 5000), or ``None`` when the shape needs none.  A caller that runs the loop on
    a side stream keeps it between calls (like its other per-call buffers) instead of allocating it there."""
    nb = lib().ipsx_scan_workspace_bytes(B, M, I, H, T)
    return torch.empty(nb, dtype=torch.uint8, device=device) if nb else None


def scan_workgroups_per_image(B, M, I, H, T):
    """Compute units the loop of ONE image occupies in a call of B images (1, or the team of csrc/scan_large_team.h)."""
    return max(1, int(lib().ipsx_scan_workgroups_per_image(B, M, I, H, T)))


def scan(lg, M, I, H, T, want_scores=False):
    """The IPS chunk loop on cached logits (B, N, H*T) -> mem_idx (B, M) int64."""
    lg = _f32(lg)
    B, N = lg.shape[:2]
    mem_idx = torch.empty((B, M), dtype=torch.int64, device=lg.device)
    sc = torch.empty((B, M), dtype=torch.float32, device=lg.device) if want_scores else None
    tie = torch.zeros((B,), dtype=torch.int32, device=lg.device)
    ws = scan_workspace(B, M, I, H, T, lg.device)
    _ck(lib().ipsx_scan(_p(lg), B, N, M, I, H, T, _p(mem_idx), _p(sc), _p(tie), _p(ws), ws.numel() if ws is not None else 0,
                        _stream()), "ipsx_scan")
    scan.last_tie = tie
    return (mem_idx, sc) if want_scores else mem_idx


def scan_range(lg, M, I, H, T, it_begin, it_end, mem_idx, tie, workspace=None):
    """Iterations [it_begin, it_end) of the loop, state carried in ``mem_idx`` (B, M) int64 / ``tie`` (B,) int32.
    ``workspace``: ``scan_workspace(...)`` kept by the caller; allocated here (on the current stream) when missing."""
    B, N = lg.shape[:2]
    if not lg.is_contiguous():
        raise ValueError("scan_range needs the full contiguous (B, N, H*T) logits buffer")
    ws = workspace if workspace is not None else scan_workspace(B, M, I, H, T, lg.device)
    _ck(lib().ipsx_scan_range(_p(lg), B, N, M, I, H, T, it_begin, it_end, _p(mem_idx), None, _p(tie),
                              _p(ws), ws.numel() if ws is not None else 0, _stream()), "ipsx_scan_range")
    return mem_idx


def scan_range_if(lg, M, I, H, T, it_begin, it_end, mem_idx, tie, cond, mask=1, workspace=None):
    """``scan_range`` that every workgroup abandons at once unless ``cond`` (int32 device scalar) has a bit of ``mask``
    set: the in-call recovery of a persistent loop that gave up waiting (its status word, bit 0)."""
    B, N = lg.shape[:2]
    if workspace is not None:
        _ck(lib().ipsx_scan_range_if_ws(_p(lg), B, N, M, I, H, T, it_begin, it_end, _p(mem_idx), None, _p(tie), _p(cond), mask,
                                        _p(workspace), workspace.numel(), _stream()), "ipsx_scan_range_if_ws")
        return mem_idx
    _ck(lib().ipsx_scan_range_if(_p(lg), B, N, M, I, H, T, it_begin, it_end, _p(mem_idx), None, _p(tie), _p(cond), mask,
                                 _stream()), "ipsx_scan_range_if")
    return mem_idx


Geometry = collections.namedtuple("Geometry", "cus xcds cus_per_xcd")
_GEOMETRY = {}


def device_geometry(dev):
    """(compute units, XCDs, compute units per XCD) of a device as this process sees it.  gfx950 / gfx942 chiplets have 32
    units each - SPX 256 units = 8 XCDs, DPX / QPX / CPX partitions 4 / 2 / 1 - and deal workgroups to their XCDs
    round-robin; anything else counts as one XCD.  Launch sizes of the selection pipelines are derived from this."""
    dev = torch.device(dev)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    g = _GEOMETRY.get(idx)
    if g is None:
        props = torch.cuda.get_device_properties(idx)
        cus = int(props.multi_processor_count)
        arch = str(getattr(props, "gcnArchName", ""))
        xcds = cus // 32 if (arch.startswith("gfx95") or arch.startswith("gfx94")) and cus % 32 == 0 and cus >= 32 else 1
        g = _GEOMETRY[idx] = Geometry(cus, xcds, cus // xcds)
    return g


# Persistent kernels (a selection loop that waits for rows its producers publish while both run) need kernels of
# different streams to actually run side by side.  Under rocprofv3's counter collection, thread traces or a serialising
# debug switch they do not - the loop could only time out.  Instead of guessing from environment variables this is
# established ONCE per device by doing it: a four-iteration loop is launched before its rows are published.  And it is
# revoked for the whole process when a loop ever times out in production (IPSNet mirrors the status word to the host).
_PERSIST_OK = {}
_PERSIST_OFF = None


def persistent_wait_ms(ms=0):
    """Longest wait of a persistent loop (and its gate) without any progress, in ms (default 50); ms > 0 sets it.
    ``IPSX_PERSIST_WAIT_MS`` sets it once per process, when the first device is asked about (``persistent_ok``)."""
    return lib().ipsx_set_persistent_wait_ms(int(ms))


_PERSIST_STRIKES = 0
_PERSIST_WAIT_SET = False


def persistent_ok(dev):
    global _PERSIST_WAIT_SET
    if _PERSIST_OFF is not None:
        return False
    if not _PERSIST_WAIT_SET:
        _PERSIST_WAIT_SET = True
        ms = int(os.environ.get("IPSX_PERSIST_WAIT_MS", "0") or 0)
        if ms > 0:
            persistent_wait_ms(ms)
    dev = torch.device(dev)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    ok = _PERSIST_OK.get(idx)
    if ok is None:
        ok = _PERSIST_OK[idx] = _persistent_selftest(torch.device("cuda", idx))
    return ok


def persistent_disable(reason):
    """Switch the persistent pipelines off for the rest of the process (per-part launches take over) and say so once."""
    global _PERSIST_OFF
    if _PERSIST_OFF is None:
        import warnings
        _PERSIST_OFF = reason
        warnings.warn("ips_amd: persistent selection loops are switched off for this process: " + reason +
                      " (IPSX_SCAN_PERSIST=0 avoids the attempt)")


_PERSIST_CALLS = 0            # calls that launched a resident loop (Selection.persistent_begin counts them)
_PERSIST_RECENT = []          # _PERSIST_CALLS at the time of every recent timeout


def persistent_timed_out(dev):
    """A persistent loop gave up waiting (its call was redone by the conditional launch: results valid).  ONE such event
    on healthy hardware is a stalled host between the loop's launch and its producer's - the soak run
    (tools/soak.py, profiles/r05_soak.txt) saw one in ~5,000 calls while the cyclic garbage collector could land there
    (40-60 ms per full collection; it is held off across those lines now) and one in ~19,000 since: it does not cost a
    long-running job its pipelines.  The device's self-test is run again (a device synchronisation, once per event), and
    the pipelines stay on while it passes and fewer than ``IPSX_PERSIST_STRIKES`` (default 3) loops have timed out within
    the last ``IPSX_PERSIST_WINDOW`` (default 1,000) calls that launched one.  -> True when they stay on."""
    global _PERSIST_STRIKES
    import warnings
    _PERSIST_STRIKES += 1
    limit = max(1, int(os.environ.get("IPSX_PERSIST_STRIKES", "3") or 3))
    window = max(1, int(os.environ.get("IPSX_PERSIST_WINDOW", "1000") or 1000))
    _PERSIST_RECENT[:] = [c for c in _PERSIST_RECENT if _PERSIST_CALLS - c < window] + [_PERSIST_CALLS]
    what = "a persistent selection loop timed out waiting for rows (its call was redone with per-part launches: results valid)"
    if len(_PERSIST_RECENT) >= limit:
        persistent_disable("%s - %d such events within %d calls" % (what, len(_PERSIST_RECENT), window))
        return False
    dev = torch.device(dev)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    ok = _PERSIST_OK[idx] = _persistent_selftest(torch.device("cuda", idx))
    if not ok:
        persistent_disable(what + "; a loop beside its producer no longer passes the device's self-test")
        return False
    warnings.warn("ips_amd: %s; the device's self-test passes, the persistent pipelines stay on (event %d of %d allowed within "
                  "%d calls; IPSX_PERSIST_WAIT_MS raises the wait, IPSX_PERSIST_STRIKES the allowance)"
                  % (what, len(_PERSIST_RECENT), limit, window))
    return True


def _persistent_selftest(dev):
    M = I = 16
    H, T = 8, 1
    if not scan_persistent_supported(M, I, H, T):
        return False
    N = M + 4 * I
    with torch.cuda.device(dev):
        lg = torch.zeros((1, N, H * T), dtype=torch.float32, device=dev)
        mem = torch.empty((1, M), dtype=torch.int64, device=dev)
        words = torch.zeros((3,), dtype=torch.int32, device=dev)            # tie | progress | status
        side, main = side_stream(dev), torch.cuda.current_stream(dev)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            scan_persistent(lg, M, I, H, T, mem, words[0:1], words[1:2], words[2:3])
        scan_gate(words[2:3])
        publish_rows(words[1:2], N)
        main.wait_stream(side)
        torch.cuda.synchronize(dev)
        status = int(words[2].item())
    return (status & 1) == 0 and (status & 2) == 2


_SIDE_STREAMS = {}


def side_stream(dev):
    """THE side stream of a device (high priority: the few workgroups of a selection loop must not queue behind an encoder
    grid), shared by every net of the process: the runtime maps streams onto a handful of hardware queues in creation
    order, and a net whose fresh side stream lands on its main stream's queue runs loop and encoder one after the other
    (measured: the CAMELYON M = I = 5000 leg of bench.py 12 % slower when a leg before it had created streams)."""
    dev = torch.device(dev)
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev, priority=-1)
    return st


def scan_persistent_supported(M, I, H, T):
    return bool(lib().ipsx_scan_persistent_supported(M, I, H, T))


def scan_persistent_large(M, I, H, T):
    """Does this shape run its persistent loop on ``scan_large_kernel`` (a candidate set beyond the LDS: the caller hands
    ``scan_workspace`` to ``scan_persistent`` / ``scan_range_if``)?"""
    return (not scan_persistent_supported(M, I, H, T) and lib().ipsx_scan_workspace_bytes(1, M, I, H, T) > 0
            and M + I <= 16384 and H * T <= 256)


def scan_persistent_groupable(M, I, H, T):
    """Can fewer resident workgroups than images run this shape's persistent loops (``scan_persistent(workgroups=)``)?"""
    return bool(lib().ipsx_scan_persistent_groupable(M, I, H, T))


def scan_persistent(lg, M, I, H, T, mem_idx, tie, ready, status, workgroups=0, workspace=None, stream=None):
    """The whole loop as one launch on the CURRENT stream that waits for ``ready`` (int32 device scalar, advanced with
    ``publish_rows`` on the producing stream; or one word per image - ``ready.numel() == B`` > 1 - when the producer works
    through the images one after the other) before it reads rows; see include/ipsx.h.  ``workgroups`` in (0, B): that
    many resident loops, each taking its images one after the other (``scan_persistent_groupable`` shapes)."""
    B, N = lg.shape[:2]
    if not lg.is_contiguous():
        raise ValueError("scan_persistent needs the full contiguous (B, N, H*T) logits buffer")
    if ready.numel() not in (1, B):
        raise ValueError("ready: one word, or one per image")
    st = C.c_void_p(stream.cuda_stream) if stream is not None else _stream()     # (``stream``: launch there without
    if workspace is not None:          # a candidate set beyond the LDS             #  making it the current stream)
        _ck(lib().ipsx_scan_persistent_ws(_p(lg), B, N, M, I, H, T, _p(mem_idx), None, _p(tie), _p(ready),
                                          1 if (ready.numel() == B and B > 1) else 0, _p(status), int(workgroups),
                                          _p(workspace), workspace.numel(), st), "ipsx_scan_persistent_ws")
        return mem_idx
    _ck(lib().ipsx_scan_persistent_on(_p(lg), B, N, M, I, H, T, _p(mem_idx), None, _p(tie), _p(ready),
                                      1 if (ready.numel() == B and B > 1) else 0, _p(status), int(workgroups), st),
        "ipsx_scan_persistent_on")
    return mem_idx


def scan_gate(status):
    """Hold the current stream until the persistent scan owning ``status`` is resident (bounded wait)."""
    _ck(lib().ipsx_scan_gate(_p(status), _stream()), "ipsx_scan_gate")


def publish_rows(ready, n_rows):
    _ck(lib().ipsx_publish_rows(_p(ready), int(n_rows), _stream()), "ipsx_publish_rows")


def _scores_impl(x, qs, wk, H, Dk, T, want_attn):
    x = _f32(x)
    B, L, D = x.shape
    sc = torch.empty((B, L), dtype=torch.float32, device=x.device)
    attn = torch.empty((B, H, T, L), dtype=torch.float32, device=x.device) if want_attn else None
    nb = lib().ipsx_scores_workspace_bytes(B, L, D, H, T)
    ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=x.device)
    wkp = pack_linear(wk)
    _ck(lib().ipsx_scores(_p(x), _p(wkp), _p(qs), B, L, D, H, Dk, T, _p(sc), _p(attn),
                          _p(ws), nb, _stream()), "ipsx_scores")
    return sc, attn


def scores(x, qs, wk, H, Dk, T):
    return _scores_impl(x, qs, wk, H, Dk, T, False)[0]


def attn_map(x, qs, wk, H, Dk, T):
    return _scores_impl(x, qs, wk, H, Dk, T, True)[1]


def topm(sc, M):
    sc = _f32(sc)
    B, L = sc.shape
    top = torch.empty((B, M), dtype=torch.int64, device=sc.device)
    tie = torch.zeros((B,), dtype=torch.int32, device=sc.device)
    nb = lib().ipsx_topm_workspace_bytes(B, L, M)
    ws = torch.empty(nb, dtype=torch.uint8, device=sc.device) if nb else None
    _ck(lib().ipsx_topm(_p(sc), B, L, M, _p(top), _p(tie), _p(ws), nb, _stream()), "ipsx_topm")
    return top


def gather_rows(src, idx):
    """src (B|1, N, ...) on the GPU, idx (B, M) int64 -> (B, M, ...)."""
    B, M = idx.shape
    if src.stride(0) == 0 and src.shape[0] > 1:      # expanded broadcast table
        src = src[:1]
    src = src if src.is_contiguous() else src.contiguous()
    idx = idx if idx.is_contiguous() else idx.contiguous()
    N = src.shape[1]
    row_bytes = src[0, 0].numel() * src.element_size()
    if row_bytes % 4:
        raise ValueError("row size must be a multiple of 4 bytes")
    out = torch.empty((B, M) + tuple(src.shape[2:]), dtype=src.dtype, device=src.device)
    bstride = N if src.shape[0] > 1 else 0
    _ck(lib().ipsx_gather_rows(_p(src), _p(idx), _p(out), B, N, M, row_bytes, bstride, _stream()),
        "ipsx_gather_rows")
    return out


def ips_finish_supported(src, pos):
    """Can ``ips_finish`` end the call: device tensors whose rows are whole 16-byte units at 16-byte addresses?"""
    if not (on_device(src) and src.is_contiguous() and src.dim() >= 3):
        return False
    if (src[0, 0].numel() * src.element_size()) % 16 or src.data_ptr() % 16:
        return False
    if pos is None:
        return True
    if not on_device(pos) or pos.dim() != 3:
        return False
    p1 = pos[:1] if pos.stride(0) == 0 else pos
    return p1.is_contiguous() and (pos.shape[2] * pos.element_size()) % 16 == 0 and pos.data_ptr() % 16 == 0


def ips_finish(src, pos, idx_buf, status, status_host):
    """The end of an ``ips()`` call whose loop ran resident, ONE launch: ``src[b, idx[b, m]]``, ``pos[b, idx[b, m]]`` (or None),
    a fresh copy of the loop's index buffer, the loop's status word to its pinned host mirror (``ipsx_ips_finish``).
    -> (mem_idx, mem_patch, mem_pos)"""
    B, M = idx_buf.shape
    N = src.shape[1]
    row_bytes = src[0, 0].numel() * src.element_size()
    out = torch.empty((B, M) + tuple(src.shape[2:]), dtype=src.dtype, device=src.device)
    idx = torch.empty_like(idx_buf)
    out_pos = None
    pos_bytes = pos_bs = 0
    if pos is not None:
        pos_bs = 0 if (pos.stride(0) == 0 or pos.shape[0] == 1) else pos.shape[1]
        pos_bytes = pos.shape[2] * pos.element_size()
        out_pos = torch.empty((B, M, pos.shape[2]), dtype=pos.dtype, device=pos.device)
    _ck(lib().ipsx_ips_finish(_p(src), row_bytes, N if src.shape[0] > 1 else 0, N, _p(pos), pos_bytes, pos_bs, _p(idx_buf), B, M,
                              _p(out), _p(out_pos), _p(idx), _p(status), _p(status_host), _stream()), "ipsx_ips_finish")
    return idx, out, out_pos


# ------------------------------------------------------------------ image -> patches
def patchify(img, patch_size, patch_stride):
    """img (B, C, H, W) float32 on the GPU -> (B, N, C, ph, pw): unfold + permute + reshape of the reference's
    datasets (data/megapixel_mnist/mnist_dataset.py:44-51, data/traffic/traffic_dataset.py:336-343), batched."""
    img = _f32(img)
    B, Cc, H, W = img.shape
    (ph, pw), (sh, sw) = patch_size, patch_stride
    n = lib().ipsx_patchify_count(H, W, ph, pw, sh, sw)
    if n <= 0:
        raise ValueError("patch {}x{} / stride {}x{} does not fit a {}x{} image".format(ph, pw, sh, sw, H, W))
    out = torch.empty((B, n, Cc, ph, pw), dtype=torch.float32, device=img.device)
    _ck(lib().ipsx_patchify(_p(img), B, Cc, H, W, ph, pw, sh, sw, _p(out), _stream()), "ipsx_patchify")
    return out


def patchify_sparse(index, value, offsets, canvas, patch_size, patch_stride, flags=False):
    """Sparse (H, W, C) canvases -> (B, N, C, ph, pw) patches on the GPU (mnist_dataset.py:34-51).

    ``index`` (nnz int64 flat positions), ``value`` (nnz float32), ``offsets`` (B+1 int64, image i owns
    [offsets[i], offsets[i+1])) - all on the GPU; ``canvas`` = (H, W, C).  With ``flags`` also returns the
    per-patch int32 non-blank flags (B*N)."""
    H, W, Cc = canvas
    (ph, pw), (sh, sw) = patch_size, patch_stride
    B = offsets.numel() - 1
    if index.dtype != torch.int64 or offsets.dtype != torch.int64 or value.dtype != torch.float32:
        raise TypeError("index / offsets must be int64 and value float32")
    n = lib().ipsx_patchify_count(H, W, ph, pw, sh, sw)
    if n <= 0 or B <= 0:
        raise ValueError("bad geometry")
    index, value, offsets = index.contiguous(), value.contiguous(), offsets.contiguous()
    out = torch.empty((B, n, Cc, ph, pw), dtype=torch.float32, device=offsets.device)
    nb = torch.empty((B * n,), dtype=torch.int32, device=offsets.device) if flags else None
    _ck(lib().ipsx_patchify_sparse(_p(index), _p(value), _p(offsets), index.numel(), B, Cc, H, W, ph, pw, sh, sw,
                                   _p(out), _p(nb), _stream()), "ipsx_patchify_sparse")
    return (out, nb) if flags else out


# ------------------------------------------------------------------ aggregation
def aggregate(transf, x):
    """Transformer.forward in eval / no-grad mode: x (B, M, D) -> (B, n_token, D)."""
    ca, mlp = transf.crs_attn, transf.mlp
    x = _f32(x)
    B, M, D = x.shape
    t = Transf()
    t.n_token, t.h, t.d, t.dk, t.dv, t.d_inner = ca.n_token, ca.H, D, ca.D_k, ca.D_v, mlp.w_1.out_features
    keep = []
    for name, src in (("q", ca.q[0]), ("wq", ca.q_w.weight), ("wk", ca.k_w.weight), ("wv", ca.v_w.weight),
                      ("fc", ca.fc.weight), ("ln1_g", ca.layer_norm.weight), ("ln1_b", ca.layer_norm.bias),
                      ("w1", mlp.w_1.weight), ("b1", mlp.w_1.bias), ("w2", mlp.w_2.weight),
                      ("b2", mlp.w_2.bias), ("ln2_g", mlp.layer_norm.weight), ("ln2_b", mlp.layer_norm.bias)):
        s = _f32(src.detach())
        keep.append(s)
        setattr(t, name, s.data_ptr())
    t.temperature = float(ca.attention.temperature)
    t.ln_eps = float(ca.layer_norm.eps)
    out = torch.empty((B, ca.n_token, D), dtype=torch.float32, device=x.device)
    nb = lib().ipsx_aggregate_workspace_bytes(C.byref(t), B, M)
    ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=x.device)
    # the folded query and the packed V weights depend on the parameters only: kept by the attention module between calls
    vq = ca.folded_query() if hasattr(ca, "folded_query") else None
    if vq is not None and vq.dtype != torch.float32:       # (bf16 logits operand: the aggregation folds its own fp32 one)
        vq = None
    wvp = ca.packed_v() if hasattr(ca, "packed_v") else None
    _ck(lib().ipsx_aggregate_packed(C.byref(t), _p(vq), _p(wvp), _p(x), B, M, _p(out), _p(ws), nb, _stream()),
        "ipsx_aggregate_packed")
    return out


def head(emb, token, linear, act):
    """act(Linear(emb[:, token])) with act in {'softmax', 'sigmoid'}: (B, T, D) -> (B, n_class)."""
    emb = _f32(emb)
    B, T, D = emb.shape
    w, b = _f32(linear.weight.detach()), _f32(linear.bias.detach())
    out = torch.empty((B, w.shape[0]), dtype=torch.float32, device=emb.device)
    _ck(lib().ipsx_head(_p(emb), B, T, D, token, _p(w), _p(b), w.shape[0],
                        {"softmax": 0, "sigmoid": 1}[act], _p(out), _stream()), "ipsx_head")
    return out


# ------------------------------------------------------------------ the encoder plan and the training step's kernels
# (modules of their own since round 6; every name stays reachable as ``hip.<name>``)
from .hip_encoder import (EncoderPlan, _PlanHold, _bn_affine, _pack_conv, encoder_kernel_name)          # noqa: E402,F401
from .hip_train import (PackJob, _BN_MAX_SLABS, _BN_WS_FLOATS, _CL, _PACK_BATCH_MAX, _WGRAD_MAX_BYTES, _bn_workspace_floats, _pack_conv_view, _rows_cl, bn_train_backward, bn_train_forward, bn_train_forward_partials, bn_train_supported, conv2d_nhwc, conv2d_nhwc_dgrad, conv2d_nhwc_wgrad, conv_lds_supported, conv_train_supported, maxpool_3x3s2_bwd_nhwc, maxpool_3x3s2_nhwc, maxpool_train_supported, pack_conv_views)          # noqa: E402,F401
