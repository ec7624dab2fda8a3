"""The training step's kernels behind ``training/fused_encoder.py`` (reference training/iterative.py:158-163: the with-grad
forward of the M selected patches and its backward): convolutions forward / data gradient / weight gradient, BatchNorm in
batch-statistics mode, the max-pool - ctypes bindings of csrc/bn_train.hip, conv_wgrad.hip, dgrad_s2.hip, stem_train.hip,
pool_train.hip (split out of ``hip.py`` in round 6; ``ips_amd.hip`` re-exports every public name)."""

import ctypes as C
import os

import torch

from .hip import (lib, _ck, _p, _f32, _stream, Conv)
from .hip_encoder import _pack_conv


# ---------------------------------------------------------------- training step (with-grad forward of the trunk)
_CL = torch.channels_last


def conv_train_supported(conv):
    """Can the training step's convolutions of ``conv`` (an nn.Conv2d) run on the kernels of libipsx?  Forward and data
    gradient: ``ipsx_conv2d_affine_nhwc`` (C_in % 32 == 0); weight gradient: ``ipsx_conv2d_wgrad_nhwc`` (channels % 64 == 0);
    strided layers need "same" padding for the data gradient's formulation (kernel - 1 = 2 pad)."""
    kh, kw = conv.kernel_size
    s, p = conv.stride[0], conv.padding[0]
    if (conv.bias is not None or conv.groups != 1 or conv.dilation != (1, 1) or conv.stride[0] != conv.stride[1]
            or conv.padding[0] != conv.padding[1] or kh != kw or conv.weight.dtype != torch.float32):
        return False
    if s > 1 and kh - 1 != 2 * p:
        return False
    if conv.in_channels % 32 != 0 and conv.in_channels != 1:    # forward kernels: channels-last (C_in % 32 == 0) or the 1-channel stem
        return False
    return bool(lib().ipsx_conv2d_wgrad_nhwc_supported(conv.in_channels, conv.out_channels, kh, kw, s, p))


def _pack_conv_view(weight, dgrad=False):
    """``_pack_conv`` of a weight tensor as it lies in memory (any strides: no contiguous copy) - or, ``dgrad``, of the
    weights rotated by 180 degrees and transposed: -> (packed, C_out, C_in) of the convolution they describe."""
    co, ci, kh, kw = weight.shape
    s_co, s_ci, s_kh, s_kw = weight.stride()
    if dgrad:
        n_out, n_in = ci, co
        base, sn, sc, sky, skx = (kh - 1) * s_kh + (kw - 1) * s_kw, s_ci, s_co, -s_kh, -s_kw
    else:
        n_out, n_in = co, ci
        base, sn, sc, sky, skx = 0, s_co, s_ci, s_kh, s_kw
    packed = torch.empty(lib().ipsx_packed_conv_weight_elems(n_out, n_in, kh, kw), dtype=torch.float32, device=weight.device)
    _ck(lib().ipsx_pack_conv_weight_strided(_p(weight), base, n_out, n_in, kh, kw, sn, sc, sky, skx, _p(packed), _stream()),
        "ipsx_pack_conv_weight_strided")
    return packed, n_out, n_in


class PackJob(C.Structure):
    """``ipsx_pack_job`` of include/ipsx.h"""
    _fields_ = [("w", C.c_void_p), ("base", C.c_int64), ("c_out", C.c_int), ("c_in", C.c_int), ("kh", C.c_int), ("kw", C.c_int),
                ("s_out", C.c_int64), ("s_in", C.c_int64), ("s_ky", C.c_int64), ("s_kx", C.c_int64), ("packed", C.c_void_p)]


_PACK_BATCH_MAX = 32


def pack_conv_views(views):
    """``_pack_conv_view`` of several (weight, dgrad) pairs as ONE launch per 32 (ipsx_pack_conv_weights_batch): the packed
    tensors are slices of one buffer.  -> list of packed tensors, in the order of ``views``."""
    if not views:
        return []
    dev = views[0][0].device
    sizes = []
    for weight, dgrad in views:
        co, ci, kh, kw = weight.shape
        n_out, n_in = (ci, co) if dgrad else (co, ci)
        sizes.append(lib().ipsx_packed_conv_weight_elems(n_out, n_in, kh, kw))
    arena = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
    out, off = [], 0
    for sz in sizes:
        out.append(arena[off:off + sz])
        off += sz
    for k0 in range(0, len(views), _PACK_BATCH_MAX):
        chunk = views[k0:k0 + _PACK_BATCH_MAX]
        jobs = (PackJob * len(chunk))()
        for j, (weight, dgrad) in enumerate(chunk):
            co, ci, kh, kw = weight.shape
            s_co, s_ci, s_kh, s_kw = weight.stride()
            if dgrad:
                jobs[j].c_out, jobs[j].c_in = ci, co
                jobs[j].base, jobs[j].s_out, jobs[j].s_in, jobs[j].s_ky, jobs[j].s_kx = (kh - 1) * s_kh + (kw - 1) * s_kw, s_ci, s_co, -s_kh, -s_kw
            else:
                jobs[j].c_out, jobs[j].c_in = co, ci
                jobs[j].base, jobs[j].s_out, jobs[j].s_in, jobs[j].s_ky, jobs[j].s_kx = 0, s_co, s_ci, s_kh, s_kw
            jobs[j].kh, jobs[j].kw = kh, kw
            jobs[j].w, jobs[j].packed = weight.data_ptr(), out[k0 + j].data_ptr()
        _ck(lib().ipsx_pack_conv_weights_batch(C.byref(jobs), len(chunk), _stream()), "ipsx_pack_conv_weights_batch")
    return out


def conv_lds_supported(conv, h, w):
    """True when ``conv2d_nhwc`` would run this ``nn.Conv2d`` on a kernel that can hand the BatchNorm behind it its batch
    statistics (``conv2d_nhwc(..., stats_shift=...)``): the LDS-resident stage kernels (maps of 32-px patches) and the
    1-channel stem on the matrix cores."""
    kh, kw = conv.kernel_size
    if os.environ.get("IPSX_TRAIN_CONV_STATS", "1") == "0":
        return False
    if conv.in_channels == 1:                    # the 32-px trunk's stem on the matrix cores (csrc/stem_train.hip)
        return bool(os.environ.get("IPSX_TRAIN_STEM_MFMA", "1") != "0" and lib().ipsx_stem7x7s2_nhwc_supported(
            1, conv.out_channels, kh, kw, conv.stride[0], conv.padding[0], h, w))
    return bool(kh == kw and os.environ.get("IPSX_TRAIN_CONV_LDS", "1") != "0"
                and lib().ipsx_conv2d_lds_nhwc_supported(conv.in_channels, conv.out_channels, kh, conv.stride[0], conv.padding[0], h, w))


def conv2d_nhwc(x, weight, stride, pad, dgrad_weights=False, packed=None, stats_shift=None):
    """Plain convolution of a channels-last (P, C_in, h, w) tensor with an OIHW ``weight`` on the fp32 matrix cores
    (conv_nhwc_kernel): -> channels-last (P, C_out, ho, wo).  ``packed``: the weight already packed for this direction
    (``pack_conv_views``).  ``stats_shift`` (a (C_out,) tensor; only where ``conv_lds_supported``): -> (y, partial, slabs),
    the output's per-slab sums around that shift for ``bn_train_forward_partials``."""
    if x.dim() != 4 or x.dtype != torch.float32 or not x.is_contiguous(memory_format=_CL):
        raise ValueError("expected a float32 channels-last (P, C, H, W) tensor")
    kh, kw = weight.shape[2:]
    n, _, h, w = x.shape
    ho, wo = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1
    if packed is None:
        packed, co, ci = _pack_conv_view(weight.detach(), dgrad_weights)
    else:
        co, ci = (weight.shape[1], weight.shape[0]) if dgrad_weights else (weight.shape[0], weight.shape[1])
    cv = Conv(ci, co, kh, kw, stride, pad, _p(packed), None, None, None)
    y = torch.empty((n, co, ho, wo), dtype=torch.float32, device=x.device, memory_format=_CL)
    if n == 0:                                  # (an empty batch: nothing to launch)
        return (y, torch.zeros((1, 2, co), dtype=torch.float32, device=x.device), 0) if stats_shift is not None else y
    if stats_shift is not None:
        slabs = int(lib().ipsx_conv2d_lds_nhwc_stats_slabs(n))
        partial = torch.empty((max(slabs, 1), 2, co), dtype=torch.float32, device=x.device)
        if ci == 1:
            _ck(lib().ipsx_stem7x7s2_nhwc(C.byref(cv), _p(x), _p(y), n, _p(stats_shift), _p(partial), _stream()), "ipsx_stem7x7s2_nhwc")
        else:
            _ck(lib().ipsx_conv2d_lds_nhwc_stats(C.byref(cv), _p(x), _p(y), n, h, w, _p(stats_shift), _p(partial), _stream()),
                "ipsx_conv2d_lds_nhwc_stats")
        return y, partial, slabs
    if kh == kw and os.environ.get("IPSX_TRAIN_CONV_LDS", "1") != "0" and lib().ipsx_conv2d_lds_nhwc_supported(ci, co, kh, stride, pad, h, w):
        # the maps of 32-px patches: the fused trunk's stage kernels, map LDS-resident for all taps
        _ck(lib().ipsx_conv2d_lds_nhwc(C.byref(cv), _p(x), _p(y), n, h, w, _stream()), "ipsx_conv2d_lds_nhwc")
    elif ci == 1 and os.environ.get("IPSX_TRAIN_STEM_MFMA", "1") != "0" and lib().ipsx_stem7x7s2_nhwc_supported(ci, co, kh, kw, stride, pad, h, w):
        # the 32-px trunk's stem on the matrix cores (csrc/stem_train.hip)
        _ck(lib().ipsx_stem7x7s2_nhwc(C.byref(cv), _p(x), _p(y), n, None, None, _stream()), "ipsx_stem7x7s2_nhwc")
    elif ci == 1:       # one input channel (NCHW = channels-last memory): the stem kernel, channels-last output
        _ck(lib().ipsx_conv2d_affine_to_nhwc(C.byref(cv), _p(x), None, _p(y), n, h, w, 0, _stream()), "ipsx_conv2d_affine_to_nhwc")
    else:
        _ck(lib().ipsx_conv2d_affine_nhwc(C.byref(cv), _p(x), None, _p(y), n, h, w, 0, _stream()), "ipsx_conv2d_affine_nhwc")
    return y


def conv2d_nhwc_dgrad(dy, weight, stride, pad, in_hw, packed=None):
    """Data gradient of ``conv2d_nhwc``: the same kernel on dy with the weights rotated by 180 degrees and transposed; a
    strided layer first spreads dy over a zero map of the input's size (needs kernel - 1 = 2 pad)."""
    co, ci, kh, kw = weight.shape
    if (stride > 1 and kh == kw and os.environ.get("IPSX_TRAIN_DGRAD_S2", "1") != "0"
            and lib().ipsx_conv2d_dgrad_s2_lds_nhwc_supported(ci, co, kh, stride, pad, in_hw[0], in_hw[1])):
        # the 32-px trunk's strided layer by parity class of the input pixel: no spread map (csrc/dgrad_s2.hip)
        if packed is None:
            packed = _pack_conv_view(weight.detach(), True)[0]
        n = dy.shape[0]
        if dy.dtype != torch.float32 or tuple(dy.shape[1:]) != (co, in_hw[0] // 2, in_hw[1] // 2):
            raise ValueError("dy: expected float32 (P, %d, %d, %d)" % (co, in_hw[0] // 2, in_hw[1] // 2))
        dy = dy.contiguous(memory_format=_CL)
        dx = torch.empty((n, ci, in_hw[0], in_hw[1]), dtype=torch.float32, device=dy.device, memory_format=_CL)
        _ck(lib().ipsx_conv2d_dgrad_s2_lds_nhwc(_p(packed), kh, _p(dy), _p(dx), n, _stream()), "ipsx_conv2d_dgrad_s2_lds_nhwc")
        return dx
    if stride > 1:
        n = dy.shape[0]
        spread = torch.empty((n, co, in_hw[0], in_hw[1]), dtype=torch.float32, device=dy.device, memory_format=_CL).zero_()
        spread[:, :, ::stride, ::stride] = dy
        dy = spread
    return conv2d_nhwc(dy, weight, 1, kh - 1 - pad, dgrad_weights=True, packed=packed)


# ipsx_conv2d_wgrad_nhwc addresses x and dy through 32-bit buffer offsets: either activation of ONE call stays below this
# many bytes (csrc/conv_wgrad.hip: "call per slice and add"); conv2d_nhwc_wgrad slices the image axis accordingly
_WGRAD_MAX_BYTES = (1 << 31) - (1 << 20)


def conv2d_nhwc_wgrad(x, dy, weight_shape, stride, pad):
    """Weight gradient of ``conv2d_nhwc`` -> (C_out, C_in, kh, kw) in channels-last memory order (ipsx_conv2d_wgrad_nhwc).
    Activations of 2 GiB and more (the kernel's buffer range) are taken in slices of whole images, the slices' gradients
    added in slice order (deterministic; the forward kernel slices per launch in the same way)."""
    co, ci, kh, kw = weight_shape
    n, _, h, w = x.shape
    dy = dy.contiguous(memory_format=_CL)
    ho, wo = dy.shape[2:]
    per_image = 4 * max(h * w * ci, ho * wo * co)
    step = max(1, min(n, _WGRAD_MAX_BYTES // per_image))
    dw = torch.empty((co, ci, kh, kw), dtype=torch.float32, device=x.device, memory_format=_CL)
    nb = max(lib().ipsx_conv2d_wgrad_nhwc_workspace_bytes(c, ci, co, kh, kw) for c in {step, n - (n - 1) // step * step})
    ws = torch.empty(max(nb, 1), dtype=torch.uint8, device=x.device)
    part = dw
    for i0 in range(0, max(n, 1), step):
        cnt = min(step, n - i0)
        if i0 > 0 and part is dw:
            part = torch.empty_like(dw)
        _ck(lib().ipsx_conv2d_wgrad_nhwc(_p(x[i0:i0 + cnt]), _p(dy[i0:i0 + cnt]), cnt, h, w, ci, co, kh, kw, stride, pad,
                                         _p(part), _p(ws), nb, _stream()), "ipsx_conv2d_wgrad_nhwc")
        if part is not dw:
            dw += part
    return dw


def _rows_cl(t):
    """(P, C, H, W) channels-last tensor -> (rows, C) of its memory."""
    if t.dim() != 4 or t.dtype != torch.float32 or not t.is_contiguous(memory_format=torch.channels_last):
        raise ValueError("expected a float32 channels-last (P, C, H, W) tensor")
    return t.shape[0] * t.shape[2] * t.shape[3], t.shape[1]


def bn_train_supported(rows, c):
    return bool(lib().ipsx_bn_train_supported(rows, c))


def bn_train_forward(x, residual, gamma, beta, eps, momentum, running_mean, running_var, relu):
    """Batch-statistics BatchNorm2d (+ residual) (+ ReLU) of a channels-last activation; updates the running
    statistics in place.  Returns y (channels-last), mean, invstd."""
    rows, c = _rows_cl(x)
    if residual is not None and _rows_cl(residual) != (rows, c):
        raise ValueError("residual shape")
    y = torch.empty_like(x)                        # (preserves channels-last)
    # mean | invstd are saved for backward, so they are an allocation of their own (2 C floats): carved out of the
    # workspace they would keep its ~4 KB x C alive until backward, per BatchNorm (~20 MB per ResNet-18 step)
    stat = torch.empty(2 * c, dtype=torch.float32, device=x.device)
    mean, invstd = stat[:c], stat[c:]
    ws = torch.empty(max(1, _bn_workspace_floats(rows, c)), dtype=torch.float32, device=x.device)
    _ck(lib().ipsx_bn_train_forward(_p(x), _p(residual), rows, c, _p(_f32(gamma)), _p(_f32(beta)), eps, momentum,
                                    _p(running_mean), _p(running_var), int(relu), _p(y), _p(mean), _p(invstd),
                                    _p(ws), _stream()), "ipsx_bn_train_forward")
    return y, mean, invstd


def maxpool_train_supported(x):
    """Can ``maxpool_3x3s2_nhwc`` / ``maxpool_3x3s2_bwd_nhwc`` take this activation (the training step's pooling behind the
    stem: float32 on the GPU, 16 x 16 maps, a multiple of 32 channels)?"""
    return bool(x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and os.environ.get("IPSX_TRAIN_POOL", "1") != "0"
                and lib().ipsx_maxpool_3x3s2_bwd_nhwc_supported(x.shape[1], x.shape[2], x.shape[3]))


def maxpool_3x3s2_nhwc(x):
    """nn.MaxPool2d(3, 2, 1) of a channels-last (P, C, h, w) activation -> channels-last (P, C, ho, wo)."""
    x = x.contiguous(memory_format=_CL)
    n, c, h, w = x.shape
    y = torch.empty((n, c, (h - 1) // 2 + 1, (w - 1) // 2 + 1), dtype=torch.float32, device=x.device, memory_format=_CL)
    _ck(lib().ipsx_maxpool_3x3s2_nhwc(_p(x), _p(y), n, c, h, w, _stream()), "ipsx_maxpool_3x3s2_nhwc")
    return y


def maxpool_3x3s2_bwd_nhwc(x, dy):
    """The gradient of ``maxpool_3x3s2_nhwc`` w.r.t. x (ATen's rule: a window's gradient goes to its first maximum)."""
    x = x.contiguous(memory_format=_CL)
    dy = dy.contiguous(memory_format=_CL)
    n, c, h, w = x.shape
    dx = torch.empty_like(x)
    _ck(lib().ipsx_maxpool_3x3s2_bwd_nhwc(_p(x), _p(dy), _p(dx), n, c, h, w, _stream()), "ipsx_maxpool_3x3s2_bwd_nhwc")
    return dx


def bn_train_forward_partials(x, residual, gamma, beta, eps, momentum, running_mean, running_var, relu, partial, slabs, shift):
    """``bn_train_forward`` without its reduction pass: ``partial`` (slabs, 2, C) are the sums ``conv2d_nhwc(..., stats_shift=
    shift)`` took off its accumulators (``shift`` may be ``running_mean`` itself)."""
    rows, c = _rows_cl(x)
    if residual is not None and _rows_cl(residual) != (rows, c):
        raise ValueError("residual shape")
    y = torch.empty_like(x)
    stat = torch.empty(2 * c, dtype=torch.float32, device=x.device)
    mean, invstd = stat[:c], stat[c:]
    _ck(lib().ipsx_bn_train_forward_partials(_p(x), _p(residual), rows, c, _p(_f32(gamma)), _p(_f32(beta)), eps, momentum,
                                             _p(running_mean), _p(running_var), int(relu), _p(y), _p(mean), _p(invstd),
                                             _p(partial), slabs, _p(shift), _stream()), "ipsx_bn_train_forward_partials")
    return y, mean, invstd


_BN_MAX_SLABS = 512      # csrc/bn_train.hip BN_MAX_SLABS (ipsx_bn_train_workspace_floats never exceeds 2 * 512 * C)
_BN_WS_FLOATS = {}


def _bn_workspace_floats(rows, c):
    """ipsx_bn_train_workspace_floats(rows, c), remembered per shape (a training step asks for the same few shapes)."""
    key = (rows, c)
    n = _BN_WS_FLOATS.get(key)
    if n is None:
        n = _BN_WS_FLOATS[key] = int(lib().ipsx_bn_train_workspace_floats(rows, c))
    return n


def bn_train_backward(dy, y, x, gamma, mean, invstd, relu, want_residual):
    """-> dx, dresidual | None, dgamma, dbeta."""
    rows, c = _rows_cl(x)
    if _rows_cl(dy) != (rows, c):
        raise ValueError("dy shape")
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_residual else None
    dgamma = torch.empty(c, dtype=torch.float32, device=x.device)
    dbeta = torch.empty_like(dgamma)
    ws = torch.empty(max(1, _bn_workspace_floats(rows, c)), dtype=torch.float32, device=x.device)
    _ck(lib().ipsx_bn_train_backward(_p(dy), _p(y), _p(x), rows, c, _p(_f32(gamma)), _p(mean), _p(invstd), int(relu),
                                     _p(dx), _p(dres), _p(dgamma), _p(dbeta), _p(ws), _stream()),
        "ipsx_bn_train_backward")
    return dx, dres, dgamma, dbeta
