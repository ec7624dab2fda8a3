"""With-grad, training-mode forward of the ResNet patch encoder with fused BatchNorm kernels (SURVEY 8 f, N-b).

Reference: ``training/iterative.py:158-163`` calls ``net(mem_patch, mem_pos)`` under ``net.train()``;
``architecture/ips_net.py:273`` sends the ``B * M`` selected patches through the torchvision ResNet trunk with
BatchNorm in batch-statistics mode.  On stock ROCm ops the step's device time is dominated not by the convolutions
(MIOpen runs them on Winograd / implicit-GEMM kernels at ~90 TFLOP/s fp32-equivalent) but by what surrounds them:
per BasicBlock two BatchNorms in each direction, the ``+= identity``, the ReLUs and their ``threshold_backward``s,
layout transposes around MIOpen's channels-last kernels.  This module keeps the modules, parameters and buffers of
``IPSNet.encoder`` exactly as they are (state dicts, the optimizer and ``ips()`` see no difference) and evaluates them
differently:

* activations stay channels-last from the stem to the average pool, so MIOpen's NHWC convolution kernels run without
  transposes;
* every ``bn -> relu`` and ``bn -> (+ identity) -> relu`` is ONE autograd node backed by ``libipsx.so``
  (``ipsx_bn_train_forward`` / ``_backward``: two memory passes each way, csrc/bn_train.hip).

* (round 4) the convolutions run on libipsx's own fp32-MFMA kernels in all three directions (``_Conv``: forward / data
  gradient ``conv_nhwc_kernel``, weight gradient ``conv_wgrad_*_kernel``; the 1-channel stem: ``conv_any_kernel`` writing
  channels-last and ``conv_wgrad_stem_kernel``); ``IPSX_TRAIN_CONV=0`` hands them back to MIOpen.  A 3-channel stem and the
  two poolings remain stock ops.

Results equal the stock path to fp32 rounding (another summation order): tests/test_hip_train.py compares loss,
gradients, running statistics and post-step weights.  ``IPSX_TRAIN_FUSED=0`` switches it off.
"""
import os

import torch
import torch.nn.functional as F
from torch import nn

from .. import hip
from ..architecture.resnet import BasicBlock

_CL = torch.channels_last


def enabled():
    return os.environ.get("IPSX_TRAIN_FUSED", "1") != "0"


def _bn_ok(bn):
    return (isinstance(bn, nn.BatchNorm2d) and bn.affine and bn.track_running_stats and bn.momentum is not None
            and hip.bn_train_supported(1, bn.num_features))


def supported(encoder):
    """True for the encoders ``IPSNet.get_conv_patch_enc`` builds from BasicBlocks (ResNet-18)."""
    mods = list(encoder.children())
    if len(mods) < 6 or not (isinstance(mods[0], nn.Conv2d) and isinstance(mods[2], nn.ReLU)
                             and isinstance(mods[3], nn.MaxPool2d) and isinstance(mods[-1], nn.AdaptiveAvgPool2d)):
        return False
    if not _bn_ok(mods[1]) or mods[0].bias is not None:
        return False
    for stage in mods[4:-1]:
        if not isinstance(stage, nn.Sequential):
            return False
        for blk in stage:
            if not isinstance(blk, BasicBlock) or not (_bn_ok(blk.bn1) and _bn_ok(blk.bn2)):
                return False
            if blk.downsample is not None and not (isinstance(blk.downsample[0], nn.Conv2d) and _bn_ok(blk.downsample[1])):
                return False
    return True


class _BnAct(torch.autograd.Function):
    """y = [relu](batch_norm_train(x) [+ residual]) on channels-last tensors; running statistics updated in place."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, bn, relu):
        x = x.contiguous(memory_format=_CL)
        if residual is not None:
            residual = residual.contiguous(memory_format=_CL)
        y, mean, invstd = hip.bn_train_forward(x, residual, gamma, beta, bn.eps, bn.momentum, bn.running_mean,
                                               bn.running_var, relu)
        ctx.relu, ctx.has_res = relu, residual is not None
        ctx.save_for_backward(x, y if relu else None, gamma, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, mean, invstd = ctx.saved_tensors
        dy = dy.contiguous(memory_format=_CL)
        dx, dres, dgamma, dbeta = hip.bn_train_backward(dy, y, x, gamma, mean, invstd, ctx.relu, ctx.has_res)
        return dx, dgamma, dbeta, dres, None, None


def bn_act(x, bn, residual=None, relu=True):
    return _BnAct.apply(x, bn.weight, bn.bias, residual, bn, relu)


class _MaxPool(torch.autograd.Function):
    """nn.MaxPool2d(3, 2, 1) behind the stem on libipsx's kernels: forward without an index tensor, backward by finding every
    window's first maximum again (ATen's rule, bit for bit; csrc/pool_train.hip)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous(memory_format=_CL)
        ctx.save_for_backward(x)
        return hip.maxpool_3x3s2_nhwc(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return hip.maxpool_3x3s2_bwd_nhwc(x, dy)


def _pool(pool, x):
    if (isinstance(pool, nn.MaxPool2d) and pool.kernel_size in (3, (3, 3)) and pool.stride in (2, (2, 2)) and pool.padding in (1, (1, 1))
            and pool.dilation in (1, (1, 1)) and not pool.ceil_mode and hip.maxpool_train_supported(x)):
        return _MaxPool.apply(x)
    return pool(x)


def conv_enabled():
    return os.environ.get("IPSX_TRAIN_CONV", "1") != "0"


class _Conv(torch.autograd.Function):
    """Convolution of a channels-last activation on the fp32 matrix cores, all three directions on kernels of libipsx:
    forward and data gradient ``conv_nhwc_kernel`` (the data gradient = the same convolution of dy with the weights rotated
    by 180 degrees and transposed), weight gradient ``conv_wgrad_kernel`` (csrc/conv_wgrad.hip)."""

    @staticmethod
    def forward(ctx, x, weight, stride, pad, packs=None):
        x = x.contiguous(memory_format=_CL)
        ctx.save_for_backward(x, weight)
        ctx.geom = (stride, pad)
        ctx.packed_dgrad = packs[1] if packs is not None else None
        return hip.conv2d_nhwc(x, weight.detach(), stride, pad, packed=packs[0] if packs is not None else None)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        stride, pad = ctx.geom
        dy = dy.contiguous(memory_format=_CL)
        dx = hip.conv2d_nhwc_dgrad(dy, weight, stride, pad, x.shape[2:], packed=ctx.packed_dgrad) if ctx.needs_input_grad[0] else None
        dw = hip.conv2d_nhwc_wgrad(x, dy, weight.shape, stride, pad) if ctx.needs_input_grad[1] else None
        return dx, dw, None, None, None


class _ConvBnAct(torch.autograd.Function):
    """conv -> batch-norm (batch statistics) [-> + residual] [-> relu] as ONE node, for the convolutions that run on the
    LDS-resident stage kernels: the convolution's epilogue hands the BatchNorm its per-slab sums (around the running mean),
    so the statistics cost no pass over the convolution's output (``ipsx_conv2d_lds_nhwc_stats`` +
    ``ipsx_bn_train_forward_partials``; VERDICT r04: "BatchNorm statistics fused into the convolutions' epilogues").
    Backward = ``_BnAct.backward`` followed by ``_Conv.backward``."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, residual, bn, relu, stride, pad, packs):
        x = x.contiguous(memory_format=_CL)
        if residual is not None:
            residual = residual.contiguous(memory_format=_CL)
        shift = _stats_shift(bn, x, weight, stride, pad, packs)
        c, partial, slabs = hip.conv2d_nhwc(x, weight.detach(), stride, pad, packed=packs[0] if packs is not None else None,
                                            stats_shift=shift)
        y, mean, invstd = hip.bn_train_forward_partials(c, residual, gamma, beta, bn.eps, bn.momentum, bn.running_mean,
                                                        bn.running_var, relu, partial, slabs, shift)
        bn._ipsx_batch_mean = mean                # (a fresh tensor every step: the next step's shift, for free)
        ctx.relu, ctx.has_res, ctx.geom = relu, residual is not None, (stride, pad)
        ctx.packed_dgrad = packs[1] if packs is not None else None
        ctx.save_for_backward(x, weight, c, y if relu else None, gamma, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, c, y, gamma, mean, invstd = ctx.saved_tensors
        stride, pad = ctx.geom
        dy = dy.contiguous(memory_format=_CL)
        dc, dres, dgamma, dbeta = hip.bn_train_backward(dy, y, c, gamma, mean, invstd, ctx.relu, ctx.has_res)
        dx = hip.conv2d_nhwc_dgrad(dc, weight, stride, pad, x.shape[2:], packed=ctx.packed_dgrad) if ctx.needs_input_grad[0] else None
        dw = hip.conv2d_nhwc_wgrad(x, dc, weight.shape, stride, pad) if ctx.needs_input_grad[1] else None
        return dx, dw, dgamma, dbeta, dres, None, None, None, None, None


def _stats_shift(bn, x, weight, stride, pad, packs):
    """What the convolution's epilogue subtracts before it sums and squares (fp32, per slab of 4 patches): var = E[d^2] -
    E[d]^2 keeps its precision only while the shift is near the batch mean.  ``running_mean`` is not - 0 at
    initialisation, anything after loading foreign statistics (advisor, round 5) - so: the batch mean of this layer's
    PREVIOUS training step (the tensor that step returned, held by reference: no copy, no launch), and before there is one
    the channel means of the convolution of the batch's first patch (one extra launch on 1 patch, once per layer)."""
    prev = getattr(bn, "_ipsx_batch_mean", None)
    if prev is not None and prev.device == x.device and prev.shape == bn.running_mean.shape:
        return prev
    with torch.no_grad():
        c0 = hip.conv2d_nhwc(x[:1], weight.detach(), stride, pad, packed=packs[0] if packs is not None else None)
        return c0.mean((0, 2, 3)).contiguous()


def conv_bn_act(conv, bn, x, packs=None, residual=None, relu=True):
    """``bn_act(_conv(conv, x), bn, residual, relu)`` - as one node with the statistics off the convolution's epilogue where
    the convolution runs on the LDS-resident kernels."""
    if x.shape[0] == 0:                          # an empty batch: no slabs, no statistics, nothing to normalise
        return _conv(conv, x, packs)
    if _conv_ok(conv, x) and hip.conv_lds_supported(conv, x.shape[2], x.shape[3]) and bn.momentum is not None:
        return _ConvBnAct.apply(x, conv.weight, bn.weight, bn.bias, residual, bn, relu, conv.stride[0], conv.padding[0],
                                packs.get(conv) if packs else None)
    return bn_act(_conv(conv, x, packs), bn, residual, relu)


def _conv_ok(conv, x):
    return conv_enabled() and x.is_cuda and x.dtype == torch.float32 and hip.conv_train_supported(conv)


def _conv(conv, x, packs=None):
    if _conv_ok(conv, x):
        return _Conv.apply(x, conv.weight, conv.stride[0], conv.padding[0], packs.get(conv) if packs else None)
    return F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)


def pack_all(encoder, x):
    """The packed weights of every convolution of the trunk that runs on libipsx's kernels - forward form, and the
    data-gradient form (rotated by 180 degrees, transposed) where a gradient flows back through the layer's input - in ONE
    launch (they were one launch each per call and direction: 19 of the step's kernels).  -> {conv: (forward, dgrad | None)}"""
    if not (conv_enabled() and x.is_cuda and x.dtype == torch.float32):
        return {}
    mods = list(encoder.children())
    convs = [(mods[0], False)]                           # (module, a gradient w.r.t. its input is needed)
    for stage in mods[4:-1]:
        for blk in stage:
            convs += [(blk.conv1, True), (blk.conv2, True)]
            if blk.downsample is not None:
                convs.append((blk.downsample[0], True))
    grad = torch.is_grad_enabled()
    views, slots = [], []
    for conv, back in convs:
        if not hip.conv_train_supported(conv):
            continue
        w = conv.weight.detach()
        slots.append((conv, len(views), back and grad))
        views.append((w, False))
        if back and grad:
            views.append((w, True))
    packed = hip.pack_conv_views(views)
    return {conv: (packed[k], packed[k + 1] if both else None) for conv, k, both in slots}


def encode(encoder, x, taps=None):
    """(P, C, h, w) patches -> (P, D) embeddings; same value as ``encoder(x).flatten(1)`` in train mode.
    ``taps`` (a list, diagnostics): receives every post-ReLU activation (stem, and per block: after bn1, after the block)."""
    mods = list(encoder.children())
    conv1, bn1, _, pool = mods[:4]
    x = x.contiguous()
    if x.shape[1] == 1:
        # one input channel: NCHW and NHWC are the same memory, and torch reads the strides as NCHW - spell the
        # channels-last strides out so that the stem's output is channels-last too (no transposes around it)
        x = x.as_strided(x.shape, (x.shape[2] * x.shape[3], 1, x.shape[3], 1))
    else:
        x = x.contiguous(memory_format=_CL)
    packs = pack_all(encoder, x)
    h = conv_bn_act(conv1, bn1, x, packs, None, True)
    h = _pool(pool, h)
    if taps is not None:
        taps.append(h)
    for stage in mods[4:-1]:
        for blk in stage:
            idt = h
            o = conv_bn_act(blk.conv1, blk.bn1, h, packs, None, True)
            if taps is not None:
                taps.append(o)
            if blk.downsample is not None:
                idt = conv_bn_act(blk.downsample[0], blk.downsample[1], h, packs, None, False)
            h = conv_bn_act(blk.conv2, blk.bn2, o, packs, idt, True)
            if taps is not None:
                taps.append(h)
    # num_batches_tracked of every BatchNorm that ran, in one launch (nn.BatchNorm2d adds 1 per forward in train mode)
    torch._foreach_add_([m.num_batches_tracked for m in encoder.modules() if isinstance(m, nn.BatchNorm2d)], 1)
    return mods[-1](h).flatten(1)
