"""``EncoderPlan``: the device-side description of ``IPSNet.encoder`` (packed weights + BatchNorm affines) and the entry
points that run it - the fused trunks, the layer-by-layer trunk, the projector and the two persistent producer streams
(split out of ``hip.py`` in round 6; ``ips_amd.hip`` re-exports every name, so ``hip.EncoderPlan`` etc. are unchanged).
Reference: ``IPSNet.encoder``, architecture/ips_net.py:17-60."""

import collections
import ctypes as C
import os

import torch

from .hip import (lib, _ck, _p, _f32, _stream, _patches, _PATCH_DTYPES, Conv, Block, Trunk, precision, dedup_blank, weights_generation)


# ------------------------------------------------------------------ encoder plan
def _bn_affine(bn, bias=None):
    """Per-channel (alpha, shift) of an eval-mode BatchNorm, on the device."""
    c = bn.num_features
    out = torch.empty((2, c), dtype=torch.float32, device=bn.weight.device)
    lin_bias = _f32(bias.detach()) if bias is not None else None
    _ck(lib().ipsx_bn_affine(_p(_f32(bn.weight.detach())), _p(_f32(bn.bias.detach())),
                             _p(_f32(bn.running_mean)), _p(_f32(bn.running_var)), _p(lin_bias),
                             C.c_float(bn.eps), c, _p(out[0]), _p(out[1]), _stream()), "ipsx_bn_affine")
    return out


def _pack_conv(weight):
    co, ci, kh, kw = weight.shape
    n = lib().ipsx_packed_conv_weight_elems(co, ci, kh, kw)
    packed = torch.empty(n, dtype=torch.float32, device=weight.device)
    _ck(lib().ipsx_pack_conv_weight(_p(_f32(weight.detach())), co, ci, kh, kw, _p(packed), _stream()),
        "ipsx_pack_conv_weight")
    return packed


class _PlanHold:
    """``EncoderPlan.hold()``: the plan's weight check runs on entry and is skipped until exit."""
    __slots__ = ("plan",)

    def __init__(self, plan):
        self.plan = plan

    def __enter__(self):
        self.plan._refresh()
        self.plan._held += 1

    def __exit__(self, *exc):
        self.plan._held -= 1


class EncoderPlan:
    """Device-side description of ``IPSNet.encoder``: packed weights + BN affines.

    Parameters change every optimiser step and BatchNorm running statistics move
    in every training-mode forward, so the plan is keyed on the tensors'
    ``_version`` counters / storage pointers and re-packed when any moved.
    """

    def __init__(self, encoder, is_image):
        self.encoder, self.is_image = encoder, is_image
        self._sig = None
        self._holders = None
        self._keep = []
        self._ws = None
        self._ws_small = 0
        self._held = 0

    def _walk(self):
        """The module tree, flattened ONCE: every module's child dictionary (the structural fingerprint is the ids of their
        values, re-read in every call - plain dictionary reads, ~5 us for a ResNet trunk, where ``encoder.modules()`` costs
        ~100 us in front of the first launch of EVERY ips() call), the (dictionary, key) slot of every parameter / buffer
        that exists, and the slots that are None today (a bias or a running statistic that appears later is seen)."""
        kids, slots, empty = [], [], []
        for mod in self.encoder.modules():
            kids.append(mod._modules)
            for d in (mod._parameters, mod._buffers):
                for k, t in d.items():
                    (slots if t is not None else empty).append((d, k))
        return kids, slots, empty

    @staticmethod
    def _structure(kids):
        return tuple(id(c) for d in kids for c in d.values())

    def _signature(self):
        """(storage pointer, version counter) of every parameter and buffer + the ids of every child module: a tensor that
        is replaced, moved or written in place, a child module exchanged at ANY depth (``layer2[0].bn1 = ...``,
        ``convert_sync_batchnorm``), an entry that appears, disappears or stops being None - each re-packs the plan."""
        h = self._holders
        if h is not None:
            try:
                if self._structure(h[1]) != h[0] or any(d[k] is not None for d, k in h[3]):
                    h = None
            except KeyError:
                h = None
        for _ in range(2):
            if h is None:
                kids, slots, empty = self._walk()
                h = self._holders = (self._structure(kids), kids, slots, empty)
            sig = [precision(), weights_generation(), h[0]]
            try:
                for d, k in h[2]:
                    t = d[k]
                    sig.append((t.data_ptr(), t._version))
                return tuple(sig)
            except (KeyError, AttributeError):         # an entry was removed / set to None since the walk: walk again
                h = None
        raise RuntimeError("EncoderPlan: the encoder's parameters changed while they were being read")

    def _conv(self, conv, bn, prec=0, stem=False):
        packed = _pack_conv(conv.weight)
        aff = _bn_affine(bn)
        self._keep += [packed, aff]
        half = None
        if prec and stem:
            if tuple(conv.weight.shape[1:]) == (1, 7, 7):     # the split trunks exist for the 1x32x32 stem only
                w = _f32(conv.weight.detach())
                planes = 1 if prec == 1 else 3
                half = torch.empty(lib().ipsx_packed_stem_weight_split_bytes(w.shape[0], planes), dtype=torch.uint8, device=w.device)
                _ck(lib().ipsx_pack_stem_weight_split(_p(w), w.shape[0], planes, _p(half), _stream()), "ipsx_pack_stem_weight_split")
                self._keep.append(half)
        elif prec:
            w = _f32(conv.weight.detach())
            co, ci, kh, kw = w.shape
            size, pack = ((lib().ipsx_packed_conv_weight_bf16_bytes, lib().ipsx_pack_conv_weight_bf16) if prec == 1 else
                          (lib().ipsx_packed_conv_weight_x3_bytes, lib().ipsx_pack_conv_weight_x3))
            half = torch.empty(size(co, ci, kh, kw), dtype=torch.uint8, device=w.device)
            _ck(pack(_p(w), co, ci, kh, kw, _p(half), _stream()), "ipsx_pack_conv_weight_bf16/x3")
            self._keep.append(half)
        return Conv(conv.in_channels, conv.out_channels, conv.kernel_size[0], conv.kernel_size[1],
                    conv.stride[0], conv.padding[0], _p(packed), _p(aff[0]), _p(aff[1]), _p(half))

    def _rebuild(self):
        self._keep = []
        enc = self.encoder
        if self.is_image:
            mods = list(enc.children())
            bf16 = {"fp32": 0, "bf16": 1, "fp32x3": 2}[precision()]
            blocks = []
            for stage in mods[4:-1]:
                for blk in stage.children():
                    b = Block()
                    pairs = [(getattr(blk, "conv%d" % i), getattr(blk, "bn%d" % i))
                             for i in (1, 2, 3) if hasattr(blk, "conv%d" % i)]
                    b.n_conv = len(pairs)
                    for j, (cv, bn) in enumerate(pairs):
                        b.conv[j] = self._conv(cv, bn, bf16)
                    b.has_down = int(blk.downsample is not None)
                    if b.has_down:
                        b.down = self._conv(blk.downsample[0], blk.downsample[1], bf16)
                    blocks.append(b)
            self._blocks = (Block * len(blocks))(*blocks)
            t = Trunk()
            t.stem = self._conv(mods[0], mods[1], bf16, stem=True)
            t.c_in = mods[0].in_channels
            t.n_block = len(blocks)
            t.blocks = C.cast(self._blocks, C.POINTER(Block))
            t.precision = bf16
            t.patch_dtype = 0
            self.trunk = t
            self.d_out = blocks[-1].conv[blocks[-1].n_conv - 1].c_out
        else:
            ln, lin, bn = enc[0], enc[1], enc[2]
            w = lin.weight.detach()
            packed = _pack_conv(w.reshape(w.shape[0], w.shape[1], 1, 1))
            aff = _bn_affine(bn, bias=lin.bias)
            # the LayerNorm in front of the Linear is folded into the GEMM's epilogue: rstd * (x W^T - mean * colsum(W))
            colsum = torch.empty(w.shape[0], dtype=torch.float32, device=w.device)
            wf = _f32(w)
            _ck(lib().ipsx_weight_colsum(_p(wf), w.shape[0], w.shape[1], _p(colsum), _stream()), "ipsx_weight_colsum")
            self._keep += [packed, aff, colsum, wf]
            self.lin = Conv(w.shape[1], w.shape[0], 1, 1, 1, 0, _p(packed), _p(aff[0]), _p(aff[1]), None, _p(colsum))
            self.ln_eps = float(ln.eps)
            self.d_out = w.shape[0]

    def _workspace(self, nbytes, device):
        # grown on demand; given back when much smaller requests keep coming (one large evaluation call must not pin
        # tens of GiB for the rest of a training run) - after several in a row, not after one: lazy slabs of 1/6, 1/2 and
        # full size, or an eval call between training steps, would otherwise free and re-allocate gigabytes per call
        small = self._ws is not None and self._ws.numel() > (256 << 20) and nbytes < self._ws.numel() // 4
        self._ws_small = self._ws_small + 1 if small else 0
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != device or self._ws_small >= 8:
            self._ws = None
            self._ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
            self._ws_small = 0
        return self._ws

    def _refresh(self):
        if self._held:
            return
        sig = self._signature()
        if sig != self._sig:
            self._rebuild()
            self._sig = sig

    def hold(self):
        """Context manager: check the weights once, then skip the check until the block ends - for a caller that makes
        several encode calls while the weights cannot change (one no-grad ``ips()`` call; the check walks ~80 tensors)."""
        return _PlanHold(self)

    def fused(self, x_shape):
        """True when encode_indexed is available for patches of this (C, h, w)."""
        if not self.is_image:
            return False
        self._refresh()
        self.trunk.h, self.trunk.w = x_shape[-2], x_shape[-1]
        return x_shape[-3] == self.trunk.c_in and lib().ipsx_trunk_kernel(C.byref(self.trunk)).startswith(b"fused")

    def encode_indexed(self, flat, index):
        """flat (P, C, h, w) contiguous on the GPU, index (n,) int32 -> (n, D) embeddings of flat[index]."""
        self._refresh()
        flat = _patches(flat)
        self.trunk.patch_dtype = _PATCH_DTYPES[flat.dtype]
        out = torch.empty((index.numel(), self.d_out), dtype=torch.float32, device=flat.device)
        try:
            _ck(lib().ipsx_trunk_encode_indexed(C.byref(self.trunk), _p(flat), _p(index), index.numel(), _p(out),
                                                _stream()), "ipsx_trunk_encode_indexed")
        finally:
            self.trunk.patch_dtype = 0
        return out

    def encode_plain(self, x, out=None):
        """The image trunk on every patch of ``x`` (no dedup)."""
        x = _patches(x)
        n = x.shape[0]
        if out is None:
            out = torch.empty((n, self.d_out), dtype=torch.float32, device=x.device)
        self.trunk.patch_dtype = _PATCH_DTYPES[x.dtype]
        try:
            # Layer-by-layer trunks: the batch goes through in two halves on two streams.  The stem and the max-pool are
            # HBM-bound (together 11-13 % of the trunk's time for 1 % of its arithmetic), the residual stages MFMA-bound:
            # side by side, one half's stem / pool / epilogues fill what the other half's convolutions leave idle
            # (50-px MNIST 14.25 -> 13.89 ms, traffic signs 22.75 -> 22.11 ms; three streams gain less).  Same kernels on
            # the same patches: results are unchanged.  IPSX_LAYERED_STREAMS=1 switches it off.
            ns = int(os.environ.get("IPSX_LAYERED_STREAMS", "2"))
            if ns > 1 and n >= 1024 and not lib().ipsx_trunk_kernel(C.byref(self.trunk)).startswith(b"fused"):
                cuts = [n * k // ns for k in range(ns + 1)]
                nb = lib().ipsx_trunk_workspace_bytes(C.byref(self.trunk), max(cuts[k + 1] - cuts[k] for k in range(ns)))
                # the library's budget (a share of the free memory) is per CALL: the ns concurrent calls split it - each
                # chunks its part of the batch to the workspace it is given
                budget = lib().ipsx_trunk_workspace_bytes(C.byref(self.trunk), 1 << 40)
                nb = min(nb, max(budget // ns, lib().ipsx_trunk_workspace_bytes(C.byref(self.trunk), 1)))
                nb -= nb % 256
                ws = self._workspace(ns * nb, x.device)
                if len(getattr(self, "_sides", [])) < ns - 1:
                    self._sides = [torch.cuda.Stream(device=x.device) for _ in range(ns - 1)]
                main = torch.cuda.current_stream(x.device)
                for k in range(1, ns):
                    st = self._sides[k - 1]
                    st.wait_stream(main)
                    with torch.cuda.stream(st):
                        _ck(lib().ipsx_trunk_encode(C.byref(self.trunk), _p(x[cuts[k]:cuts[k + 1]]), cuts[k + 1] - cuts[k],
                                                    _p(out[cuts[k]:cuts[k + 1]]), _p(ws[k * nb:]), nb, _stream()), "ipsx_trunk_encode")
                _ck(lib().ipsx_trunk_encode(C.byref(self.trunk), _p(x[:cuts[1]]), cuts[1], _p(out[:cuts[1]]), _p(ws[:nb]), nb, _stream()),
                    "ipsx_trunk_encode")
                for st in self._sides[:ns - 1]:
                    main.wait_stream(st)
                return out
            nb = lib().ipsx_trunk_workspace_bytes(C.byref(self.trunk), n)
            ws = self._workspace(nb, x.device)
            _ck(lib().ipsx_trunk_encode(C.byref(self.trunk), _p(x), n, _p(out), _p(ws), nb, _stream()), "ipsx_trunk_encode")
        finally:
            self.trunk.patch_dtype = 0
        return out

    def row_stats(self, x, out=None):
        """(mean, rstd) of every feature row of ``x`` (P, F) -> (P, 2): the LayerNorm moments the projector's GEMM applies
        to its operand; for callers that run this HBM-bound pass ahead of / beside the GEMM (``encode(x, stats=...)``)."""
        self._refresh()
        x = _f32(x)
        if out is None:
            out = torch.empty((x.shape[0], 2), dtype=torch.float32, device=x.device)
        _ck(lib().ipsx_projector_stats(_p(x), x.shape[0], x.shape[1], C.c_float(self.ln_eps), _p(out), _stream()),
            "ipsx_projector_stats")
        return out

    def image_stream_supported(self, x_shape, D, R):
        """Can ``image_stream`` encode patches of this shape (the fused fp32 1x32x32 trunk, 128 features, R <= 32)?"""
        if not self.is_image or not self.fused(x_shape) or precision() != "fp32":
            return False
        return bool(lib().ipsx_trunk_stream_supported(C.byref(self.trunk), int(D), int(R)))

    def image_stream(self, x, pos, vq, R, emb, logits, ctl, ready, workgroups=0, quad_pulls=-1):
        """Trunk + logits of ONE image's patches ``x`` (P, 1, 32, 32) as one persistent launch that advances ``ready`` (the
        progress word of ``scan_persistent``) as patches complete: ``emb`` (P, 128) and ``logits`` (P, R) are the outputs,
        ``pos`` (P, 128) or None is added to the embeddings for the logits, ``ctl`` =
        ``torch.zeros(image_stream_ctl_words(P), int32)`` zeroed before every call, ``vq`` the folded query."""
        self._refresh()
        x = _patches(x)
        if pos is not None and (pos.stride(-1) != 1 or pos.stride(-2) != pos.shape[-1]):
            pos = pos.contiguous()
        _ck(lib().ipsx_trunk_stream(C.byref(self.trunk), _p(x), x.shape[0], _p(emb), _p(pos), _p(vq), int(R), _p(logits),
                                    _p(ctl), _p(ready), int(workgroups), int(quad_pulls), _stream()), "ipsx_trunk_stream")
        return emb

    @staticmethod
    def image_stream_ctl_words(n):
        return int(lib().ipsx_trunk_stream_ctl_words(int(n)))

    def stream_supported(self, n, R):
        """Can ``stream`` run this projector on ``n`` rows with ``R`` logits per row?"""
        if self.is_image:
            return False
        self._refresh()
        return bool(lib().ipsx_projector_stream_supported(C.byref(self.lin), int(n), int(R)))

    def stream(self, x, vq, R, emb, logits, ctl, ready, workgroups=0, short_first=-1, slide_rows=None):
        """Projector + logits of the feature rows ``x`` (P, F) - one slide, or several one after the other, ``slide_rows``
        each - as one persistent launch that advances ``ready`` (the progress word(s) of ``scan_persistent``, one per
        slide) as rows complete: ``emb`` (P, 512) and ``logits`` (P, R) are the outputs, ``ctl`` =
        ``torch.zeros(stream_ctl_words(P), int32)`` zeroed before every call, ``vq`` the folded query."""
        self._refresh()
        x = _f32(x)
        _ck(lib().ipsx_projector_stream(C.byref(self.lin), _p(x), x.shape[0], int(slide_rows or x.shape[0]),
                                        C.c_float(self.ln_eps), _p(emb), _p(vq), int(R),
                                        _p(logits), _p(ctl), _p(ready), int(workgroups), int(short_first), _stream()),
            "ipsx_projector_stream")
        return emb

    @staticmethod
    def stream_ctl_words(n):
        return int(lib().ipsx_projector_stream_ctl_words(int(n)))

    @staticmethod
    def stream_ctl_zero_words(n):
        """... of which only the first this many have to be zero when a call starts."""
        return int(lib().ipsx_projector_stream_ctl_zero_words(int(n)))

    def encode(self, x, nonblank=None, stats=None, out=None, publish=None):
        """(P, C, h, w) or (P, F) float32 on the GPU  ->  (P, D).

        ``nonblank`` (P int32, 1 = the patch has a non-zero element; e.g. from ``patchify_sparse``) switches on
        the exact blank-patch dedup without the pass that looks for blank patches.  ``publish`` = (ready, value), with
        ``stats``: the GEMM launch also does ``publish_rows(ready, value)`` for what was enqueued before it."""
        self._refresh()
        x = _patches(x) if self.is_image else _f32(x)
        n = x.shape[0]
        if x.dtype != torch.float32 and (dedup_blank() or nonblank is not None):
            raise TypeError("blank-patch dedup reads float32 patches")
        if out is None:
            out = torch.empty((n, self.d_out), dtype=torch.float32, device=x.device)
        elif tuple(out.shape) != (n, self.d_out) or out.dtype != torch.float32 or not out.is_contiguous():
            raise ValueError("out must be a contiguous (P, D) float32 tensor")
        if n == 0:
            return out
        if self.is_image:
            self.trunk.h, self.trunk.w = x.shape[2], x.shape[3]
            if x.shape[1] != self.trunk.c_in:
                raise ValueError("patches have {} channels, encoder expects {}".format(x.shape[1], self.trunk.c_in))
            if (dedup_blank() or nonblank is not None) and \
                    lib().ipsx_trunk_kernel(C.byref(self.trunk)).startswith(b"fused_trunk"):      # any precision
                nb = lib().ipsx_trunk_dedup_workspace_bytes(C.byref(self.trunk), n)
                ws = self._workspace(nb, x.device)
                self.n_encoded = torch.zeros((), dtype=torch.int32, device=x.device)
                if nonblank is not None:
                    if nonblank.dtype != torch.int32 or nonblank.numel() != n or not nonblank.is_contiguous():
                        raise ValueError("nonblank must be a contiguous int32 tensor with one flag per patch")
                    _ck(lib().ipsx_trunk_encode_dedup_flagged(C.byref(self.trunk), _p(x), n, _p(nonblank), _p(out), _p(ws),
                                                              nb, _p(self.n_encoded), _stream()),
                        "ipsx_trunk_encode_dedup_flagged")
                else:
                    _ck(lib().ipsx_trunk_encode_dedup(C.byref(self.trunk), _p(x), n, _p(out), _p(ws), nb,
                                                      _p(self.n_encoded), _stream()), "ipsx_trunk_encode_dedup")
                return out
            if (dedup_blank() or nonblank is not None) and n > 1:
                # layer-by-layer trunks (other patch sizes / depths): the same exact dedup with the index handling in
                # torch - the layered launches are sized on the host, so the number of distinct patches is read back
                # (one synchronisation per call; the fused trunk above needs none)
                flags = nonblank.bool() if nonblank is not None else (x.flatten(1) != 0).any(1)
                keep = torch.nonzero(flags).flatten()
                blank = torch.nonzero(~flags).flatten()
                if blank.numel() > 1:
                    sel = torch.cat((keep, blank[:1]))
                    uniq = self.encode_plain(x[sel])
                    out[keep] = uniq[:keep.numel()]
                    out[blank] = uniq[keep.numel():keep.numel() + 1]
                    self.n_encoded = torch.tensor(sel.numel(), dtype=torch.int32, device=x.device)
                    return out
            return self.encode_plain(x, out)
        elif stats is not None:
            if stats.shape != (n, 2) or stats.dtype != torch.float32 or not stats.is_contiguous():
                raise ValueError("stats must be a contiguous (P, 2) float32 tensor")
            if publish is not None:
                _ck(lib().ipsx_projector_apply_publish(C.byref(self.lin), _p(x), n, _p(stats), _p(out), _p(publish[0]),
                                                       int(publish[1]), _stream()), "ipsx_projector_apply_publish")
            else:
                _ck(lib().ipsx_projector_apply(C.byref(self.lin), _p(x), n, _p(stats), _p(out), _stream()), "ipsx_projector_apply")
        else:
            nb = lib().ipsx_projector_workspace_bytes(n)
            ws = self._workspace(nb, x.device)
            _ck(lib().ipsx_projector(C.byref(self.lin), _p(x), n, C.c_float(self.ln_eps), _p(out),
                                     _p(ws), nb, _stream()), "ipsx_projector")
        return out


def encoder_kernel_name(plan):
    """Which kernel family the plan's encode() launches (for bench.py's roofline record)."""
    if plan is None or plan._sig is None:
        return None
    if not plan.is_image:
        return "row_stats_kernel + conv_nhwc_kernel<NORM> (projector)"
    return lib().ipsx_trunk_kernel(C.byref(plan.trunk)).decode()
