"""ctypes front-end of the CPU oracle (oracle/ips_oracle.cpp).

TEST INFRASTRUCTURE - imported only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Never by ips_amd/.

``Oracle(net)`` walks an ``IPSNet`` (this repo's or the reference's - the
attribute names are the same, /root/reference/architecture/ips_net.py:85-116),
copies its weights to numpy and replays ``ips()`` / ``forward()`` with the
canonical arithmetic of the C++ restatement.
"""

import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

f32p = C.POINTER(C.c_float)
i64p = C.POINTER(C.c_int64)
i32p = C.POINTER(C.c_int32)


class _Conv(C.Structure):
    _fields_ = [("c_in", C.c_int), ("c_out", C.c_int), ("kh", C.c_int), ("kw", C.c_int),
                ("stride", C.c_int), ("pad", C.c_int),
                ("w", f32p), ("alpha", f32p), ("shift", f32p)]


class _Block(C.Structure):
    _fields_ = [("n_conv", C.c_int), ("conv", _Conv * 3), ("has_down", C.c_int), ("down", _Conv)]


class _Trunk(C.Structure):
    _fields_ = [("c_in", C.c_int), ("h", C.c_int), ("w", C.c_int), ("stem", _Conv),
                ("n_block", C.c_int), ("blocks", C.POINTER(_Block))]


class _Transf(C.Structure):
    _fields_ = [("n_token", C.c_int), ("h", C.c_int), ("d", C.c_int), ("dk", C.c_int),
                ("dv", C.c_int), ("d_inner", C.c_int)] + \
               [(n, f32p) for n in ("q", "wq", "wk", "wv", "fc", "ln1_g", "ln1_b", "w1", "b1",
                                    "w2", "b2", "ln2_g", "ln2_b")] + \
               [("temperature", C.c_float), ("ln_eps", C.c_float)]


def build(force=False):
    so = os.path.join(_HERE, "libipsoracle.so")
    src = os.path.join(_HERE, "ips_oracle.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_expf.restype = C.c_float
        _LIB.orc_expf.argtypes = [C.c_float]
        _LIB.orc_wave_sum64.restype = C.c_float
        _LIB.orc_wave_sum64.argtypes = [f32p, C.c_int64]
    return _LIB


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(f32p)


def _np(t):
    return np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float32))


def expf(x):
    return float(lib().orc_expf(C.c_float(float(x))))


def wave_sum64(x):
    a, p = _f(x)
    return float(lib().orc_wave_sum64(p, a.size))


def bn_affine(bn):
    g, b = _np(bn.weight), _np(bn.bias)
    m, v = _np(bn.running_mean), _np(bn.running_var)
    alpha, shift = np.empty_like(g), np.empty_like(g)
    lib().orc_bn_affine(_f(g)[1], _f(b)[1], _f(m)[1], _f(v)[1], C.c_float(bn.eps), g.size,
                        alpha.ctypes.data_as(f32p), shift.ctypes.data_as(f32p))
    return alpha, shift


def topm(scores, M, aten_ties=False, rows=None):
    """``torch.topk(scores, M)[1]`` of one row: the canonical tie rule (earlier position first), ``aten_ties``: torch's
    CPU order wherever scores tie (orc_topm_aten) - or, with ``rows`` = the candidates' (L, R) attention logits, the
    selection LOOP's rule (orc_topm_loop): torch's order when two tied neighbours among the first M + 1 ranks are
    bit-identical rows, the canonical order otherwise."""
    s, sp = _f(scores)
    top = np.empty(M, dtype=np.int64)
    tie = C.c_int32(0)
    if aten_ties and rows is not None:
        r, rp = _f(rows)
        assert r.ndim == 2 and r.shape[0] == s.size
        lib().orc_topm_loop(sp, rp, s.size, r.shape[1], M, top.ctypes.data_as(i64p), C.byref(tie))
    elif aten_ties:
        lib().orc_topm_aten(sp, s.size, M, top.ctypes.data_as(i64p))
    else:
        lib().orc_topm(sp, s.size, M, top.ctypes.data_as(i64p), C.byref(tie))
    return top, tie.value


class Oracle:
    def __init__(self, net):
        self.L = lib()
        self.net = net
        self._keep = []           # numpy arrays referenced by the C structs
        self.M, self.I, self.D = net.M, net.I, net.D
        self.use_pos = net.use_pos
        self.is_image = net.is_image
        ca = net.transf.crs_attn
        self.T, self.H, self.Dk, self.Dv = ca.n_token, ca.H, ca.D_k, ca.D_v
        self.temperature = float(ca.attention.temperature)
        self.q = _np(ca.q[0])
        self.wq, self.wk = _np(ca.q_w.weight), _np(ca.k_w.weight)
        self.qs = np.empty((self.T, self.H * self.Dk), dtype=np.float32)
        self.L.orc_query_proj(_f(self.q)[1], _f(self.wq)[1], C.c_float(self.temperature), self.T,
                              self.D, self.H * self.Dk, self.qs.ctypes.data_as(f32p))
        if self.is_image:
            self.trunk = self._trunk(net.encoder)
        else:
            lin, bn = net.encoder[1], net.encoder[2]
            self.proj = dict(w=_np(lin.weight), b=_np(lin.bias), eps=float(net.encoder[0].eps))
            self.proj["alpha"], self.proj["shift"] = bn_affine(bn)
        self.transf = self._transf(net.transf)

    # ---- descriptors
    def _hold(self, a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        self._keep.append(a)
        return a.ctypes.data_as(f32p)

    def _conv(self, conv, bn):
        alpha, shift = bn_affine(bn)
        w = _np(conv.weight)
        return _Conv(conv.in_channels, conv.out_channels, conv.kernel_size[0], conv.kernel_size[1],
                     conv.stride[0], conv.padding[0], self._hold(w), self._hold(alpha), self._hold(shift))

    def _trunk(self, enc):
        mods = list(enc.children())
        stem_conv, stem_bn = mods[0], mods[1]
        blocks = []
        for stage in mods[4:-1]:
            for blk in stage.children():
                b = _Block()
                names = [("conv1", "bn1"), ("conv2", "bn2"), ("conv3", "bn3")]
                convs = [(getattr(blk, c), getattr(blk, n)) for c, n in names if hasattr(blk, c)]
                b.n_conv = len(convs)
                for j, (cv, bn) in enumerate(convs):
                    b.conv[j] = self._conv(cv, bn)
                b.has_down = 0 if blk.downsample is None else 1
                if b.has_down:
                    b.down = self._conv(blk.downsample[0], blk.downsample[1])
                blocks.append(b)
        arr = (_Block * len(blocks))(*blocks)
        self._keep.append(arr)
        t = _Trunk()
        t.stem = self._conv(stem_conv, stem_bn)
        t.c_in = stem_conv.in_channels
        t.n_block = len(blocks)
        t.blocks = C.cast(arr, C.POINTER(_Block))
        return t

    def _transf(self, tr):
        ca, mlp = tr.crs_attn, tr.mlp
        t = _Transf()
        t.n_token, t.h, t.d, t.dk, t.dv = ca.n_token, ca.H, self.D, ca.D_k, ca.D_v
        t.d_inner = mlp.w_1.out_features
        for name, src in (("q", ca.q[0]), ("wq", ca.q_w.weight), ("wk", ca.k_w.weight),
                          ("wv", ca.v_w.weight), ("fc", ca.fc.weight),
                          ("ln1_g", ca.layer_norm.weight), ("ln1_b", ca.layer_norm.bias),
                          ("w1", mlp.w_1.weight), ("b1", mlp.w_1.bias), ("w2", mlp.w_2.weight),
                          ("b2", mlp.w_2.bias), ("ln2_g", mlp.layer_norm.weight),
                          ("ln2_b", mlp.layer_norm.bias)):
            setattr(t, name, self._hold(_np(src)))
        t.temperature = self.temperature
        t.ln_eps = float(ca.layer_norm.eps)
        return t

    # ---- pieces
    def encode(self, x):
        """(P,C,h,w) or (P,F) -> (P,D)"""
        x, xp = _f(x)
        P = x.shape[0]
        out = np.empty((P, self.D), dtype=np.float32)
        if self.is_image:
            self.trunk.h, self.trunk.w = x.shape[2], x.shape[3]
            self.L.orc_trunk_encode(C.byref(self.trunk), xp, C.c_int64(P), out.ctypes.data_as(f32p))
        else:
            p = self.proj
            self.L.orc_projector(xp, C.c_int64(P), x.shape[1], self.D, C.c_float(p["eps"]),
                                 _f(p["w"])[1], _f(p["b"])[1], _f(p["alpha"])[1], _f(p["shift"])[1],
                                 out.ctypes.data_as(f32p))
        return out

    def logits(self, emb, pos=None):
        emb, ep = _f(emb)
        n = emb.shape[0]
        out = np.empty((n, self.H * self.T), dtype=np.float32)
        pp = _f(pos)[1] if pos is not None else None
        self.L.orc_logits(ep, pp, _f(self.wk)[1], _f(self.qs)[1], C.c_int64(n), self.D, self.H,
                          self.Dk, self.T, out.ctypes.data_as(f32p))
        return out

    def scores(self, x, want_attn=False):
        """Transformer.get_scores for x (L,D)"""
        x, xp = _f(x)
        L = x.shape[0]
        sc = np.empty(L, dtype=np.float32)
        attn = np.empty((self.H, self.T, L), dtype=np.float32) if want_attn else None
        self.L.orc_scores(xp, _f(self.wk)[1], _f(self.qs)[1], L, self.D, self.H, self.Dk, self.T,
                          sc.ctypes.data_as(f32p), attn.ctypes.data_as(f32p) if want_attn else None)
        return (sc, attn) if want_attn else sc

    def scan(self, emb, pos=None, aten_ties=False):
        """emb (B,N,D), pos (B,N,D)|None -> dict(mem_idx (B,M), trace_idx, trace_score, tie)"""
        emb = np.ascontiguousarray(emb, dtype=np.float32)
        B, N, D = emb.shape
        M, I = self.M, self.I
        n_iter = math.ceil((N - M) / I)
        mem = np.empty((B, M), dtype=np.int64)
        tr_i = np.empty((B, n_iter, M), dtype=np.int64)
        tr_s = np.empty((B, n_iter, M), dtype=np.float32)
        tie = np.zeros((B, n_iter), dtype=np.int32)
        for b in range(B):
            pb = None
            if pos is not None:
                pb = np.ascontiguousarray(np.broadcast_to(pos, emb.shape)[b], dtype=np.float32)
            self.L.orc_ips_scan(_f(emb[b])[1], _f(pb)[1] if pb is not None else None,
                                _f(self.wk)[1], _f(self.qs)[1], C.c_int64(N), D, self.H, self.Dk,
                                self.T, M, I, int(aten_ties), mem[b].ctypes.data_as(i64p),
                                tr_i[b].ctypes.data_as(i64p), tr_s[b].ctypes.data_as(f32p),
                                tie[b].ctypes.data_as(i32p))
        return dict(mem_idx=mem, trace_idx=tr_i, trace_score=tr_s, tie=tie)

    def ips(self, patches, pos_enc=None, aten_ties=False):
        """patches (B,N,...) numpy in the order ips() sees them after any shuffle."""
        patches = np.ascontiguousarray(patches, dtype=np.float32)
        B, N = patches.shape[:2]
        emb = self.encode(patches.reshape(B * N, *patches.shape[2:])).reshape(B, N, self.D)
        pos = None
        if self.use_pos:
            pos = np.broadcast_to(np.asarray(pos_enc, dtype=np.float32), (B, N, self.D))
        out = self.scan(emb, pos, aten_ties=aten_ties)
        idx = out["mem_idx"]
        out["emb"] = emb
        out["mem_patch"] = np.stack([patches[b][idx[b]] for b in range(B)])
        out["mem_pos"] = np.stack([pos[b][idx[b]] for b in range(B)]) if self.use_pos else None
        return out

    def tournament(self, patches, pos_enc, world):
        """The north star's literal multi-GPU scheme, restated: the patch axis is cut into
        `world` contiguous slabs, every slab runs the full selection loop on its own (local memory of M), and one final
        top-M step over the world*M slab winners (slab order, each slab's winners in its final order) picks the
        result.  NOT the single-device selection (softmax denominators differ per slab) - this is the checker of
        ips_amd.dist.ips_tournament only.  Returns mem_idx (B, M) global indices."""
        patches = np.ascontiguousarray(patches, dtype=np.float32)
        B, N = patches.shape[:2]
        M = self.M
        emb = self.encode(patches.reshape(B * N, *patches.shape[2:])).reshape(B, N, self.D)
        pos = np.broadcast_to(np.asarray(pos_enc, dtype=np.float32), (B, N, self.D)) if self.use_pos else None
        per = -(-N // world)
        cand = []
        for r in range(world):
            lo, hi = min(r * per, N), min((r + 1) * per, N)
            if hi - lo <= M:                                   # a slab no larger than the memory keeps everything
                cand.append(np.broadcast_to(np.arange(lo, hi, dtype=np.int64), (B, hi - lo)))
                continue
            loc = self.scan(emb[:, lo:hi], pos[:, lo:hi] if pos is not None else None)["mem_idx"]
            cand.append(loc + lo)
        cand = np.concatenate(cand, axis=1)                    # (B, <= world*M)
        out = np.empty((B, M), dtype=np.int64)
        for b in range(B):
            x = emb[b][cand[b]] + (pos[b][cand[b]] if pos is not None else 0)
            top, _ = topm(self.scores(x), M)
            out[b] = cand[b][top]
        return out

    def aggregate(self, x):
        """Transformer.forward: x (B,M,D) -> (B,T,D)"""
        x = np.ascontiguousarray(x, dtype=np.float32)
        B, M, _ = x.shape
        out = np.empty((B, self.T, self.D), dtype=np.float32)
        for b in range(B):
            self.L.orc_aggregate(C.byref(self.transf), _f(x[b])[1], M, out[b].ctypes.data_as(f32p))
        return out

    def forward(self, mem_patch, mem_pos=None):
        """IPSNet.forward in eval mode -> {task: (B,n_class)}"""
        mem_patch = np.ascontiguousarray(mem_patch, dtype=np.float32)
        B, M = mem_patch.shape[:2]
        emb = self.encode(mem_patch.reshape(B * M, *mem_patch.shape[2:])).reshape(B, M, self.D)
        if mem_pos is not None:
            emb = emb + np.asarray(mem_pos, dtype=np.float32)
        agg = self.aggregate(emb)
        preds = {}
        for task in self.net.tasks.values():
            head = self.net.output_layers[task['name']][0]
            w, bias = _np(head.weight), _np(head.bias)
            act = 0 if task['act_fn'] == 'softmax' else 1
            out = np.empty((B, w.shape[0]), dtype=np.float32)
            for b in range(B):
                self.L.orc_head(_f(agg[b, task['id']])[1], self.D, _f(w)[1], _f(bias)[1], w.shape[0],
                                act, out[b].ctypes.data_as(f32p))
            preds[task['name']] = out
        return preds
