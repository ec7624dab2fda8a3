"""How ``IPSNet.ips`` runs on a ROCm device: encode -> logits -> selection loop -> ``mem_idx``.

The reference runs a Python loop of ~75 stock kernels per chunk (architecture/ips_net.py:213-241).  In eval / no-grad
mode a patch's embedding and its attention logits depend on that patch alone, so the HIP path encodes and scores every
patch once and replays the loop on the cached logits.  This module is the scheduling around those
kernels - ONE pipeline with two choices:

  producer   who turns patches into (embeddings, logits), and in which pieces
               ``parts``    a list of launches over slices / index lists of the patch axis (any encoder)
               ``stream``   one persistent launch that publishes rows as they complete (the fused 1x32x32 trunk for one
                            image, the projector for feature slides)
  loop       how the selection loop follows the producer
               ``persistent``  launched once, up front, on a compute unit of its own; waits (bounded) for published rows
               ``ranges``      ``scan_range`` per part on a high-priority side stream (resumes from ``mem_idx``)
               ``after``       one ``scan`` behind the last part (lazy slabs run ranges beside the copies)

Every choice computes the same per-patch arithmetic, so all of them select the same patches
(tests/test_hip_e2e.py::test_every_variant_..., test_one_image_every_schedule_...).

Launch sizes come from the device (``hip.device_geometry``): a producer that runs beside resident loops is sized for the
compute units they leave free on the FULLEST XCD (workgroups are dealt to XCDs round-robin), the fused trunk's
"round" is 8 patches per compute unit.  Whether persistent kernels can run beside their producers at all is established
ONCE per device by a self-test (``hip.persistent_ok``: under counter collection or a serialising debug switch they
cannot), checked again when a loop times out, and revoked for the process when that check fails or loops keep timing out
(``hip.persistent_timed_out``).
"""

import contextlib
import ctypes as C
import gc
import math
import os

import torch

from . import hip


@contextlib.contextmanager
def _no_gc_pause():
    """Between the launch of a resident loop and the launch of its producers the host must not stall: the loop gives up
    after 50 ms without progress (the call is then redone, results valid - but it costs 50 ms and counts as a strike).
    What did stall it, once in ~5,000 calls of a soak run (tools/soak.py, profiles/r05_soak.txt): a full collection of the
    interpreter's cyclic garbage collector, 40-60 ms in a process that has torch loaded, landing between the two launches.
    The collector is held off for those few lines (an allocation-count trigger is only postponed)."""
    was = gc.isenabled()
    if was:
        gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


def _env_on(name):
    return os.environ.get(name, "1") != "0"


class Selection:
    """The selection pipelines of one ``IPSNet`` (buffers and streams are kept between calls of the same shape: a buffer
    that a side stream has used cannot be recycled by the allocator until that stream's work is known to be over, and
    allocating afresh in every call sends a host that runs ahead of the GPU back to the driver)."""

    OVERLAP_PARTS = 4
    LAZY_SLAB_BYTES = 48 << 20

    def __init__(self, net):
        self.net = net
        self._bufs = {}
        self._side = None
        self._copy = None
        self.scan_status = None          # status word of the last persistent loop (device), its host mirror
        self.scan_status_host = None
        self._mirror_pending = None
        self._unfinished = None
        self._done = None                          # (mem_idx, mem_patch, mem_pos) of a call the library enqueued whole
        self._calls = {}                           # hip.IpsCall structures, per pipeline
        self.timing_hook = None                    # callable(rows) -> slot of a library-owned event pair around the producer, or None
        self._part_index = None

    # ------------------------------------------------------------------ small helpers
    def plan(self):
        net = self.net
        if net._plan is None:
            net._plan = hip.EncoderPlan(net.encoder, net.is_image)
        return net._plan

    def n_iter(self, N):
        return math.ceil((N - self.net.M) / self.net.I)

    def streams(self, dev):
        if self._side is None or self._side.device != torch.device(dev):
            self._side = hip.side_stream(dev)      # (one per device and process, high priority: hip.side_stream)
        return self._side, torch.cuda.current_stream(dev)

    def buffers(self, name, key, make):
        """Per-call device buffers of pipeline ``name``, re-made only when ``key`` (the shape) changes."""
        held = self._bufs.get(name)
        if held is None or held[0] != key:
            bufs = make()
            side = self._side
            for t in bufs:
                if t is not None and side is not None:
                    t.record_stream(side)
            self._bufs[name] = held = (key, bufs)
        return held[1]

    def persistent_allowed(self, dev, M, I, H, T, loops=1, large=False):
        """Resident loops beside their producers: switched on, the shape is covered, the device has units to spare for
        ``loops`` of them (a small partition: the producers need the units more) and it has been SEEN to work there."""
        return (_env_on("IPSX_SCAN_PERSIST") and _env_on("IPSX_OVERLAP_SCAN") and not hip.dedup_blank()
                and (hip.scan_persistent_supported(M, I, H, T) or (large and hip.scan_persistent_large(M, I, H, T)))
                and hip.device_geometry(dev).cus >= 8 * max(1, loops)
                and hip.persistent_ok(dev))

    # ------------------------------------------------------------------ the persistent loop: begin / end
    def persistent_begin(self, logits, mem_idx_buf, zeroed, B, dev, loops=0, scan_ws=None, zero_words=None):
        """Zero the call's words (tie flags | a progress word per image | status | producer control words: ONE fill), launch
        the loop on the side stream and hold the main stream until it is resident.  -> (tie, ready, status, ctl).

        A loop that gave up waiting (no progress for hip.persistent_wait_ms: something serialises the kernels, so that its
        producers cannot run beside it) is REDONE in the same call by the conditional launch of ``persistent_end``, so a
        call's results are valid either way and no host synchronisation is added.  The status word is mirrored into pinned
        host memory, asynchronously, and looked at in the NEXT call - by then it has long arrived: a timeout re-runs the
        device's self-test, and repeated timeouts (``hip.persistent_timed_out``: IPSX_PERSIST_STRIKES, default 3) or a failing
        self-test switch the persistent pipelines off for the rest of the process (the per-part launches take over)."""
        # (first: everything below is host work in front of the call's second launch; zero_words: the projector stream's
        #  control words end in hand-over accumulators that need no zeroing)
        (zeroed if zero_words is None else zeroed[:zero_words]).zero_()
        hip._PERSIST_CALLS += 1
        net = self.net
        ca = net.transf.crs_attn
        mirror = self.scan_status_host
        if mirror is not None and int(mirror.item()) & 1:
            mirror.zero_()
            hip.persistent_timed_out(dev)          # (self-test again; off for the process only after repeated events)
        side, main = self.streams(dev)
        tie, words, ctl = zeroed[:B], zeroed[B:2 * B + 1], zeroed[2 * B + 1:]
        ready, status = words[:B], words[B:B + 1]
        self.scan_status = status
        side.wait_stream(main)                     # the buffers are the main stream's; previous readers are done
        hip.scan_persistent(logits, net.M, net.I, ca.H, ca.n_token, mem_idx_buf, tie, ready, status, workgroups=loops,
                            workspace=scan_ws, stream=side)
        # producers must not take the compute units before a loop has its own.  (Also true of a persistent producer whose
        # workgroups sit one to a unit and leave a unit per loop free: a workgroup is dealt to an XCD before it looks for a
        # unit there, so a loop that arrives second may be dealt to a FULL XCD and start when the producer ends - measured
        # without the gate: 31 M patches/s with calls back to back against 43 M, one CAMELYON slide.)
        hip.scan_gate(status)
        return tie, ready, status, ctl

    def persistent_end(self, logits, mem_idx_buf, tie, status, n_iter, dev, scan_ws=None):
        net = self.net
        ca = net.transf.crs_attn
        side, main = self.streams(dev)
        main.wait_stream(side)
        hip.scan_range_if(logits, net.M, net.I, ca.H, ca.n_token, 0, n_iter, mem_idx_buf, tie, status, 1,
                          workspace=scan_ws)                                                           # no-op unless timed out
        self._mirror_pending = status              # (copied to the host by after_call: behind the call's gathers)
        hip.scan.last_tie = tie
        # the loop's own buffer - overwritten by the next call: ``finish`` (one launch: both gathers, a fresh copy of these
        # indices, the status word to the host) or ``take_unfinished`` + a copy is the caller's next step
        self._unfinished = mem_idx_buf
        return mem_idx_buf

    def finish(self, src, pos):
        """The end of an ``ips()`` call whose loop ran resident, as ONE launch (``hip.ips_finish``: both gathers, a fresh copy
        of the selected indices, the loop's status word to its pinned host mirror) -> (mem_idx, mem_patch, mem_pos), or None
        when this call has nothing of the kind pending or the tensors' rows are not whole 16-byte units (the caller then
        gathers itself and ``after_call`` mirrors the status word)."""
        done, self._done = self._done, None
        if done is not None:
            return done
        buf, self._unfinished = self._unfinished, None
        if buf is None:
            return None
        status = self._mirror_pending
        if status is None or not hip.ips_finish_supported(src, pos):
            self._unfinished = buf
            return None
        if self.scan_status_host is None:
            self.scan_status_host = torch.zeros((1,), dtype=torch.int32).pin_memory()
        self._mirror_pending = None
        return hip.ips_finish(src, pos, buf, status, self.scan_status_host)

    def take_unfinished(self, mem_idx):
        """``mem_idx`` as a tensor of the caller's own (the loop's buffer is overwritten by the next call)."""
        buf, self._unfinished = self._unfinished, None
        return mem_idx.clone() if buf is not None and mem_idx is buf else mem_idx

    def native_ok(self, src, pos):
        """Can the library enqueue this call whole (``ipsx_ips_call_run``)?  Needs what ``ips_finish`` needs of the tensors."""
        return _env_on("IPSX_NATIVE_CALL") and hip.ips_finish_supported(src, pos)

    def native_call(self, name, patches, pos_enc, logits, mem_idx_buf, zeroed, emb_buf, scan_ws, loops, wgs, trunk_pos=None,
                    quad_pulls=-1, short_first=-1):
        """One ips() call with a resident loop, enqueued by ONE library call (csrc/call.hip): fill, loop on the side stream,
        gate, producer (the fused trunk for one image / the projector for feature slides), conditional recovery, and the end
        of the call (both gathers, indices, status word to the host).  Same kernels and results as the launches one by one
        (``IPSX_NATIVE_CALL=0``); the host's share of a call is one ctypes call, and nothing of the interpreter sits between
        the loop's launch and its producer's.  -> mem_idx (fresh); ``finish`` hands out (mem_idx, mem_patch, mem_pos)."""
        net, plan = self.net, self.plan()
        B, N = patches.shape[:2]
        dev = patches.device
        ca = net.transf.crs_attn
        vq, R = ca.folded_query(), ca.H * ca.n_token
        mirror = self.scan_status_host
        if mirror is None:
            mirror = self.scan_status_host = torch.zeros((1,), dtype=torch.int32).pin_memory()
        elif int(mirror.item()) & 1:
            mirror.zero_()
            hip.persistent_timed_out(dev)          # (self-test again; off for the process only after repeated events)
        hip._PERSIST_CALLS += 1
        side, _ = self.streams(dev)
        plan._refresh()
        c = self._calls.get(name)
        if c is None or c[0] is not logits:
            call = hip.IpsCall()
            call.b, call.n, call.m, call.i, call.h, call.n_token = B, N, net.M, net.I, ca.H, ca.n_token
            call.logits, call.mem_idx = logits.data_ptr(), mem_idx_buf.data_ptr()
            call.words, call.words_total = zeroed.data_ptr(), zeroed.numel()
            call.loops = loops
            call.scan_workspace = scan_ws.data_ptr() if scan_ws is not None else None
            call.scan_workspace_bytes = scan_ws.numel() if scan_ws is not None else 0
            call.emb, call.r, call.workgroups = emb_buf.data_ptr(), R, wgs
            call.quad_pulls, call.short_first = quad_pulls, short_first
            call.status_host = mirror.data_ptr()
            call.side_stream = side.cuda_stream
            c = self._calls[name] = (logits, call, (mem_idx_buf, zeroed, emb_buf, scan_ws, mirror))
        call = c[1]
        if net.is_image:
            call.trunk, call.lin = C.pointer(plan.trunk), None
            call.pos = trunk_pos.data_ptr() if trunk_pos is not None else None
        else:
            call.trunk, call.lin, call.ln_eps = None, C.pointer(plan.lin), plan.ln_eps
        call.x = patches.data_ptr()
        call.v_packed = vq.data_ptr()
        M = net.M
        mem_patch = torch.empty((B, M) + tuple(patches.shape[2:]), dtype=patches.dtype, device=dev)
        mem_idx = torch.empty((B, M), dtype=torch.int64, device=dev)
        mem_pos = None
        call.src = patches.data_ptr()
        call.src_row_bytes = patches[0, 0].numel() * patches.element_size()
        call.src_bstride_rows = N if B > 1 else 0
        if pos_enc is not None:
            mem_pos = torch.empty((B, M, pos_enc.shape[2]), dtype=pos_enc.dtype, device=dev)
            call.pos_table = pos_enc.data_ptr()
            call.pos_row_bytes = pos_enc.shape[2] * pos_enc.element_size()
            call.pos_bstride_rows = 0 if (pos_enc.stride(0) == 0 or B == 1) else N
            call.mem_pos = mem_pos.data_ptr()
        else:
            call.pos_table, call.pos_row_bytes, call.pos_bstride_rows, call.mem_pos = None, 0, 0, None
        call.mem_patch, call.mem_idx_out = mem_patch.data_ptr(), mem_idx.data_ptr()
        call.timing_slot = self.timing_hook(B * N) if self.timing_hook is not None else -1
        call.stream = hip._stream().value
        hip._ck(hip.lib().ipsx_ips_call_run(C.byref(call)), "ipsx_ips_call_run")
        tie, status = zeroed[:B], zeroed[2 * B:2 * B + 1]
        self.scan_status = status
        hip.scan.last_tie = tie
        net._emb_parts = [emb_buf]
        self._done = (mem_idx, mem_patch, mem_pos)
        return mem_idx

    def after_call(self):
        """The last thing an ``ips()`` call enqueues: the status word of its persistent loop goes to its pinned host mirror
        (looked at in the NEXT call) - behind the gathers of the selected patches, not in front of them."""
        status = self._mirror_pending
        if status is not None:
            self._mirror_pending = None
            if self.scan_status_host is None:
                self.scan_status_host = torch.zeros((1,), dtype=torch.int32).pin_memory()
            self.scan_status_host.copy_(status, non_blocking=True)

    # ------------------------------------------------------------------ which pipeline
    def select(self, patches, pos_enc):
        """(B, N, ...) patches (device, or host for lazy loading) -> mem_idx (B, M) int64 on the device."""
        net = self.net
        net._device_patches = None
        self._done = self._unfinished = None       # (whatever a call that raised half-way left behind)
        if patches.is_cuda and self.can_stream_image(patches):
            return self.image_stream(patches, pos_enc)
        if patches.is_cuda and self.can_overlap(patches):
            ca = net.transf.crs_attn
            if (not net.is_image and patches.is_contiguous() and patches.shape[0] <= int(os.environ.get("IPSX_PERSIST_MAX_B", "16"))
                    and self.persistent_allowed(patches.device, net.M, net.I, ca.H, ca.n_token, self.feature_loops(patches.shape[0]),
                                                large=_env_on("IPSX_LARGE_PERSIST"))):
                return self.features_persistent(patches, pos_enc)
            return self.parts_with_ranges(patches, pos_enc)
        return self.slabs(patches, pos_enc)

    def can_overlap(self, patches):
        """Does the loop run beside the encoder (parts + ranges, or a persistent loop)?"""
        net = self.net
        if not _env_on("IPSX_OVERLAP_SCAN") or hip.dedup_blank() or net.encoder.training:
            return False
        B, N = patches.shape[:2]
        n_iter = self.n_iter(N)
        if net.is_image and B * N < self.small_batch_limit(patches.device) and n_iter < 100:
            # a small batch does not fill the GPU four times over: it is encoded in the whole rounds of the fused trunk it
            # fills plus the remainder, the loop over the first part beside the remainder's encoding (small_batch_split);
            # a layer-by-layer trunk cut in two just runs every layer twice at half the occupancy (measured: traffic
            # signs 25.3 ms against 22.7 ms in one piece)
            return self.plan().fused(patches.shape) and self.small_batch_split(B, N, patches.device) is not None
        # (feature inputs: the loop is the long pole whatever its length - a slide at the reference's shipped M = I = 5000
        #  has 7 iterations of 10,000 candidates - so any loop of a few iterations runs beside the projector's later parts)
        return n_iter >= (2 * self.OVERLAP_PARTS if net.is_image else 3)

    def can_stream_image(self, patches):
        """ONE image on the fused fp32 1x32x32 trunk: trunk + logits as one persistent launch beside a resident loop."""
        net = self.net
        if (not net.is_image or patches.shape[0] != 1 or net.encoder.training or not patches.is_contiguous()
                or not _env_on("IPSX_IMAGE_STREAM") or patches.shape[1] < net.M + 2 * net.I):
            return False
        ca = net.transf.crs_attn
        return (self.persistent_allowed(patches.device, net.M, net.I, ca.H, ca.n_token)
                and self.plan().image_stream_supported(patches.shape, net.D, ca.H * ca.n_token)
                and ca.folded_query().dtype == torch.float32)

    # ------------------------------------------------------------------ sizes that come from the device
    @staticmethod
    def round_patches(dev):
        """Patches of one full round of the fused trunk: 8 per compute unit (two workgroups of four wavefronts = patches)."""
        return 8 * hip.device_geometry(dev).cus

    def small_batch_limit(self, dev):
        return 16 * self.round_patches(dev)

    def free_units(self, dev, loops):
        """Compute units a single-round launch can count on beside ``loops`` resident loop workgroups: workgroups are
        dealt to the XCDs round-robin whatever is free there, so it is the free units of the FULLEST XCD, times the XCDs."""
        g = hip.device_geometry(dev)
        return g.xcds * (g.cus_per_xcd - -(-loops // g.xcds))

    def small_batch_split(self, B, N, dev):
        """How to cut a small image batch (one round of the fused trunk and a remainder): (edges, its) - part k encodes rows
        edges[k]..edges[k+1] of every image and the loop then runs iterations its[k]..its[k+1], those whose rows are
        encoded by then - or None when that leaves nothing on either side.  The parts are the ENCODER's units: half a round
        first (one wavefront per SIMD), then half a round less one workgroup per XCD and shader-engine pair - a compute
        unit stays free for the loop of the part before wherever the dispatcher lands (on a unit it shares with fp32 MFMA
        wavefronts an iteration takes 24 us instead of 6.7, tools/scan_beside.py) - then the rest, which goes through the
        two-wavefronts-per-patch kernel.  On 256 units: 1,024 + 992 + 484 patches of one 2,500-patch image, iterations
        15 + 15 + 9.  Measured (bench.py --config b1 / --batch 2): +8 % at one round + remainder, -4 % at two rounds."""
        M, I = self.net.M, self.net.I
        n_iter = self.n_iter(N)
        rnd = self.round_patches(dev)
        if (B * N) // rnd != 1:
            return None
        half = rnd // 2
        g = hip.device_geometry(dev)
        edges, its = [0], [0]
        for total in (half, half + half - 4 * g.xcds):
            e = total // B
            it = (e - M) // I                                  # iterations whose rows lie inside the first e of every image
            if e < N and its[-1] < it < n_iter:
                edges.append(e)
                its.append(it)
        if len(its) == 1:
            return None
        return edges + [N], its + [n_iter]

    def feature_loops(self, B):
        """Resident loop workgroups of a call of B slides: one per slide, or - shapes whose loop kernel takes its slides
        in turn (ipsx_scan_persistent_on) - two for any number of slides."""
        net = self.net
        ca = net.transf.crs_attn
        if B > 2 and hip.scan_persistent_groupable(net.M, net.I, ca.H, ca.n_token):
            return max(1, min(B, int(os.environ.get("IPSX_CAM_LOOPS", "2"))))
        return B

    def feature_parts(self, B, N, dev, persistent):
        """Iterations at which the rows of ONE slide are cut into projector launches.  Persistent loops (each owns a compute
        unit, the projector goes slide by slide): equal parts that fill the other units exactly once; otherwise every
        launch takes its rows of all B slides and is sized to fill every unit once."""
        M, I = self.net.M, self.net.I
        n_iter = self.n_iter(N)
        rows = self.free_units(dev, B) * 64 if persistent else hip.device_geometry(dev).cus * 64 // max(B, 1)
        cap = max(I, rows // I * I)                          # most rows of a slide one launch can take, whole chunks
        n_part = min(16, max(1, math.ceil(N / cap)))
        its = [0]
        for k in range(1, n_part):                           # equal parts: edge k at about k * N / n_part rows
            nxt = max(its[-1] + 1, round((k * N / n_part - M) / I))
            if nxt >= n_iter:
                break
            its.append(nxt)
        its.append(n_iter)
        ca = self.net.transf.crs_attn
        if not persistent and n_iter >= 3 and hip.lib().ipsx_scan_workspace_bytes(1, M, I, ca.H, ca.n_token) > 0:
            # a candidate set beyond the LDS (the shipped CAMELYON M = I = 5000: 7 iterations of 0.2 ms): the loop is the
            # long pole and every further launch costs it a hand-over - TWO parts: the rows of the first two iterations,
            # then all the rest beside them (measured at 38,000 rows: parts 2+2+3 17.1 M patches/s, 2+5 17.7, 1+2+4 17.1,
            # 1+6 16.8, 3+4 16.6)
            its = [0, 2, n_iter]
        return its

    def feature_launches(self, B, N, edges, dev, loops=None):
        """The projector launches of ``features_persistent``: (first row, end row) in the FLAT (B * N) row space + what each
        makes visible, [(slide, rows)].  One slide, or positional encodings (a table per slide position): a slide's parts.
        Several slides without them: the slides are one stream of rows cut into full launches wherever a slide ends (the
        patch tensor is contiguous).  A full launch: the free units of the fullest XCD less one per XCD (measured on 256
        units at 2 / 16 slides: 208 workgroups 36.3 / 43.1 M patches/s, 224: 36.5 / 45.9, 240: 30.4 / 37.8 - now and then a
        workgroup of a fuller launch waits for a second round)."""
        I = self.net.I
        P = len(edges) - 1
        if B == 1 or self.net.use_pos:
            return [(b_ * N + edges[k], b_ * N + edges[k + 1], [(b_, edges[k + 1])]) for b_ in range(B) for k in range(P)]
        g = hip.device_geometry(dev)
        cap = max(I, min(g.cus - 4 * g.xcds, self.free_units(dev, loops or B)) * 64 // I * I)
        launches, r0 = [], 0
        while r0 < B * N:
            # the first launch of the call stays short: the first loop starts after M + I rows' worth of projector
            r1 = min(B * N, r0 + (cap if r0 > 0 else min(cap, edges[1])))
            pubs = [(b_, min(N, r1 - b_ * N)) for b_ in range(r0 // N, (r1 - 1) // N + 1)]
            launches.append((r0, r1, pubs))
            r0 = r1
        return launches

    # ------------------------------------------------------------------ pipeline: feature slides, persistent loops
    def features_persistent(self, patches, pos_enc):
        """Feature inputs, up to IPSX_PERSIST_MAX_B slides: every slide's loop is resident from the start of the call and
        follows its own progress word; the projector works through the slides one after the other, so the loop of slide b
        runs beside the projector of slide b + 1.  Without positional encoding and with fp32 logits the projector is ONE
        persistent launch (ipsx_projector_stream: moments + Linear + logits per tile, tiles published as they complete,
        across the slides' ends); otherwise launch by launch - per part the GEMM (LayerNorm in its operand load; its first
        thread publishes what was enqueued before it) and the logits of the part together with the row moments of the next."""
        net, plan = self.net, self.plan()
        B, N = patches.shape[:2]
        M, I, dev = net.M, net.I, patches.device
        ca = net.transf.crs_attn
        vq, R = ca.folded_query(), ca.H * ca.n_token
        n_iter = self.n_iter(N)
        self.streams(dev)
        logits, mem_idx_buf, zeroed, stats, emb_buf, scan_ws = self.buffers(
            "features", (B, N, M, I, R, net.D, str(dev)),
            lambda: (torch.empty((B, N, R), dtype=torch.float32, device=dev),
                     torch.empty((B, M), dtype=torch.int64, device=dev),
                     torch.zeros((2 * B + 1 + plan.stream_ctl_words(B * N),), dtype=torch.int32, device=dev),
                     torch.empty((B * N, 2), dtype=torch.float32, device=dev),      # LayerNorm moments
                     torch.empty((B, N, net.D), dtype=torch.float32, device=dev),
                     hip.scan_workspace(B, M, I, ca.H, ca.n_token, dev)))            # (candidate sets beyond the LDS; else None)
        # Resident loops: the projector goes through the slides in order and needs longer for a slide than its loop does
        # (65,536 x 2048 rows: 1.2 ms against 0.94), so TWO loop workgroups, each taking its slides one after the other,
        # keep up with any number of slides - and the compute units of the other loops stay the projector's
        # (16 slides: 240 -> 254 units, 52.5 -> [DESIGN 6] M patches/s).
        loops = self.feature_loops(B)
        plan._refresh()
        fused2 = vq.dtype == torch.float32         # (bf16 logits: a launch of their own, plain statistics and publication)
        xf, ef, lf = patches.view(B * N, -1), emb_buf.view(B * N, -1), logits.view(1, B * N, R)
        streamed = (not net.use_pos and fused2 and _env_on("IPSX_CAM_STREAM") and (B == 1 or N % 32 == 0)
                    and plan.stream_supported(B * N, R))
        # (candidate sets beyond the LDS - the shipped M = I = 5000 - run as a TEAM of workgroups per slide, csrc/scan_large_team.h:
        #  each of them keeps a compute unit)
        team = hip.scan_workgroups_per_image(B, M, I, ca.H, ca.n_token)
        if streamed and patches.dtype == torch.float32 and self.native_ok(patches, None):
            free = hip.device_geometry(dev).cus - loops * team
            wgs = int(os.environ.get("IPSX_CAM_WGS", "0")) or free
            # (a team's iteration is ~55 us and the last TWO of them run behind the producer - the rows of the last chunk all
            #  complete in its last round - so a finer end of the stream pays: the last 2 x workgroups units as 32-row tiles,
            #  38,000 rows at M = I = 5000: 39.3 -> 39.6 M patches/s)
            short = int(os.environ.get("IPSX_CAM_SHORT", "0")) or ((-19 if team > 1 else -11) if B == 1 else -1)
            return self.native_call("features", patches, None, logits, mem_idx_buf, zeroed, emb_buf, scan_ws, loops, wgs,
                                    short_first=short)
        with _no_gc_pause():                       # from the loop's launch to its producers': no host stall
            tie, ready, status, ctl = self.persistent_begin(logits, mem_idx_buf, zeroed, B, dev, loops, scan_ws,
                                                            zero_words=2 * B + 1 + plan.stream_ctl_zero_words(B * N))
            if streamed:
                # one workgroup per compute unit the loops leave free (with dynamic pulls one that is placed late just starts
                # late).  Round 4: the loop (3.7 us per iteration) is no longer the bound of a lone slide, the projector is:
                # 64-row tiles at full rate, the last round handed out as 32-row tiles and what is left over then as column
                # quarters, so that the launch ends evenly (short_first = -11: half the workgroups start with a 32-row tile, one round of single units;
                # M patches/s per slide synced / back to back: all tiles 32 rows 39.9 / 41.1, this 41.6 / 43.4; without the
                # quarters the leftover 8-16 units were a round of their own: stream 1.38 -> 1.30 ms)
                free = hip.device_geometry(dev).cus - loops * team
                wgs = int(os.environ.get("IPSX_CAM_WGS", "0")) or free
                short = int(os.environ.get("IPSX_CAM_SHORT", "0")) or (-11 if B == 1 else -1)
                plan.stream(xf, vq, R, ef, logits.view(B * N, R), ctl, ready, workgroups=wgs, slide_rows=N, short_first=short)
                # (whatever two simultaneous finishers leave to each other is published by the last workgroup out: round 5 -
                #  it was one publish_rows launch per slide behind the stream)
            else:
                its = self.feature_parts(loops * team, N, dev, True)
                edges = [0] + [min(N, M + it * I) for it in its[1:]]
                edges[-1] = N
                launches = self.feature_launches(B, N, edges, dev, loops * team)
                if fused2:
                    plan.row_stats(xf[launches[0][0]:launches[0][1]], out=stats[launches[0][0]:launches[0][1]])
                published = None                       # (slide, rows) whose publication rides on the next GEMM launch
                for n_step, (r0, r1, pubs) in enumerate(launches):
                    if not fused2:
                        plan.row_stats(xf[r0:r1], out=stats[r0:r1])
                    emb = plan.encode(xf[r0:r1], stats=stats[r0:r1], out=ef[r0:r1],
                                      publish=(ready[published[0]:published[0] + 1], published[1]) if published else None)
                    published = None
                    emb = emb.view(1, r1 - r0, -1)
                    pos = pos_enc[r0 // N:r0 // N + 1, r0 % N:r0 % N + (r1 - r0)] if net.use_pos else None
                    nxt = launches[n_step + 1] if n_step + 1 < len(launches) else None
                    if fused2 and nxt is not None:
                        hip.logits_stats(emb, pos, vq, R, lf[:, r0:r1], xf[nxt[0]:nxt[1]], stats[nxt[0]:nxt[1]], plan.ln_eps)
                        for b_, rows in pubs[:-1]:     # (a launch across a slide's end: the finished slide is published at once)
                            hip.publish_rows(ready[b_:b_ + 1], rows)
                        published = pubs[-1]
                    else:
                        hip.logits(emb, pos, vq, R, out=lf[:, r0:r1])
                        for b_, rows in pubs:          # after the kernels that wrote those rows
                            hip.publish_rows(ready[b_:b_ + 1], rows)
        net._emb_parts = [emb_buf]
        return self.persistent_end(logits, mem_idx_buf, tie, status, n_iter, dev, scan_ws)

    # ------------------------------------------------------------------ pipeline: one image, persistent trunk stream
    def image_stream(self, patches, pos_enc):
        """One image (the reference's eager-sequential mode, B_seq = 1): the loop is resident on a compute unit of its own;
        the trunk's workgroups - one per remaining unit - pull four, later two patches at a time, encode them, compute their
        logits and publish them (ipsx_trunk_stream).  No parts, no launch between trunk and logits, no loop workgroup looking
        for a free unit beside a trunk launch; what is exposed of the loop is the iteration after the last patch."""
        net, plan = self.net, self.plan()
        N = patches.shape[1]
        M, I, dev = net.M, net.I, patches.device
        ca = net.transf.crs_attn
        vq, R = ca.folded_query(), ca.H * ca.n_token
        self.streams(dev)
        logits, mem_idx_buf, emb_buf, zeroed = self.buffers(
            "image", (N, M, I, R, net.D, str(dev)),
            lambda: (torch.empty((1, N, R), dtype=torch.float32, device=dev),
                     torch.empty((1, M), dtype=torch.int64, device=dev),
                     torch.empty((1, N, net.D), dtype=torch.float32, device=dev),
                     torch.zeros((3 + plan.image_stream_ctl_words(N),), dtype=torch.int32, device=dev)))
        if patches.dtype == torch.float32 and self.native_ok(patches, pos_enc if net.use_pos else None):
            tp = (pos_enc[0] if pos_enc[0].is_contiguous() else pos_enc[0].contiguous()) if net.use_pos else None
            return self.native_call("image", patches, pos_enc if net.use_pos else None, logits, mem_idx_buf, zeroed, emb_buf, None, 0, 0,
                                    trunk_pos=tp, quad_pulls=-1)
        with _no_gc_pause():                       # from the loop's launch to its producer's: no host stall
            tie, ready, status, ctl = self.persistent_begin(logits, mem_idx_buf, zeroed, 1, dev)
            plan.image_stream(patches[0], pos_enc[0] if net.use_pos else None, vq, R, emb_buf[0], logits[0], ctl, ready)
        # (whatever two simultaneous finishers leave to each other is published by the last workgroup out: no launch for it)
        net._emb_parts = [emb_buf]
        return self.persistent_end(logits, mem_idx_buf, tie, status, self.n_iter(N), dev)

    # ------------------------------------------------------------------ pipeline: parts, the loop in ranges beside them
    def parts_with_ranges(self, patches, pos_enc):
        """The patch axis in parts cut at chunk boundaries; part k is encoded (the fused trunk through an index list -
        nothing is copied - every other encoder through a slice) and scored, and the loop iterations that part makes
        possible run on the side stream (ipsx_scan_range resumes from the memory indices) while the encoder is already
        working on part k + 1.  Only the last part's iterations are exposed - what keeps the loop off the critical path when
        the image (and with it the iteration count) grows across GPUs.  Image encoders: parts shrinking towards the end;
        a small batch: small_batch_split; feature inputs without persistent loops: equal parts."""
        net, plan = self.net, self.plan()
        B, N = patches.shape[:2]
        M, I, dev = net.M, net.I, patches.device
        ca = net.transf.crs_attn
        vq, R = ca.folded_query(), ca.H * ca.n_token
        n_iter = self.n_iter(N)
        from .dist import part_iterations
        indexed = net.is_image and patches.is_contiguous() and plan.fused(patches.shape)
        edges = None
        if net.is_image and B * N < self.small_batch_limit(dev) and n_iter < 100:
            edges, its = self.small_batch_split(B, N, dev)
        elif net.is_image and indexed and hip.precision() == "bf16":
            # the bf16 trunk's workgroups take EIGHT patches (fused_trunk_bf16v3.h): a round is 16 patches per unit, and the
            # fixed 50 / 30 / 15 / 5 % cut of a 40,000-patch batch is 12 rounds of work for 9.8 - every part a whole number of
            # rounds instead (dist.plan_iterations: the sharded path's planner, one rank)
            from .dist import launch_model, plan_iterations
            its = plan_iterations(N, M, I, 1, B, launch_model(net, tuple(patches.shape[2:]), hip.device_geometry(dev).cus),
                                  self.OVERLAP_PARTS)
        elif net.is_image:
            its = part_iterations(n_iter, self.OVERLAP_PARTS)
        else:
            its = self.feature_parts(B, N, dev, False)
        P = len(its) - 1
        if edges is None:                                      # parts cut at the loop's chunk boundaries
            edges = [0] + [min(N, M + it * I) for it in its[1:]]
            edges[-1] = N
        key = (B, N, tuple(edges), str(dev))
        if indexed and (self._part_index is None or self._part_index[0] != key):      # int32 patch indices of every part, cached
            rows = torch.arange(B, device=dev, dtype=torch.int32).unsqueeze(1) * N
            self._part_index = (key, [(rows + torch.arange(edges[k], edges[k + 1], device=dev, dtype=torch.int32)).reshape(-1)
                                      for k in range(P)])
        side, main = self.streams(dev)
        flat = patches.reshape(B * N, *patches.shape[2:]) if indexed else None
        logits, mem_idx_buf, tie, scan_ws = self.buffers(
            "parts", (B, N, M, I, R, str(dev)),
            lambda: (torch.empty((B, N, R), dtype=torch.float32, device=dev),
                     torch.empty((B, M), dtype=torch.int64, device=dev),
                     torch.zeros((B,), dtype=torch.int32, device=dev),
                     hip.scan_workspace(B, M, I, ca.H, ca.n_token, dev)))   # None unless M + I exceeds the LDS
        tie.zero_()
        net._emb_parts = parts = []
        side.wait_stream(main)
        for k in range(P):
            lo, hi = edges[k], edges[k + 1]
            if indexed:
                emb = plan.encode_indexed(flat, self._part_index[1][k]).view(B, hi - lo, -1)
            else:
                emb = net._embed(patches[:, lo:hi].reshape(-1, *patches.shape[2:])).view(B, hi - lo, -1)
            parts.append(emb)
            pos = pos_enc[:, lo:hi] if net.use_pos else None
            if k == P - 1:
                # the last part has nothing to run beside: its logits and iterations stay on the main stream (one
                # cross-stream hand-over less on the critical path; it only has to follow the side stream's earlier parts)
                main.wait_stream(side)
                hip.logits(emb, pos, vq, R, out=logits[:, lo:hi])
                hip.scan_range(logits, M, I, ca.H, ca.n_token, its[k], its[k + 1], mem_idx_buf, tie, scan_ws)
                continue
            done = torch.cuda.Event()
            done.record(main)
            emb.record_stream(side)
            with torch.cuda.stream(side):
                side.wait_event(done)
                hip.logits(emb, pos, vq, R, out=logits[:, lo:hi])
                hip.scan_range(logits, M, I, ca.H, ca.n_token, its[k], its[k + 1], mem_idx_buf, tie, scan_ws)
        main.wait_stream(side)
        hip.scan.last_tie = tie
        return mem_idx_buf.clone()                 # the buffer is overwritten by the next call

    # ------------------------------------------------------------------ pipeline: everything else (one piece, or lazy slabs)
    def slabs(self, patches, pos_enc):
        """Encode -> logits -> one scan; patches on the host (lazy loading, reference ips_net.py:204-206,223) arrive in a few
        large slabs on a copy stream while the previous slab is being encoded, and the loop runs on the side stream over the
        iterations whose rows have arrived - only the last slab's iterations are exposed."""
        net = self.net
        B, N = patches.shape[:2]
        M, I, dev = net.M, net.I, net.device
        ca = net.transf.crs_attn
        vq, R = ca.folded_query(), ca.H * ca.n_token
        logits = torch.empty((B, N, R), dtype=torch.float32, device=dev)
        if patches.is_cuda:
            spans, fetch, prefetch = [(0, N)], lambda k: patches, lambda k: None
        else:
            spans, fetch, prefetch = self.lazy_slabs(patches)
        n_iter = self.n_iter(N)
        beside = len(spans) > 1 and _env_on("IPSX_OVERLAP_SCAN") and not hip.dedup_blank()
        if beside:
            side, main = self.streams(dev)
            mem_idx = torch.empty((B, M), dtype=torch.int64, device=dev)
            tie = torch.zeros((B,), dtype=torch.int32, device=dev)
            scan_ws = hip.scan_workspace(B, M, I, ca.H, ca.n_token, dev)     # None unless M + I exceeds the LDS
            for t in (logits, mem_idx, tie) + ((scan_ws,) if scan_ws is not None else ()):
                t.record_stream(side)
            side.wait_stream(main)
            it_prev = 0
        parts = []
        for k, (lo, hi) in enumerate(spans):
            part = fetch(k)
            emb = net._embed(part.reshape(-1, *patches.shape[2:])).view(B, hi - lo, -1)
            hip.logits(emb, pos_enc[:, lo:hi] if net.use_pos else None, vq, R, out=logits[:, lo:hi])
            prefetch(k + 1)          # after the encoder is enqueued: a pageable-memory copy blocks the host, not the GPU
            parts.append(emb)
            if beside:
                it_k = n_iter if hi >= N else max(it_prev, (hi - M) // I)
                if it_k > it_prev:
                    done = torch.cuda.Event()
                    done.record(main)
                    with torch.cuda.stream(side):
                        side.wait_event(done)
                        hip.scan_range(logits, M, I, ca.H, ca.n_token, it_prev, it_k, mem_idx, tie, scan_ws)
                    it_prev = it_k
        net._emb_parts = parts
        if beside:
            main.wait_stream(side)
            hip.scan.last_tie = tie
            return mem_idx
        return hip.scan(logits, M, I, ca.H, ca.n_token)

    def lazy_slabs(self, patches):
        """Host tensor -> (spans, fetch, prefetch): slab k travels on the copy stream while slab k - 1 is encoded.  The first
        slab is the only copy nothing hides, so the slabs GROW (1/6, 1/2, then full slabs of IPSX_LAZY_SLAB_MB: the copy
        engine moves patches ~3x faster than the fp32 encoder consumes them).  With 288 GB of HBM the slabs are kept (up to
        IPSX_LAZY_KEEP_MB, default 16 GiB) so that the M winners are gathered on the device; beyond that a two-slot ring is
        recycled and the final gather happens on the host exactly as in the reference."""
        net = self.net
        B, N = patches.shape[:2]
        row_bytes = patches[0, 0].numel() * patches.element_size()
        slab_bytes = int(os.environ.get("IPSX_LAZY_SLAB_MB", "0")) << 20 or self.LAZY_SLAB_BYTES
        per = max(1, min(N, slab_bytes // max(1, B * row_bytes)))
        spans, lo = [], 0
        for size in (max(1, per // 6), max(1, per // 2)):
            if N - lo > per:
                spans.append((lo, lo + size))
                lo += size
        spans += [(a, min(a + per, N)) for a in range(lo, N, per)]
        keep = patches.numel() * patches.element_size() <= int(os.environ.get("IPSX_LAZY_KEEP_MB", "16384")) << 20
        dev = net.device
        bufs = self.buffers("lazy", (keep, tuple(patches.shape), per, patches.dtype, str(dev)),
                            lambda: [torch.empty(patches.shape, dtype=patches.dtype, device=dev)] if keep else
                            [torch.empty((B, per) + tuple(patches.shape[2:]), dtype=patches.dtype, device=dev) for _ in range(2)])
        if keep:
            store = bufs[0]
            net._device_patches = store
            dst = lambda k, lo, hi: store[:, lo:hi]
        else:
            dst = lambda k, lo, hi: bufs[k % 2][:, :hi - lo]
        if self._copy is None or self._copy.device != torch.device(dev):
            self._copy = torch.cuda.Stream(device=dev)
        copy_stream, main = self._copy, torch.cuda.current_stream(dev)
        # the buffers come from the main stream's allocator pool: a block freed in Python a moment ago may still be read by
        # kernels queued on the main stream (the previous step's backward / optimizer), so the copy stream must not write
        # it before the main stream got that far - and the allocator must know the copy stream uses it
        copy_stream.wait_stream(main)
        for buf in bufs:
            buf.record_stream(copy_stream)
        ready, freed = {}, {}

        def issue(k):
            lo, hi = spans[k]
            with torch.cuda.stream(copy_stream):
                if not keep and k - 2 in freed:
                    copy_stream.wait_event(freed[k - 2])          # ring slot must have been consumed
                d = dst(k, lo, hi)
                for b in range(B):                                  # per image: contiguous on both sides
                    d[b].copy_(patches[b, lo:hi], non_blocking=True)
                ready[k] = torch.cuda.Event()
                ready[k].record(copy_stream)

        issue(0)

        def fetch(k):
            main.wait_event(ready[k])
            part = dst(k, *spans[k])
            if not keep:
                part = part.clone()
                freed[k] = torch.cuda.Event()
                freed[k].record(main)
            return part

        def prefetch(k):
            if k < len(spans):
                issue(k)                                            # travels while slab k-1 is being encoded

        return spans, fetch, prefetch
