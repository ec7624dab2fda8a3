"""IPSNet: patch encoder + Iterative Patch Selection + aggregation + task heads.

Drop-in mirror of /root/reference/architecture/ips_net.py:11-283: same class
name, ``IPSNet(device, conf)`` constructor, ``ips(patches) -> (mem_patch,
mem_pos)`` and ``forward(mem_patch, mem_pos=None) -> {task: probs}`` signatures,
same sub-module / parameter names (``encoder.N...``, ``transf...``,
``output_layers.<task>.0``), so the reference's ``main.py`` and
``training/iterative.py`` drive it unchanged and state-dicts interchange.

What is different is how ``ips()`` executes on a ROCm device.  The reference
runs a Python loop of ~75 stock kernels per chunk.  In eval / no-grad mode the
encoder is a pure per-patch function (BatchNorm uses running statistics,
reference :191-193) and a patch's attention logits do not depend on which other
patches are in the candidate set (only the softmax denominator does), so the
HIP path is three launches' worth of work:

  1. ``encode``   every patch once (fused ResNet trunk / projector kernels),
  2. ``logits``   K-projection and q.k per patch, once,
  3. ``scan``     one persistent workgroup per image replays the reference's
                  chunk loop on the cached logits: softmax over memory+chunk,
                  mean over heads and tokens, top-M, repeat,

followed by row gathers of the M winners.  Results are those of the reference
loop: same candidate order (memory first, reference :231-232), same arithmetic
order per patch.  ``last_mem_idx`` exposes the selected indices (the reference
only returns the gathered patches).
"""

import math
import os

import torch
from torch import nn

from .. import hip
from ..shuffle import shuffle_batch, shuffle_instance
from .resnet import load_torchvision_checkpoint, resnet18_trunk, resnet50_trunk
from .transformer import Transformer, pos_enc_1d


class IPSNet(nn.Module):
    """Patch encoder, IPS, cross-attention aggregator and classification heads."""

    # ---------------------------------------------------------------- construction
    def get_conv_patch_enc(self, enc_type, pretrained, n_chan_in, n_res_blocks):
        """ResNet stem + 2 or 4 residual stages + global average pool (reference :17-52)."""
        if enc_type == 'resnet18':
            trunk = resnet18_trunk()
        elif enc_type == 'resnet50':
            trunk = resnet50_trunk()
        else:
            raise ValueError("unknown enc_type {!r}".format(enc_type))
        if pretrained:
            # the reference lets torchvision download IMAGENET1K_V1 (:19-27); this framework targets machines
            # without network access, so the same file is read from a local path instead
            path = os.environ.get("IPSX_PRETRAINED_" + enc_type.upper()) or os.environ.get("IPSX_PRETRAINED")
            if not path:
                raise RuntimeError(
                    "pretrained=True: point IPSX_PRETRAINED_{} (or IPSX_PRETRAINED) at a local torchvision {} "
                    "state-dict file (IMAGENET1K_V1 .pth); this image cannot download it".format(enc_type.upper(), enc_type))
            load_torchvision_checkpoint(trunk, path)
        if n_chan_in == 1:
            # the reference swaps the 3-channel stem for a fresh 1-channel one (:29-31)
            trunk.conv1 = nn.Conv2d(n_chan_in, 64, kernel_size=7, stride=2, padding=3, bias=False)
        stages = [trunk.conv1, trunk.bn1, trunk.relu, trunk.maxpool, trunk.layer1, trunk.layer2]
        if n_res_blocks == 4:
            stages += [trunk.layer3, trunk.layer4]
        stages.append(trunk.avgpool)
        return nn.Sequential(*stages)

    def get_projector(self, n_chan_in, D):
        """LN(no affine) -> Linear -> BatchNorm1d -> ReLU for pre-extracted features (:54-60)."""
        return nn.Sequential(
            nn.LayerNorm(n_chan_in, eps=1e-05, elementwise_affine=False),
            nn.Linear(n_chan_in, D),
            nn.BatchNorm1d(D),
            nn.ReLU(),
        )

    def get_output_layers(self, tasks):
        """One ``Linear(D, n_class) -> softmax|sigmoid`` head per task (:62-83)."""
        heads = nn.ModuleDict()
        for task in tasks.values():
            act = {'softmax': lambda: nn.Softmax(dim=-1), 'sigmoid': nn.Sigmoid}[task['act_fn']]()
            heads[task['name']] = nn.Sequential(nn.Linear(self.D, self.n_class), act)
        return heads

    def __init__(self, device, conf):
        super().__init__()
        hip.install_optimizer_hook()        # every optimizer step invalidates the packed weights (see hip.py)
        self.device = device
        self.n_class = conf.n_class
        self.M, self.I, self.D = conf.M, conf.I, conf.D
        self.use_pos = conf.use_pos
        self.tasks = conf.tasks
        self.shuffle = conf.shuffle
        self.shuffle_style = conf.shuffle_style
        self.is_image = conf.is_image

        if self.is_image:
            self.encoder = self.get_conv_patch_enc(conf.enc_type, conf.pretrained,
                                                   conf.n_chan_in, conf.n_res_blocks)
        else:
            self.encoder = self.get_projector(conf.n_chan_in, self.D)

        self.transf = Transformer(conf.n_token, conf.H, conf.D, conf.D_k, conf.D_v,
                                  conf.D_inner, conf.attn_dropout, conf.dropout)

        # plain tensor attribute, not a buffer - exactly as the reference (:110-113)
        self.pos_enc = pos_enc_1d(conf.D, conf.N).unsqueeze(0).to(device) if conf.use_pos else None

        self.output_layers = self.get_output_layers(conf.tasks)

        # additions that do not change the drop-in surface
        self.last_mem_idx = None      # (B, M) int64 indices chosen by the last ips() call
        self._plan = None             # packed-weight cache of the HIP encoder
        self._emb_parts = None        # eval-mode embeddings of the last ips() call (see last_mem_emb)
        self._mem_emb = None
        if self.is_image and hip.on_device(device):
            # the training step's convolutions run channels-last (training/fused_encoder.py): weights stored that way
            # from the start (before any optimizer state exists) are not re-laid-out in every step.  Shapes, names and
            # values are untouched - state dicts interchange with the reference as before.
            from ..training import fused_encoder
            if fused_encoder.enabled() and fused_encoder.supported(self.encoder):
                self.encoder.to(memory_format=torch.channels_last)

    # ---------------------------------------------------------------- small pieces
    def do_shuffle(self, patches, pos_enc):
        """Permute the patch axis (and pos_enc identically) to randomise ties (:118-134)."""
        if self.shuffle_style == 'batch':
            patches, perm = shuffle_batch(patches)
            if torch.is_tensor(pos_enc):
                pos_enc, _ = shuffle_batch(pos_enc, perm)
        elif self.shuffle_style == 'instance':
            patches, perm = shuffle_instance(patches, 1)
            if torch.is_tensor(pos_enc):
                pos_enc, _ = shuffle_instance(pos_enc, 1, perm)
        return patches, pos_enc

    def score_and_select(self, emb, emb_pos, M, idx):
        """Score ``L`` candidates, keep the top ``M`` (:136-155).

        Scores come from ``emb_pos`` when given, the gathered memory from ``emb``.
        """
        scored = emb_pos if torch.is_tensor(emb_pos) else emb
        if hip.on_device(scored):
            top = hip.topm(self.transf.get_scores(scored), M)
        else:
            top = torch.topk(self.transf.get_scores(scored), M, dim=-1)[1]
        mem_emb = torch.gather(emb, 1, top.unsqueeze(-1).expand(-1, -1, emb.shape[2]))
        return mem_emb, torch.gather(idx, 1, top)

    def get_preds(self, embeddings):
        """Task ``t`` reads aggregated token ``t_id`` (:157-166)."""
        fused = hip.on_device(embeddings) and not (
            torch.is_grad_enabled() and (embeddings.requires_grad or
                                         any(p.requires_grad for p in self.output_layers.parameters())))
        preds = {}
        for task in self.tasks.values():
            layer = self.output_layers[task['name']]
            if fused:   # Linear + softmax|sigmoid in one kernel
                preds[task['name']] = hip.head(embeddings, task['id'], layer[0], task['act_fn'])
            else:
                preds[task['name']] = layer(embeddings[:, task['id']])
        return preds

    def _embed(self, x):
        """(P, C, h, w) | (P, F)  ->  (P, D) with the encoder's CURRENT mode."""
        if hip.on_device(x) and not self.encoder.training and not (
                torch.is_grad_enabled() and any(p.requires_grad for p in self.encoder.parameters())):
            if self._plan is None:
                self._plan = hip.EncoderPlan(self.encoder, self.is_image)
            return self._plan.encode(x)
        if hip.on_device(x) and self.encoder.training and self.is_image and torch.is_grad_enabled():
            # training step (reference training/iterative.py:158-163): same modules, BatchNorm + add + ReLU fused
            from ..training import fused_encoder
            if getattr(self, "_fused_train_ok", None) is None:        # (the module tree does not change after construction)
                self._fused_train_ok = fused_encoder.supported(self.encoder)
            if self._fused_train_ok and fused_encoder.enabled():
                return fused_encoder.encode(self.encoder, x)
        return self.encoder(x).flatten(1)

    # ---------------------------------------------------------------- IPS
    @torch.no_grad()
    def ips(self, patches):
        """Iterative Patch Selection (reference :169-262).

        ``patches``: (B, N, C, h, w) images or (B, N, F) features, on the device
        (eager loading) or on the host (lazy loading).  Returns the M selected
        patches ``(B, M, ...)`` and their positional encodings ``(B, M, D)`` (or
        ``None``), both on ``self.device``, ordered by score of the last round.
        """
        M, device, pos_enc = self.M, self.device, self.pos_enc
        B, N = patches.shape[:2]

        self._emb_parts = self._mem_emb = None
        if M >= N:  # nothing to select (:185-188)
            self.last_mem_idx = None
            return patches.to(device), (pos_enc.expand(B, -1, -1) if self.use_pos else None)

        was_training = self.training
        if was_training:  # IPS always scores with running BN statistics and no dropout
            self.encoder.eval()
            self.transf.eval()
        try:
            if self.use_pos:
                pos_enc = pos_enc.expand(B, -1, -1)
            if self.shuffle:
                patches, pos_enc = self.do_shuffle(patches, pos_enc)

            if hip.on_device(device):
                if self._plan is None:
                    self._plan = hip.EncoderPlan(self.encoder, self.is_image)
                with self._plan.hold():            # weights cannot change inside a no-grad call: check them once
                    mem_idx = self._select_hip(patches, pos_enc)
            else:
                mem_idx = self._select_aten(patches, pos_enc)

            src = self._device_patches if getattr(self, "_device_patches", None) is not None else patches
            mem_patch = self._take(src, mem_idx).to(device)
            self._device_patches = None
            mem_pos = self._take(pos_enc, mem_idx) if self.use_pos else None
        finally:
            if was_training:
                self.encoder.train()
                self.transf.train()

        self.last_mem_idx = mem_idx
        return mem_patch, mem_pos

    def _chunks(self, N):
        """[0, M) then ceil((N-M)/I) chunks of I (last one ragged) - reference :206,217-221."""
        yield 0, self.M
        for start in range(self.M, N, self.I):
            yield start, min(start + self.I, N)

    @staticmethod
    def _take(src, idx):
        """``src[b, idx[b, m]]`` on the device ``src`` lives on."""
        if hip.on_device(src):
            return hip.gather_rows(src, idx)
        idx = idx.to(src.device)
        view = idx.view(*idx.shape, *(1,) * (src.dim() - 2)).expand(-1, -1, *src.shape[2:])
        return torch.gather(src.expand(idx.shape[0], *src.shape[1:]), 1, view)

    def _select_hip(self, patches, pos_enc):
        """encode-all -> logits -> one scan launch.  Patches may still be on the host (lazy loading)."""
        B, N = patches.shape[:2]
        ca = self.transf.crs_attn
        if patches.is_cuda and self._can_stream_image(patches):
            return self._select_image_stream(patches, pos_enc)
        if patches.is_cuda and self._can_overlap(patches):
            return self._select_hip_overlapped(patches, pos_enc)
        vq, R = ca.folded_query(), ca.H * ca.n_token
        logits = torch.empty((B, N, ca.H * ca.n_token), dtype=torch.float32, device=self.device)
        self._device_patches = None
        if patches.is_cuda:
            spans, fetch, prefetch = [(0, N)], lambda k: patches, lambda k: None
        else:
            spans, fetch, prefetch = self._lazy_slabs(patches)
        parts = []
        n_iter = math.ceil((N - self.M) / self.I)
        # several slabs (lazy loading): the selection loop runs on a side stream over the iterations whose rows have
        # arrived while the next slab is being encoded, as in _select_hip_overlapped - only the last slab's iterations
        # are exposed
        beside = len(spans) > 1 and os.environ.get("IPSX_OVERLAP_SCAN", "1") != "0" and not hip.dedup_blank()
        if beside:
            dev = self.device
            if getattr(self, "_side_stream", None) is None or self._side_stream.device != torch.device(dev):
                self._side_stream = torch.cuda.Stream(device=dev, priority=-1)
            side, main = self._side_stream, torch.cuda.current_stream(dev)
            mem_idx = torch.empty((B, self.M), dtype=torch.int64, device=dev)
            tie = torch.zeros((B,), dtype=torch.int32, device=dev)
            scan_ws = hip.scan_workspace(B, self.M, self.I, ca.H, ca.n_token, dev)     # None unless M + I exceeds the LDS
            for t in (logits, mem_idx, tie) + ((scan_ws,) if scan_ws is not None else ()):
                t.record_stream(side)
            side.wait_stream(main)
            it_prev = 0
        for k, (lo, hi) in enumerate(spans):
            part = fetch(k)
            emb = self._embed(part.reshape(-1, *patches.shape[2:])).view(B, hi - lo, -1)
            pos = pos_enc[:, lo:hi] if self.use_pos else None
            hip.logits(emb, pos, vq, R, out=logits[:, lo:hi])
            prefetch(k + 1)          # after the encoder is enqueued: a pageable-memory copy blocks the host, not the GPU
            parts.append(emb)
            if beside:
                it_k = n_iter if hi >= N else max(it_prev, (hi - self.M) // self.I)
                if it_k > it_prev:
                    done = torch.cuda.Event()
                    done.record(main)
                    with torch.cuda.stream(side):
                        side.wait_event(done)
                        hip.scan_range(logits, self.M, self.I, ca.H, ca.n_token, it_prev, it_k, mem_idx, tie, scan_ws)
                    it_prev = it_k
        self._emb_parts = parts
        if beside:
            main.wait_stream(side)
            hip.scan.last_tie = tie
            return mem_idx
        return hip.scan(logits, self.M, self.I, ca.H, ca.n_token)

    # The selection loop is sequential over chunks but only ever needs the logits of the chunks it has
    # reached, so it runs on a side stream over the part of the image that is already encoded while the
    # encoder works on the next part (ipsx_scan_range resumes from the memory indices).  Only the last part's
    # iterations are exposed - this is what keeps the scan off the critical path when the image (and with it
    # the iteration count) grows across GPUs.  Parts are cut at chunk boundaries; the fused trunk encodes a part
    # through an index list (nothing is copied), every other encoder through a slice of the patch axis.  With
    # feature inputs (projector) the loop is the long pole instead, and the parts GROW so that it starts early
    # (dist.PART_SHARES_LOOP_BOUND).  IPSX_OVERLAP_SCAN=0 switches it off.
    _OVERLAP_PARTS = 4

    def _can_overlap(self, patches):
        if os.environ.get("IPSX_OVERLAP_SCAN", "1") == "0" or hip.dedup_blank():
            return False
        n_iter = math.ceil((patches.shape[1] - self.M) / self.I)
        if self.is_image and patches.shape[0] * patches.shape[1] < 32768 and n_iter < 100:
            # a small batch does not fill the GPU four times over: it is encoded in the whole workgroup rounds it fills
            # (2048 patches each) plus the remainder, and the loop over the first part runs beside the remainder's
            # encoding, where most compute units are idle anyway (_small_batch_split); below one round there is
            # nothing to run beside
            # (the split is about the FUSED trunk's rounds: a layer-by-layer trunk - other patch sizes - cut in two just runs
            #  every layer twice at half the occupancy: traffic signs 25.3 ms against 22.7 ms in one piece)
            if self._plan is None:
                self._plan = hip.EncoderPlan(self.encoder, self.is_image)
            return (not self.encoder.training) and self._plan.fused(patches.shape) and \
                self._small_batch_split(patches.shape[0], patches.shape[1]) is not None
        # (feature inputs: the loop is the long pole whatever its length - a slide at the reference's shipped M = I = 5000 has
        #  7 iterations of 10,000 candidates - so any loop of a few iterations runs beside the projector's later parts)
        return (not self.encoder.training) and n_iter >= (2 * self._OVERLAP_PARTS if self.is_image else 3)

    def _small_batch_split(self, B, N):
        """How to cut a small image batch (one to two rounds of the fused trunk): (edges, its) - part k encodes rows
        edges[k]..edges[k+1] of every image and the loop then runs iterations its[k]..its[k+1], those whose rows are
        encoded by then - or None when that leaves nothing on either side.  The parts are the ENCODER's units, not the
        loop's: half a round first (1,024 patches: one wavefront per SIMD), then 992 - 248 workgroups, a compute unit of
        every XCD stays free for the loop of the part before, which runs beside it - then the rest.  One image of the
        headline workload: 1,024 + 992 + 484 rows, iterations 15 + 15 + 9; the 484 go through the
        two-wavefronts-per-patch kernel (csrc/fused_trunk_pair.h).
        Why a free unit: the loop's workgroup on a unit it shares with fp32 MFMA wavefronts takes 24 us per iteration
        instead of 6.7 (tools/scan_beside.py; both want the same fp32 lanes).  Whether it FINDS the free unit is the
        dispatcher's business: workgroups are dealt to XCDs and their shader engines in turn, and only a launch of at most
        224 workgroups (7 per engine) leaves a unit free wherever the next workgroup lands.  Parts of 896 would cost more
        than they save here (measured, one image: 1,024 + 896 + 448 + 132 -> 1.00 ms, this split 0.91, and 1.00 when the
        caller's own event records shift the dispatcher's turn - bench.py --no-kernel-events tells the two apart)."""
        n_iter = math.ceil((N - self.M) / self.I)
        rounds = (B * N) // 2048
        if rounds != 1:        # measured (bench.py --config b1 / --batch 2): +8 % at one round + remainder, -4 % at two
            return None
        edges, its = [0], [0]
        for total in (1024, 1024 + 992):
            e = total // B
            it = (e - self.M) // self.I                        # iterations whose rows lie inside the first e of every image
            if e < N and its[-1] < it < n_iter:
                edges.append(e)
                its.append(it)
        if len(its) == 1:
            return None
        return edges + [N], its + [n_iter]

    def _scan_side_stream(self, dev):
        if getattr(self, "_side_stream", None) is None or self._side_stream.device != dev:
            self._side_stream = torch.cuda.Stream(device=dev, priority=-1)   # its few workgroups must not queue behind the encoder grid
        return self._side_stream, torch.cuda.current_stream(dev)

    def _feature_parts(self, B, N):
        """Iterations at which ONE slide's rows are cut into projector launches (the persistent feature pipeline)."""
        M, I = self.M, self.I
        n_iter = math.ceil((N - M) / I)
        # persistent loops: every slide's loop owns a compute unit, the projector - which goes slide by slide - has the
        # others.  Workgroups go to the 8 XCDs round-robin whatever is free there (timeline of a 252-workgroup launch beside
        # one loop: two rounds), so what a single-round launch can count on is the free units of the FULLEST XCD, eight times
        cus = 8 * (32 - -(-B // 8))
        cap = max(I, (cus * 64) // I * I)                    # most rows of a slide one launch can take, whole chunks
        half = max(I, (cus * 32) // I * I)                   # ... one launch of half-size workgroups (<= 127 row tiles:
        #                                                        conv_nhwc_impl then halves the tile and the launch time)
        its = [0]
        if self.D >= 512 and N > cap + half and os.environ.get("IPSX_CAM_PARTS", "equal") == "latency":
            # (opt-in, measured in round 3 and NOT the default.)  A launch costs one workgroup's time whatever its size,
            # the loop can only take a part once ALL of it is published, and what is left of the loop after the last
            # publication is serial time.  So: a HALF-TILE launch first (the loop starts after half the time), full
            # launches in the middle, and the end of the slide as half-tile launches with the smallest last: the
            # projector's chain shrinks from 1.55 + a 0.29 ms tail to 1.70 + 0.03 ms (kernel timeline,
            # profiles/r03c_cam_timeline_latency.txt) - and the slide takes 1.98 ms instead of 1.95, because with
            # its rows always there the LOOP is the bound: 255 iterations x 5.4 - 6.3 us beside the GEMM.  It pays
            # once the loop is faster.
            rows = [half]
            left = N - half
            while left > cap + half:
                rows.append(cap)
                left -= cap
            if left > cap:                               # a full launch and a small rest
                rows.append(cap)
                left -= cap
            while left > 0:
                take = min(half, left)
                rows.append(take)
                left -= take
            if len(rows) >= 2 and rows[-1] > rows[-2]:   # the smallest part last
                rows[-1], rows[-2] = rows[-2], rows[-1]
            edge = 0
            for rws in rows[:-1]:
                edge += rws
                nxt = max(its[-1] + 1, (edge - M) // I)
                if nxt >= n_iter:
                    break
                its.append(nxt)
        else:
            n_part = min(16, max(1, math.ceil(N / cap)))
            for k in range(1, n_part):                       # equal parts: edge k at about k * N / n_part rows
                nxt = max(its[-1] + 1, round((k * N / n_part - M) / I))
                if nxt >= n_iter:
                    break
                its.append(nxt)
        its.append(n_iter)
        return its

    def _feature_parts_plain(self, B, N):
        """Feature inputs WITHOUT the persistent loop: equal parts, every launch takes its rows of all B slides and is sized
        to fill the 256 compute units once."""
        M, I = self.M, self.I
        n_iter = math.ceil((N - M) / I)
        cap = max(I, (256 * 64 // max(B, 1)) // I * I)
        n_part = min(16, max(1, math.ceil(N / cap)))
        its = [0]
        for k in range(1, n_part):
            nxt = max(its[-1] + 1, round((k * N / n_part - M) / I))
            if nxt >= n_iter:
                break
            its.append(nxt)
        its.append(n_iter)
        return its

    def _feature_launches(self, B, N, P, edges):
        """The projector launches of _select_features_persistent: (first row, end row) in the FLAT (B * N) row space + what
        each makes visible, [(slide, rows)]."""
        I = self.I
        launches = []
        if B == 1 or self.use_pos:
            for b_ in range(B):
                for k in range(P):
                    launches.append((b_ * N + edges[k], b_ * N + edges[k + 1], [(b_, edges[k + 1])]))
        else:
            # rows of a full launch: 224 workgroups (28 per XCD).  Measured at 2 / 16 slides: 208 -> 36.3 / 43.1 M patches/s,
            # 224 -> 36.5 / 45.9, 240 -> 30.4 / 37.8 (now and then a workgroup waits for a second round: the free units of
            # the fullest XCD are a bound, not a promise - round 2 had found the same cliff between 224 and 232)
            cap = max(I, min(224, 8 * (32 - -(-B // 8))) * 64 // I * I)
            r0 = 0
            while r0 < B * N:
                # the first launch of the call stays short: the first loop starts after M + I rows' worth of projector
                r1 = min(B * N, r0 + (cap if r0 > 0 else min(cap, edges[1])))
                pubs = [(b_, min(N, r1 - b_ * N)) for b_ in range(r0 // N, (r1 - 1) // N + 1)]
                launches.append((r0, r1, pubs))
                r0 = r1
        return launches

    def _select_features_persistent(self, patches, pos_enc):
        """Feature inputs, up to IPSX_PERSIST_MAX_B slides (the loop is the long pole and a slide's loop occupies ONE
        compute unit): every slide's loop is launched once, up front, as a persistent kernel that owns its compute unit and
        waits for the rows as the projector publishes them - no re-launch per part, no waiting for a compute unit to drain,
        no projector workgroups competing for the loop's issue slots.  The projector works through the slides ONE AFTER THE
        OTHER (a slide's parts fill the other compute units exactly once each), publishing to the slide's own progress word:
        the loop of slide b runs beside the projector of slide b + 1, so from the second slide on the call runs at the
        projector's rate.  (Until round 3 more than one slide went through a copy of the part and un-fused launches, and was
        SLOWER per patch than one slide: 30 / 25 / 34.5 M patches/s at 2 / 4 / 8 slides against 34.7 M at one.)
        Per part two launches: the GEMM (LayerNorm in its operand load; its first thread publishes what was enqueued before
        it) and the logits of the part together with the row moments of the NEXT part (of this slide or the next one)."""
        B, N = patches.shape[:2]
        M, I, dev = self.M, self.I, patches.device
        ca = self.transf.crs_attn
        vq, R = ca.folded_query(), ca.H * ca.n_token
        n_iter = math.ceil((N - M) / I)
        its = self._feature_parts(B, N)
        P = len(its) - 1
        edges = [0] + [min(N, M + it * I) for it in its[1:]]
        edges[-1] = N
        side, main = self._scan_side_stream(dev)
        # per-call device buffers are kept between calls of the same shape: a buffer that another stream has used cannot be
        # re-used by the allocator until that stream's work is known to be over, and allocating afresh in every call makes
        # the host stall in hipMalloc now and then (embeddings of the whole batch too: five 27 MB parts per call made the
        # caching allocator go back to the driver - tens of milliseconds on the host)
        bkey = ("features", B, N, M, I, R, self.D, str(dev))
        if getattr(self, "_feat_bufs_key", None) != bkey:
            self._feat_bufs = (torch.empty((B, N, R), dtype=torch.float32, device=dev),
                               torch.empty((B, M), dtype=torch.int64, device=dev),
                               # tie flags | progress word per slide | status | control words of the projector stream: ONE fill per call
                               torch.zeros((2 * B + 1 + self._plan.stream_ctl_words(B * N),), dtype=torch.int32, device=dev),
                               torch.empty((B * N, 2), dtype=torch.float32, device=dev),      # LayerNorm moments
                               torch.empty((B, N, self.D), dtype=torch.float32, device=dev))
            self._feat_bufs_key = bkey
            for t in self._feat_bufs:
                t.record_stream(side)
        logits, mem_idx_buf, zeroed, stats, emb_buf = self._feat_bufs
        tie, words, ctl = zeroed[:B], zeroed[B:2 * B + 1], zeroed[2 * B + 1:]
        zeroed.zero_()
        # A loop that gave up waiting (bounded at ~5 s: e.g. something serialises the kernels, so that its producers
        # cannot run beside it) is REDONE in the same call by the conditional launch behind it (scan_range_if below:
        # every workgroup leaves at once unless the status word says "timed out"), so this call's results are valid
        # either way and no host synchronisation is added.  The status word is also mirrored into pinned host memory,
        # asynchronously, and looked at in the NEXT call - by then it has long arrived - to say so once.
        mirror = getattr(self, "_scan_status_host", None)
        if mirror is not None and int(mirror.item()) & 1 and not getattr(self, "_scan_timeout_warned", False):
            import warnings
            warnings.warn("the persistent selection loop of an earlier ips() call timed out waiting for rows and was "
                          "redone with per-call launches (results valid; IPSX_SCAN_PERSIST=0 avoids the wait)")
            self._scan_timeout_warned = True
        ready, status = words[:B], words[B:B + 1]
        self._scan_status = status
        side.wait_stream(main)                     # the buffers above are the main stream's; previous readers are done
        with torch.cuda.stream(side):
            hip.scan_persistent(logits, M, I, ca.H, ca.n_token, mem_idx_buf, tie, ready, status)
        # the projector must not take the compute units before a loop has its own.  (Also true of the persistent stream,
        # whose workgroups sit one to a unit and leave a unit per loop free: a workgroup is dealt to an XCD before it looks
        # for a unit there, so a loop that arrives second may be dealt to a FULL XCD and start when the stream ends - measured
        # without the gate: 31 M patches/s with calls back to back against 43 M.)
        hip.scan_gate(status)
        self._plan._refresh()
        fused2 = vq.dtype == torch.float32         # (bf16 logits: a launch of their own, plain statistics and publication)
        # The launches: (first row, end row) in the FLAT (B * N) row space + what each makes visible, [(slide, rows)].  One
        # slide, or positional encodings (a table per slide position): a slide's parts.  Several slides without them: the
        # slides are one stream of rows cut into full launches wherever a slide ends (the patch tensor is contiguous, a
        # launch may take the end of one slide and the start of the next) - 4.13 launches per 65,536-row slide instead of 5.
        # No positional encoding, fp32 logits: the projector as ONE persistent launch (ipsx_projector_stream) whose
        # workgroups pull 64-row tiles off the flat stream of the slides' rows, do moments + Linear + logits per tile and
        # advance the slides' progress words as tiles complete - a slide's loop starts after half a tile time and is never
        # a whole part behind, and with several slides the projector runs without a seam from the first row to the last.
        stream = (not self.use_pos and fused2 and os.environ.get("IPSX_CAM_STREAM", "1") != "0"
                  and (B == 1 or N % 32 == 0) and self._plan.stream_supported(B * N, R))
        if stream:
            # one workgroup per compute unit the loops leave free (with dynamic pulls one that is placed late just starts
            # late).  Round 4 (the loop at 3.7 us per iteration is no longer the bound - the projector is): 64-row tiles at full
            # rate, the first rows of a lone slide from half the workgroups' short first tiles, and the last two rounds
            # handed out as 32-row tiles so that the launch ends evenly (short_first = -20: measured, M patches/s per
            # slide, synced: all tiles 32 rows on 248 / 255 units 39.3 / 39.9, this on 248 / 255 units 41.1 / 41.5)
            free = torch.cuda.get_device_properties(dev).multi_processor_count - B
            wgs = int(os.environ.get("IPSX_CAM_WGS", "0")) or free
            short = int(os.environ.get("IPSX_CAM_SHORT", "0")) or (-20 if B == 1 else -1)
            self._plan.stream(patches.view(B * N, -1), vq, R, emb_buf.view(B * N, -1), logits.view(B * N, R), ctl, ready,
                              workgroups=wgs, slide_rows=N, short_first=short)
            for b_ in range(B):                    # (whatever the last finishers left to each other; the launch is over)
                hip.publish_rows(ready[b_:b_ + 1], N)
            launches = []
        else:
            launches = self._feature_launches(B, N, P, edges)
        xf, ef, lf = patches.view(B * N, -1), emb_buf.view(B * N, -1), logits.view(1, B * N, R)
        if fused2 and launches:
            self._plan.row_stats(xf[launches[0][0]:launches[0][1]], out=stats[launches[0][0]:launches[0][1]])
        published = None                           # (slide, rows) whose publication rides on the next GEMM launch
        for n_step, (r0, r1, pubs) in enumerate(launches):
            if not fused2:
                self._plan.row_stats(xf[r0:r1], out=stats[r0:r1])
            emb = self._plan.encode(xf[r0:r1], stats=stats[r0:r1], out=ef[r0:r1],
                                    publish=(ready[published[0]:published[0] + 1], published[1]) if published else None)
            published = None
            emb = emb.view(1, r1 - r0, -1)
            pos = pos_enc[r0 // N:r0 // N + 1, r0 % N:r0 % N + (r1 - r0)] if self.use_pos else None
            nxt = launches[n_step + 1] if n_step + 1 < len(launches) else None
            if fused2 and nxt is not None:
                hip.logits_stats(emb, pos, vq, R, lf[:, r0:r1], xf[nxt[0]:nxt[1]], stats[nxt[0]:nxt[1]], self._plan.ln_eps)
                for b_, rows in pubs[:-1]:         # (a launch across a slide's end: the finished slide is published at once)
                    hip.publish_rows(ready[b_:b_ + 1], rows)
                published = pubs[-1]
            else:
                hip.logits(emb, pos, vq, R, out=lf[:, r0:r1])
                for b_, rows in pubs:              # after the kernels that wrote those rows
                    hip.publish_rows(ready[b_:b_ + 1], rows)
        self._emb_parts = [emb_buf]
        main.wait_stream(side)
        hip.scan_range_if(logits, M, I, ca.H, ca.n_token, 0, n_iter, mem_idx_buf, tie, status, 1)   # no-op unless timed out
        if getattr(self, "_scan_status_host", None) is None:
            self._scan_status_host = torch.zeros((1,), dtype=torch.int32).pin_memory()
        self._scan_status_host.copy_(status, non_blocking=True)
        mem_idx = mem_idx_buf.clone()                  # the buffer is overwritten by the next call
        hip.scan.last_tie = tie
        return mem_idx

    def _can_stream_image(self, patches):
        """ONE image on the fused 1x32x32 trunk: trunk + logits as one persistent launch beside a resident loop."""
        if (not self.is_image or patches.shape[0] != 1 or self.encoder.training or not patches.is_contiguous()
                or os.environ.get("IPSX_OVERLAP_SCAN", "1") == "0" or os.environ.get("IPSX_SCAN_PERSIST", "1") == "0"
                or os.environ.get("IPSX_IMAGE_STREAM", "1") == "0" or hip.dedup_blank() or hip.kernels_serialised()
                or patches.shape[1] < self.M + 2 * self.I):
            return False
        ca = self.transf.crs_attn
        if self._plan is None:
            self._plan = hip.EncoderPlan(self.encoder, self.is_image)
        return (hip.scan_persistent_supported(self.M, self.I, ca.H, ca.n_token)
                and self._plan.image_stream_supported(patches.shape, self.D, ca.H * ca.n_token)
                and ca.folded_query().dtype == torch.float32)

    def _select_image_stream(self, patches, pos_enc):
        """One image (the reference's eager-sequential mode, `B_seq = 1`): the loop is launched once, up front, as a
        persistent kernel on a compute unit of its own; the trunk's workgroups - one per remaining unit - pull four, later
        two patches at a time, encode them, compute their logits and publish them (ipsx_trunk_stream).  No parts, no launch
        between trunk and logits, no loop workgroup looking for a free unit beside a trunk launch, and what is exposed of
        the loop is the iteration after the last patch: 2,500 patches in 0.90 ms whatever else is in the stream (the parts
        of _small_batch_split: 0.89 ms when the loop's workgroup finds its free unit, 1.00 when it does not - as under
        bench.py's own event records)."""
        N = patches.shape[1]
        M, I, dev = self.M, self.I, patches.device
        ca = self.transf.crs_attn
        vq, R = ca.folded_query(), ca.H * ca.n_token
        n_iter = math.ceil((N - M) / I)
        self._device_patches = None
        side, main = self._scan_side_stream(dev)
        bkey = ("image stream", N, M, I, R, self.D, str(dev))
        if getattr(self, "_img_bufs_key", None) != bkey:       # (kept between calls: see _select_features_persistent)
            self._img_bufs = (torch.empty((1, N, R), dtype=torch.float32, device=dev),
                              torch.empty((1, M), dtype=torch.int64, device=dev),
                              torch.empty((1, N, self.D), dtype=torch.float32, device=dev),
                              # tie flag | progress word | status | the stream's control words: ONE fill per call
                              torch.zeros((3 + self._plan.image_stream_ctl_words(N),), dtype=torch.int32, device=dev))
            self._img_bufs_key = bkey
            for t in self._img_bufs:
                t.record_stream(side)
        logits, mem_idx_buf, emb_buf, zeroed = self._img_bufs
        tie, words, ctl = zeroed[:1], zeroed[1:3], zeroed[3:]
        mirror = getattr(self, "_scan_status_host", None)
        if mirror is not None and int(mirror.item()) & 1 and not getattr(self, "_scan_timeout_warned", False):
            import warnings
            warnings.warn("the persistent selection loop of an earlier ips() call timed out waiting for rows and was "
                          "redone with per-call launches (results valid; IPSX_SCAN_PERSIST=0 avoids the wait)")
            self._scan_timeout_warned = True
        zeroed.zero_()
        ready, status = words[:1], words[1:2]
        self._scan_status = status
        side.wait_stream(main)
        with torch.cuda.stream(side):
            hip.scan_persistent(logits, M, I, ca.H, ca.n_token, mem_idx_buf, tie, ready, status)
        hip.scan_gate(status)                      # the trunk must not take the compute units before the loop has its own
        pos = pos_enc[0] if self.use_pos else None
        self._plan.image_stream(patches[0], pos, vq, R, emb_buf[0], logits[0], ctl, ready)
        hip.publish_rows(ready, N)                 # (whatever the last finishers left to each other; the launch is over)
        self._emb_parts = [emb_buf]
        main.wait_stream(side)
        hip.scan_range_if(logits, M, I, ca.H, ca.n_token, 0, n_iter, mem_idx_buf, tie, status, 1)   # no-op unless timed out
        if getattr(self, "_scan_status_host", None) is None:
            self._scan_status_host = torch.zeros((1,), dtype=torch.int32).pin_memory()
        self._scan_status_host.copy_(status, non_blocking=True)
        mem_idx = mem_idx_buf.clone()                  # the buffer is overwritten by the next call
        hip.scan.last_tie = tie
        return mem_idx

    def _select_hip_overlapped(self, patches, pos_enc):
        B, N = patches.shape[:2]
        M, I, dev = self.M, self.I, patches.device
        ca = self.transf.crs_attn
        vq, R = ca.folded_query(), ca.H * ca.n_token
        n_iter = math.ceil((N - M) / I)
        from ..dist import part_iterations
        if self._plan is None:
            self._plan = hip.EncoderPlan(self.encoder, self.is_image)
        if (not self.is_image and patches.is_contiguous() and B <= int(os.environ.get("IPSX_PERSIST_MAX_B", "16"))
                and os.environ.get("IPSX_SCAN_PERSIST", "1") != "0"
                and not hip.kernels_serialised()      # (counter collection, serialising debug switches: it could only time out)
                and hip.scan_persistent_supported(M, I, ca.H, ca.n_token)):
            return self._select_features_persistent(patches, pos_enc)
        indexed = self.is_image and patches.is_contiguous() and self._plan.fused(patches.shape)
        # image encoders: parts shrinking towards the end (only the last scan is exposed); a small batch: the whole rounds
        # of the fused trunk first, the loop over them beside the remainder (_small_batch_split); feature inputs without
        # the persistent loop (more slides than IPSX_PERSIST_MAX_B, candidate sets beyond the LDS, serialised kernels):
        # equal parts, each slide's share of a launch sized to fill the GPU once
        edges = None
        if self.is_image and B * N < 32768 and n_iter < 100:
            edges, its = self._small_batch_split(B, N)
        elif self.is_image:
            its = part_iterations(n_iter, self._OVERLAP_PARTS)
        else:
            its = self._feature_parts_plain(B, N)
        P = len(its) - 1
        if edges is None:                                      # parts cut at the loop's chunk boundaries
            edges = [0] + [min(N, M + it * I) for it in its[1:]]
            edges[-1] = N
        key = (B, N, tuple(edges), str(dev))
        if indexed and getattr(self, "_part_index_key", None) != key:  # int32 patch indices of every part, cached
            rows = torch.arange(B, device=dev, dtype=torch.int32).unsqueeze(1) * N
            self._part_index = [(rows + torch.arange(edges[k], edges[k + 1], device=dev, dtype=torch.int32)).reshape(-1)
                                for k in range(P)]
            self._part_index_key = key
        side, main = self._scan_side_stream(dev)
        flat = patches.reshape(B * N, *patches.shape[2:]) if indexed else None
        # per-call device buffers are kept between calls of the same shape (see _select_features_persistent)
        bkey = (B, N, M, I, R, str(dev))
        if getattr(self, "_scan_bufs_key", None) != bkey:
            self._scan_bufs = (torch.empty((B, N, R), dtype=torch.float32, device=dev),
                               torch.empty((B, M), dtype=torch.int64, device=dev),
                               torch.zeros((B,), dtype=torch.int32, device=dev),
                               hip.scan_workspace(B, M, I, ca.H, ca.n_token, dev))   # None unless M + I exceeds the LDS
            self._scan_bufs_key = bkey
            for t in self._scan_bufs:
                if t is not None:
                    t.record_stream(side)
        logits, mem_idx_buf, tie, scan_ws = self._scan_bufs
        tie.zero_()
        self._emb_parts = parts = []
        side.wait_stream(main)
        for k in range(P):
            lo, hi = edges[k], edges[k + 1]
            if indexed:
                emb = self._plan.encode_indexed(flat, self._part_index[k]).view(B, hi - lo, -1)
            else:
                emb = self._embed(patches[:, lo:hi].reshape(-1, *patches.shape[2:])).view(B, hi - lo, -1)
            parts.append(emb)
            pos = pos_enc[:, lo:hi] if self.use_pos else None
            if k == P - 1:
                # the last part has nothing to run beside: its logits and iterations stay on the main stream (one
                # cross-stream hand-over less on the critical path; it only has to follow the side stream's earlier parts)
                main.wait_stream(side)
                hip.logits(emb, pos, vq, R, out=logits[:, lo:hi])
                hip.scan_range(logits, M, I, ca.H, ca.n_token, its[k], its[k + 1], mem_idx_buf, tie, scan_ws)
                continue
            # logits and loop of this part on the side stream: the main stream goes straight on to the next part's encoder
            done = torch.cuda.Event()
            done.record(main)
            emb.record_stream(side)
            with torch.cuda.stream(side):
                side.wait_event(done)
                hip.logits(emb, pos, vq, R, out=logits[:, lo:hi])
                hip.scan_range(logits, M, I, ca.H, ca.n_token, its[k], its[k + 1], mem_idx_buf, tie, scan_ws)
        main.wait_stream(side)
        mem_idx = mem_idx_buf.clone()                  # the buffer is overwritten by the next call
        hip.scan.last_tie = tie
        return mem_idx

    # lazy loading (reference :204-206,223,245-247): the reference moves M / I patches per iteration to
    # bound device memory.  Here the host tensor is streamed in a few large slabs on a copy stream while the
    # previous slab is being encoded (PCIe Gen5 moves 4 KiB patches ~5x faster than the fp32 encoder consumes
    # them, so the transfer hides behind the encoder).  With 288 GB of HBM the slabs are kept (up to
    # IPSX_LAZY_KEEP_MB, default 16 GiB) so the M winners are gathered on the device; beyond that the final
    # gather happens on the host exactly as in the reference.
    _LAZY_SLAB_BYTES = 48 << 20

    def _lazy_slabs(self, patches):
        B, N = patches.shape[:2]
        row_bytes = patches[0, 0].numel() * patches.element_size()
        slab_bytes = int(os.environ.get("IPSX_LAZY_SLAB_MB", "0")) << 20 or self._LAZY_SLAB_BYTES
        per = max(1, min(N, slab_bytes // max(1, B * row_bytes)))
        # The first slab is the only copy nothing hides, so the slabs GROW: the copy engine moves patches ~3x faster than
        # the fp32 encoder consumes them, i.e. a slab up to 3x the previous one still arrives behind the previous
        # one's encoding (1/6, 1/2, then full slabs: 8 / 24 / 48 MB - the first is one workgroup round of the fused trunk).
        spans, lo = [], 0
        for size in (max(1, per // 6), max(1, per // 2)):
            if N - lo > per:
                spans.append((lo, lo + size))
                lo += size
        spans += [(a, min(a + per, N)) for a in range(lo, N, per)]
        keep = patches.numel() * patches.element_size() <= int(os.environ.get("IPSX_LAZY_KEEP_MB", "16384")) << 20
        dev = self.device
        # the device-side buffers are kept between calls of the same shape: allocated afresh, a block that the copy stream
        # has used cannot be recycled until that stream's work is known to be over, and a host that runs many calls ahead
        # of the GPU piles up one image batch per call (8.6 GB after 300 un-synchronised calls of the headline batch)
        bkey = (keep, tuple(patches.shape), per, patches.dtype, str(dev))
        if getattr(self, "_lazy_bufs_key", None) != bkey:
            self._lazy_bufs = [torch.empty(patches.shape, dtype=patches.dtype, device=dev)] if keep else \
                [torch.empty((B, per) + tuple(patches.shape[2:]), dtype=patches.dtype, device=dev) for _ in range(2)]
            self._lazy_bufs_key = bkey
        if keep:
            store = self._lazy_bufs[0]
            self._device_patches = store
            dst = lambda k, lo, hi: store[:, lo:hi]
        else:
            ring = self._lazy_bufs
            dst = lambda k, lo, hi: ring[k % 2][:, :hi - lo]
        copy_stream = getattr(self, "_copy_stream", None)
        if copy_stream is None or copy_stream.device != torch.device(dev):
            copy_stream = self._copy_stream = torch.cuda.Stream(device=dev)
        main = torch.cuda.current_stream(dev)
        # `store` / `ring` come from the main stream's allocator pool: a block freed in Python a moment ago may still be
        # read by kernels queued on the main stream (the previous step's backward / optimizer), so the copy stream must
        # not write it before the main stream got that far - and the allocator must know the copy stream uses it
        copy_stream.wait_stream(main)
        for buf in ([store] if keep else ring):
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

    def _select_aten(self, patches, pos_enc):
        """The reference's loop on stock ATen ops (CPU plumbing path)."""
        B, N = patches.shape[:2]
        D, device = self.D, self.device
        order = torch.arange(N, dtype=torch.int64, device=device).unsqueeze(0).expand(B, -1)
        mem_emb = mem_idx = None
        for lo, hi in self._chunks(N):
            part = patches[:, lo:hi].to(device)
            emb = self._embed(part.reshape(-1, *patches.shape[2:])).view(B, hi - lo, D)
            if mem_emb is None:
                mem_emb, mem_idx = emb, order[:, lo:hi]
                continue
            cand_emb = torch.cat((mem_emb, emb), dim=1)          # memory first, chunk after
            cand_idx = torch.cat((mem_idx, order[:, lo:hi]), dim=1)
            cand_pos = None
            if self.use_pos:
                cand_pos = cand_emb + torch.gather(pos_enc, 1, cand_idx.unsqueeze(-1).expand(-1, -1, D))
            mem_emb, mem_idx = self.score_and_select(cand_emb, cand_pos, self.M, cand_idx)
        self._mem_emb = mem_emb
        return mem_idx

    @property
    def last_mem_emb(self):
        """(B, M, D) embeddings (without positional encoding) of the patches the last ``ips()`` selected,
        as the encoder computed them in eval mode - or ``None`` (shortcut M >= N).  In eval mode
        ``encoder(mem_patch)`` in ``forward`` recomputes exactly these (reference :273 after :209/:227 with
        running BatchNorm statistics), so an eval loop can hand them back through ``forward(..., mem_emb=)``
        and skip the second encoder pass (SURVEY.md section 8 f, N-a).  Gathered on first use."""
        if self._mem_emb is None and self._emb_parts and self.last_mem_idx is not None:
            emb = self._emb_parts[0] if len(self._emb_parts) == 1 else torch.cat(self._emb_parts, dim=1)
            self._mem_emb = self._take(emb, self.last_mem_idx)
            self._emb_parts = None
        return self._mem_emb

    # ---------------------------------------------------------------- aggregation
    def forward(self, mem_patch, mem_pos=None, mem_emb=None):
        """Embed the M selected patches, aggregate, classify (reference :264-283).

        ``mem_emb`` (optional, not in the reference): embeddings of ``mem_patch`` already computed by ``ips()``
        in eval mode (``last_mem_emb``); used instead of a second encoder pass when the encoder is in eval
        mode, ignored in train mode where BatchNorm uses batch statistics and the pass carries gradients."""
        B, M = mem_patch.shape[:2]
        if mem_emb is None or self.encoder.training:
            mem_emb = self._embed(mem_patch.reshape(-1, *mem_patch.shape[2:])).view(B, M, -1)
        if torch.is_tensor(mem_pos):
            mem_emb = mem_emb + mem_pos
        return self.get_preds(self.transf(mem_emb))
