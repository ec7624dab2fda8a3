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
        self._selection = None        # the HIP selection pipelines (ips_amd/selection.py), built on first use
        self._device_patches = None   # lazy loading: the device copy of the host tensor, when it is kept
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

            src = self._device_patches if self._device_patches is not None else patches
            self._device_patches = None
            sel = self._selection
            done = sel.finish(src, pos_enc if self.use_pos else None) if sel is not None else None
            if done is not None:       # a resident loop's call ends in ONE launch: gathers, indices, status word (round 5)
                mem_idx, mem_patch, mem_pos = done
            else:
                if sel is not None:
                    mem_idx = sel.take_unfinished(mem_idx)
                mem_patch = self._take(src, mem_idx).to(device)
                mem_pos = self._take(pos_enc, mem_idx) if self.use_pos else None
                if sel is not None:
                    sel.after_call()
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

    @property
    def selection(self):
        """The HIP selection pipelines of this net (ips_amd/selection.py), built on first use."""
        if self._selection is None:
            from ..selection import Selection
            self._selection = Selection(self)
        return self._selection

    def _select_hip(self, patches, pos_enc):
        """encode -> logits -> selection loop on the ROCm device (which producer, how the loop follows it: selection.py);
        patches may still be on the host (lazy loading)."""
        return self.selection.select(patches, pos_enc)

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
