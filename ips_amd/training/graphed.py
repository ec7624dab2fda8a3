"""The training step on the M selected patches as ONE HIP graph (SURVEY.md section 8 f, N-b).

After ``ips()`` a step is forward (batch-statistics BatchNorm, dropout), the task losses, backward and AdamW over
B x M small patches - a few hundred short kernels of stock PyTorch-ROCm, launch-bound on the host
(tools/train_step_breakdown.py: 4.6 ms eager at B=16, M=64).  All shapes are fixed by the configuration, so the whole
sequence is captured once into a HIP graph (``torch.cuda.CUDAGraph`` is hipGraph on ROCm) and replayed per step: the
arithmetic and its order are exactly the eager ones (same kernels, same arguments), only the launches are batched.

What capture needs and how it is met:
  * static inputs: the (B, M, ...) patch buffer, positional buffer and label buffers of ``init_batch`` are copied into
    graph-owned tensors before each replay (a partial last batch - ``shrink_batch`` - falls back to the eager step);
  * an optimizer whose step does not read host scalars: ``AdamW(capturable=True)`` with the learning rate in a device
    tensor, refreshed from the value ``adjust_learning_rate`` wrote (reference utils/utils.py:14-31);
  * warm-up iterations on a side stream before capture: parameters, BatchNorm statistics, optimizer state and the RNG
    offset are snapshotted before and restored after, so capturing does not train.
"""

import copy

import torch

from .. import hip
from .iterative import compute_loss


class GraphedStep:
    """``step(mem_patch, mem_pos_enc, labels) -> (loss, [task_losses, task_preds, task_labels])`` like the body of
    ``train_one_epoch`` (zero_grad, compute_loss, backward, optimizer.step), replayed from a HIP graph."""

    def __init__(self, net, criterions, optimizer, conf):
        self.net, self.criterions, self.optimizer, self.conf = net, criterions, optimizer, conf
        self.graph = None
        self._key = None

    # ------------------------------------------------------------------ eager fallback (also the warm-up body)
    def _eager(self, mem_patch, mem_pos_enc, labels):
        self.optimizer.zero_grad(set_to_none=False)
        loss, info = compute_loss(self.net, mem_patch, mem_pos_enc, self.criterions, labels, self.conf)
        loss.backward()
        self.optimizer.step()
        return loss, info

    def _make_capturable(self):
        """Every param group: ``capturable`` AdamW with the learning rate in a device tensor (one per group)."""
        self._lr_tensors = []
        for group in self.optimizer.param_groups:
            group['capturable'] = True
            lr = group['lr']
            t = lr if torch.is_tensor(lr) else torch.tensor(float(lr), dtype=torch.float32, device=self.net.device)
            group['lr'] = t
            self._lr_tensors.append(t)
        for p, st in self.optimizer.state.items():               # eager steps keep `step` on the host; capturable needs it
            if torch.is_tensor(st.get('step')) and st['step'].device != p.device:       # next to the parameter
                st['step'] = st['step'].to(device=p.device, dtype=torch.float32)

    def _capture(self, mem_patch, mem_pos_enc, labels):
        dev = mem_patch.device
        self._make_capturable()
        self.s_patch = mem_patch.clone()
        self.s_pos = mem_pos_enc.clone() if torch.is_tensor(mem_pos_enc) else None
        self.s_labels = {k: v.clone() for k, v in labels.items()}
        # snapshot everything the warm-up steps will move
        snap_model = copy.deepcopy(self.net.state_dict())
        had_state = any(len(st) for st in self.optimizer.state.values())
        snap_opt = {id(p): {k: (v.clone() if torch.is_tensor(v) else copy.deepcopy(v)) for k, v in st.items()}
                    for p, st in self.optimizer.state.items()}
        rng = torch.cuda.get_rng_state(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(3):
                self._eager(self.s_patch, self.s_pos, self.s_labels)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)

        # Gradients are NOT kept in static buffers (rounds 2-4 did: zero_grad(set_to_none=False)): autograd then ACCUMULATES
        # into them - one more elementwise kernel per parameter, 51 kernels / 0.18 ms of a 2.9 ms step
        # (tools/train_graph_vs_eager.py: why the replay was slower in device time than the eager step).  With the gradients
        # dropped in front of the capture, backward allocates them from the graph's private pool - the same addresses in
        # every replay - and hands the freshly computed tensors over, as the eager step does.
        self.optimizer.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            preds = self.net(self.s_patch, self.s_pos)
            tasks = list(self.conf.tasks.values())
            total, per_task, shown = 0, [], []
            for task in tasks:
                name = task['name']
                label, pred = self.s_labels[name], preds[name].squeeze(-1)
                if task['act_fn'] == 'softmax':
                    t_loss = self.criterions[name](torch.log(pred + self.conf.eps), label)
                else:
                    t_loss = self.criterions[name](pred.view(-1), label.view(-1).type(torch.float32))
                per_task.append(t_loss)
                shown.append(pred.detach())
                total = total + t_loss
            total = total / len(tasks)
            total.backward()
            self.optimizer.step()
            self.s_loss = total.detach()
            self.s_flat = torch.cat([torch.stack(per_task).detach().float().reshape(-1)] +
                                    [p.float().reshape(-1) for p in shown])
        self._shown_shapes = [tuple(p.shape) for p in shown]

        # undo the warm-up: weights, BatchNorm statistics, optimizer moments and step counters, RNG.  The optimizer's
        # state tensors are graph inputs, so they are restored IN PLACE: to the snapshot when the optimizer had already
        # taken steps (a re-capture after the batch geometry changed), to zero when the warm-up created them.
        self.net.load_state_dict(snap_model)
        for p, state in self.optimizer.state.items():
            before = snap_opt.get(id(p), {}) if had_state else {}
            for k, v in state.items():
                if torch.is_tensor(v):
                    if k in before and torch.is_tensor(before[k]):
                        v.copy_(before[k])
                    else:
                        v.zero_()
        torch.cuda.set_rng_state(rng, dev)
        hip.weights_changed()

    # ------------------------------------------------------------------ the step
    def __call__(self, mem_patch, mem_pos_enc, labels):
        conf = self.conf
        if not (mem_patch.is_cuda and mem_patch.shape[0] == conf.B):     # CPU, or the shrunk last batch of an epoch
            self._sync_lr()
            return self._eager(mem_patch, mem_pos_enc, labels)
        key = (tuple(mem_patch.shape), torch.is_tensor(mem_pos_enc), tuple(sorted(labels)))
        if self.graph is None or key != self._key:
            self._capture(mem_patch, mem_pos_enc, labels)
            self._key = key
        self._sync_lr()
        self.s_patch.copy_(mem_patch)
        if self.s_pos is not None:
            self.s_pos.copy_(mem_pos_enc)
        for k, v in labels.items():
            self.s_labels[k].copy_(v)
        self.graph.replay()
        # the replay moved every parameter and the BatchNorm running statistics without bumping their _version
        # counters: the packed-weight caches of the HIP path (EncoderPlan, folded query) must not survive it
        hip.weights_changed()

        flat = self.s_flat.cpu().numpy()
        tasks = list(conf.tasks.values())
        task_losses, task_preds, task_labels = {}, {}, {}
        at = len(tasks)
        for k, (task, shape) in enumerate(zip(tasks, self._shown_shapes)):
            name, n = task['name'], 1
            for d in shape:
                n *= d
            task_losses[name] = float(flat[k])
            task_preds[name] = flat[at:at + n].reshape(shape).copy()
            at += n
            task_labels[name] = labels[name].detach().cpu().numpy()
        return self.s_loss, [task_losses, task_preds, task_labels]

    def _sync_lr(self):
        """``adjust_learning_rate`` assigns a Python float to param_groups[0]['lr'] (reference utils/utils.py:31); the
        captured AdamW reads a device tensor: move the value over and put the tensor back."""
        for group, t in zip(self.optimizer.param_groups, getattr(self, "_lr_tensors", [])):
            if group['lr'] is t:
                continue
            t.fill_(float(group['lr']))
            group['lr'] = t
