"""The loops that call the hot path: ``train_one_epoch`` / ``evaluate`` and their batch-assembly helpers,
with the names, signatures and results of /root/reference/training/iterative.py.

What they do (reference lines in brackets): every loader item is a sequence of ``B_seq`` images whose N
patches go through ``net.ips`` under no-grad [135, 211]; the M winners of each image are written into a
``B``-image buffer [31-50] until it is full or the loader is exhausted [143-150, 219-225]; then one
forward (+ backward and optimizer step when training) runs on the assembled ``(B, M, ...)`` batch [65-103].

Differences that do not change results:
  * the buffers are written in place on the device (the reference does the same through slice assignment);
  * in ``evaluate`` the embeddings ``ips()`` already computed for the winners ride along with the patches
    (``net.last_mem_emb``) and are handed to ``forward`` so the encoder does not run a second time on them
    (SURVEY.md section 8 f, N-a) - only when the network offers them, and bit-identical in eval mode;
  * the per-task loss values and predictions leave the device in ONE transfer per step instead of one
    ``.item()`` / ``.cpu()`` per task (each of those is a full device synchronisation).
"""

import sys

import numpy as np
import torch

from ..utils.utils import adjust_learning_rate


def init_batch(device, conf):
    """Zeroed (B, M, ...) patch buffer, (B, M, D) positional buffer or None, and per-task label buffers [7-29]."""
    shape = (conf.B, conf.M, conf.n_chan_in, *conf.patch_size) if conf.is_image else (conf.B, conf.M, conf.n_chan_in)
    mem_patch = torch.zeros(shape, device=device)
    mem_pos_enc = torch.zeros((conf.B, conf.M, conf.D), device=device) if conf.use_pos else None
    labels = {}
    for task in conf.tasks.values():
        if task['metric'] == 'multilabel_accuracy':
            labels[task['name']] = torch.zeros((conf.B, conf.n_class), dtype=torch.float32, device=device)
        else:
            labels[task['name']] = torch.zeros((conf.B,), dtype=torch.int64, device=device)
    return mem_patch, mem_pos_enc, labels


def fill_batch(mem_patch, mem_pos_enc, labels, data, n_prep, n_prep_batch,
               mem_patch_iter, mem_pos_enc_iter, conf):
    """Write one IPS result (n_seq images, len_seq <= M patches each) into rows [n_prep, n_prep + n_seq) [31-50]."""
    n_seq, len_seq = mem_patch_iter.shape[:2]
    mem_patch[n_prep:n_prep + n_seq, :len_seq] = mem_patch_iter
    if conf.use_pos:
        mem_pos_enc[n_prep:n_prep + n_seq, :len_seq] = mem_pos_enc_iter
    for task in conf.tasks.values():
        labels[task['name']][n_prep:n_prep + n_seq] = data[task['name']]
    return mem_patch, mem_pos_enc, labels, n_prep + n_seq, n_prep_batch + 1


def shrink_batch(mem_patch, mem_pos_enc, labels, n_prep, conf):
    """Drop the unfilled rows of the last batch of an epoch [52-63]."""
    mem_patch = mem_patch[:n_prep]
    if conf.use_pos:
        mem_pos_enc = mem_pos_enc[:n_prep]
    for task in conf.tasks.values():
        labels[task['name']] = labels[task['name']][:n_prep]
    return mem_patch, mem_pos_enc, labels


def compute_loss(net, mem_patch, mem_pos_enc, criterions, labels, conf, mem_emb=None):
    """Predictions, mean of the task losses, and [losses, predictions, labels] per task for the log [65-103].

    softmax heads: NLL of log(p + eps); sigmoid heads: BCE on flattened probabilities."""
    preds = net(mem_patch, mem_pos_enc) if mem_emb is None else net(mem_patch, mem_pos_enc, mem_emb=mem_emb)
    tasks = list(conf.tasks.values())
    loss, per_task, shown = 0, [], []
    for task in tasks:
        name = task['name']
        label, pred = labels[name], preds[name].squeeze(-1)
        if task['act_fn'] == 'softmax':
            task_loss = criterions[name](torch.log(pred + conf.eps), label)
        else:
            task_loss = criterions[name](pred.view(-1), label.view(-1).type(torch.float32))
        per_task.append(task_loss)
        shown.append(pred.detach())
        loss = loss + task_loss
    loss = loss / len(tasks)

    # one device->host transfer for everything the log needs
    flat = torch.cat([torch.stack(per_task).detach().float().reshape(-1)] + [p.float().reshape(-1) for p in shown]).cpu().numpy()
    task_losses, task_preds, task_labels = {}, {}, {}
    at = len(tasks)
    for k, (task, pred) in enumerate(zip(tasks, shown)):
        name = task['name']
        task_losses[name] = float(flat[k])
        task_preds[name] = flat[at:at + pred.numel()].reshape(tuple(pred.shape)).astype(np.float32, copy=True)
        at += pred.numel()
        task_labels[name] = labels[name].detach().cpu().numpy()
    return loss, [task_losses, task_preds, task_labels]


def _patches_of(data, device, conf):
    """Eager loading moves the whole patch tensor to the device, lazy loading leaves it on the host [122, 205].
    A loader built with ``collate_sparse`` (ips_amd/data/megapixel_mnist.py) delivers the non-zero pixels instead;
    they are scattered into the patch tensor on the device (eager by nature: 0.4 MB per image cross PCIe)."""
    if 'input' not in data and 'sparse' in data:
        return data['sparse'].to(device, non_blocking=True).patches(conf.patch_size, conf.patch_stride)
    return data['input'].to(device, non_blocking=True) if conf.eager else data['input']


def train_one_epoch(net, criterions, data_loader, optimizer, device, epoch, log_writer, conf):
    """One epoch of IPS + training steps [105-189]."""
    net.train()
    n_prep, n_prep_batch = 0, 0
    start_new_batch = True
    times = []
    track = bool(getattr(conf, 'track_efficiency', False))
    graphed = None
    if getattr(conf, 'hip_graph', False) and torch.device(device).type == 'cuda':     # addition: see training/graphed.py
        from .graphed import GraphedStep
        graphed = getattr(net, '_graphed_step', None)
        if graphed is None or graphed.optimizer is not optimizer:
            graphed = net._graphed_step = GraphedStep(net, criterions, optimizer, conf)

    for data_it, data in enumerate(data_loader, start=epoch * len(data_loader)):
        image_patches = _patches_of(data, device, conf)
        if start_new_batch:
            mem_patch, mem_pos_enc, labels = init_batch(device, conf)
            start_new_batch = False
            if track:
                start_event = torch.cuda.Event(enable_timing=True)
                end_event = torch.cuda.Event(enable_timing=True)
                start_event.record()

        mem_patch_iter, mem_pos_enc_iter = net.ips(image_patches)
        mem_patch, mem_pos_enc, labels, n_prep, n_prep_batch = fill_batch(
            mem_patch, mem_pos_enc, labels, data, n_prep, n_prep_batch, mem_patch_iter, mem_pos_enc_iter, conf)

        batch_full = n_prep == conf.B
        is_last_batch = n_prep_batch == len(data_loader)
        if not (batch_full or is_last_batch):
            continue
        if not batch_full:
            mem_patch, mem_pos_enc, labels = shrink_batch(mem_patch, mem_pos_enc, labels, n_prep, conf)

        adjust_learning_rate(conf.n_epoch_warmup, conf.n_epoch, conf.lr, optimizer, data_loader, data_it + 1)
        if graphed is not None:          # zero_grad, forward, losses, backward, optimizer.step replayed as one HIP graph
            loss, (task_losses, task_preds, task_labels) = graphed(mem_patch, mem_pos_enc, labels)
        else:
            optimizer.zero_grad()
            loss, (task_losses, task_preds, task_labels) = compute_loss(net, mem_patch, mem_pos_enc, criterions, labels, conf)
            loss.backward()
            optimizer.step()

        if track:
            end_event.record()
            torch.cuda.synchronize()
            if epoch == conf.track_epoch and data_it > 0 and not is_last_batch:
                times.append(start_event.elapsed_time(end_event))
                print("time: ", times[-1])

        log_writer.update(task_losses, task_preds, task_labels)
        n_prep = 0
        start_new_batch = True

    if track and epoch == conf.track_epoch:
        print("avg. time: ", np.mean(times))
        peak = torch.cuda.memory_stats()["allocated_bytes.all.peak"]
        print(f"Peak memory requirement: {peak / 1024 ** 3:.4f} GB")
        print("TORCH.CUDA.MEMORY_SUMMARY: ", torch.cuda.memory_summary())
        sys.exit()


@torch.no_grad()
def evaluate(net, criterions, data_loader, device, log_writer, conf):
    """One pass of IPS + forward in eval mode [191-231]."""
    net.eval()
    n_prep, n_prep_batch = 0, 0
    start_new_batch = True
    reuse = hasattr(net, 'last_mem_emb')                 # embeddings of the winners come with the winners

    for data in data_loader:
        image_patches = _patches_of(data, device, conf)
        if start_new_batch:
            mem_patch, mem_pos_enc, labels = init_batch(device, conf)
            mem_emb = torch.zeros((conf.B, conf.M, conf.D), device=device) if reuse else None
            start_new_batch = False

        mem_patch_iter, mem_pos_enc_iter = net.ips(image_patches)
        if reuse:
            emb_iter = net.last_mem_emb                   # None when ips() took the M >= N shortcut
            if emb_iter is None:
                reuse, mem_emb = False, None
            else:
                mem_emb[n_prep:n_prep + emb_iter.shape[0], :emb_iter.shape[1]] = emb_iter
        mem_patch, mem_pos_enc, labels, n_prep, n_prep_batch = fill_batch(
            mem_patch, mem_pos_enc, labels, data, n_prep, n_prep_batch, mem_patch_iter, mem_pos_enc_iter, conf)

        batch_full = n_prep == conf.B
        is_last_batch = n_prep_batch == len(data_loader)
        if not (batch_full or is_last_batch):
            continue
        if not batch_full:
            mem_patch, mem_pos_enc, labels = shrink_batch(mem_patch, mem_pos_enc, labels, n_prep, conf)
            mem_emb = mem_emb[:n_prep] if mem_emb is not None else None

        _, (task_losses, task_preds, task_labels) = compute_loss(net, mem_patch, mem_pos_enc, criterions, labels, conf,
                                                                 mem_emb=mem_emb)
        log_writer.update(task_losses, task_preds, task_labels)
        n_prep = 0
        start_new_batch = True
