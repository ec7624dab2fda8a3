"""Caller-side helpers with the names and behaviour of /root/reference/utils/utils.py.

They sit on either side of the hot path: the attribute bag the
configuration is read into, the cosine learning-rate schedule, the patch-axis shuffles
(which must consume torch's RNG streams exactly like the reference, see ips_amd/shuffle.py)
and the per-epoch statistics.  Nothing here touches the device.
"""

import math
from collections import defaultdict

import numpy as np
from torch import nn

from ..shuffle import shuffle_batch, shuffle_instance  # noqa: F401  (utils/utils.py:33-58)


class Struct:
    """Attribute bag built from the YAML dict (utils/utils.py:10-12)."""

    def __init__(self, **entries):
        self.__dict__.update(entries)


def adjust_learning_rate(n_epoch_warmup, n_epoch, max_lr, optimizer, dloader, step):
    """Linear warm-up, then cosine decay to max_lr / 1000 (utils/utils.py:14-31); sets param group 0 only."""
    per_epoch = len(dloader)
    total, warm = int(n_epoch * per_epoch), int(n_epoch_warmup * per_epoch)
    if step < warm:
        lr = max_lr * step / warm
    else:
        q = 0.5 * (1 + math.cos(math.pi * (step - warm) / (total - warm)))
        lr = max_lr * q + max_lr * 0.001 * (1 - q)
    optimizer.param_groups[0]['lr'] = lr


def _accuracy(y_true, y_pred):
    y_true, y_pred = np.asarray(y_true), np.asarray(y_pred)
    return float((y_true == y_pred).mean())


def _auc(y_true, y_score):
    """Area under the ROC curve by the rank statistic, average ranks for tied scores
    (what sklearn.metrics.roc_auc_score returns for binary labels, utils/utils.py:108-112)."""
    y_true = np.asarray(y_true).reshape(-1).astype(bool)
    y_score = np.asarray(y_score, dtype=np.float64).reshape(-1)
    n_pos, n_neg = int(y_true.sum()), int((~y_true).sum())
    if n_pos == 0 or n_neg == 0:
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    order = np.argsort(y_score, kind="mergesort")
    s = y_score[order]
    ranks = np.empty(len(s), dtype=np.float64)
    start = 0
    for end in range(1, len(s) + 1):                     # average rank of every run of equal scores
        if end == len(s) or s[end] != s[start]:
            ranks[order[start:end]] = 0.5 * (start + end - 1) + 1.0
            start = end
    return float((ranks[y_true].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


class Logger(nn.Module):
    """Per-iteration losses / predictions, reduced to per-epoch loss and metric (utils/utils.py:60-143).

    Same attributes and methods as the reference's (``losses_it``, ``losses_epoch``, ``y_preds``,
    ``y_trues``, ``metrics``; ``update``, ``compute_metric``, ``print_stats``)."""

    def __init__(self, task_dict):
        super().__init__()
        self.task_dict = task_dict
        self.losses_it = defaultdict(list)
        self.losses_epoch = defaultdict(list)
        self.y_preds = defaultdict(list)
        self.y_trues = defaultdict(list)
        self.metrics = defaultdict(list)

    def update(self, next_loss, next_y_pred, next_y_true):
        for task in self.task_dict.values():
            t, kind = task['name'], task['metric']
            self.losses_it[t].append(next_loss[t])
            if kind == 'accuracy':
                self.y_preds[t].extend(np.argmax(next_y_pred[t], axis=-1))
            elif kind in ('multilabel_accuracy', 'auc'):
                self.y_preds[t].extend(next_y_pred[t].tolist())
            self.y_trues[t].extend(next_y_true[t])

    def compute_metric(self):
        for task in self.task_dict.values():
            t, kind = task['name'], task['metric']
            self.losses_epoch[t].append(np.mean(self.losses_it[t]))
            if kind == 'accuracy':
                self.metrics[t].append(_accuracy(self.y_trues[t], self.y_preds[t]))
            elif kind == 'multilabel_accuracy':
                hit = np.where(np.array(self.y_preds[t]) >= 0.5, 1., 0.) == np.array(self.y_trues[t])
                self.metrics[t].append(np.all(hit, axis=-1).sum() / hit.shape[0])
            elif kind == 'auc':
                self.metrics[t].append(_auc(self.y_trues[t], self.y_preds[t]))
            self.losses_it[t], self.y_preds[t], self.y_trues[t] = [], [], []

    def print_stats(self, epoch, train, **kwargs):
        line = ('Train' if train else 'Test') + " Epoch: {} \n".format(epoch + 1)
        total = 0
        for task in self.task_dict.values():
            t = task['name']
            loss, metric = self.losses_epoch[t][epoch], self.metrics[t][epoch]
            total += loss
            line += "task: {}, mean loss: {:.5f}, {}: {:.5f}, ".format(t, loss, task['metric'], metric)
        line += "avg. loss over tasks: {:.5f}".format(total / len(self.task_dict.values()))
        for k, v in kwargs.items():
            line += ", {}: {}".format(k, v)
        print(line + "\n")
