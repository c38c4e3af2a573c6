"""A thin epoch-level caller of the ST-GCN classifiers: the loop semantics of the reference's
``Processor`` (classifier_stgcn_real_only/utils/processor.py) around the step_amd modules, host Python only.

What is reproduced, statement by statement:
  weights_init             processor.py:117-129  (dispatch on the CLASS NAME: 'Conv1d' / 'Conv2d' / 'BatchNorm')
  step epochs / adjust_lr  processor.py:238,260-268
  per_train                processor.py:309-335  (train(); forward, CrossEntropyLoss, zero_grad, backward, step)
  per_test / top-k         processor.py:346-379,294-307 (eval(); no_grad; hit = label among the k largest outputs)
  train                    processor.py:381-409  (eval every `eval_interval` epochs and after the last one; a
                                                  checkpoint `epoch{E}_acc{A:.2f}_model.pth.tar` on every improvement)
  best-checkpoint lookup   processor.py:142-148,447-455 (epoch and accuracy parsed back out of the file name)
  loaders                  main.py:78-88 (batch_size, shuffle=True, drop_last=True for BOTH splits)

What is not: logging (torchlight), h5 loading, plots.  The data are tensors already shaped (N, C, T, V, 1)
(utils/loader.py:65-66); batches are index slices of a resident tensor instead of a DataLoader's workers.
"""
import math
import os
import re

import numpy as np
import torch
import torch.nn as nn

CKPT_FORMAT = "epoch{}_acc{:.2f}_model.pth.tar"
_CKPT_RE = re.compile(r"^epoch(\d+)_acc(\d+(?:\.\d+)?)_model\.pth\.tar$")


def weights_init(m):
    """processor.py:117-129: N(0, 0.02) weights / zero bias for every module whose class name contains 'Conv1d' or
    'Conv2d', N(1, 0.02) / zero for 'BatchNorm'.  'ConvTranspose2d' contains neither, so the decoder's transposed
    convolutions keep torch's default initialisation - as in the reference."""
    name = type(m).__name__
    if "Conv1d" in name or "Conv2d" in name:
        m.weight.data.normal_(0.0, 0.02)
        if m.bias is not None:
            m.bias.data.fill_(0)
    elif "BatchNorm" in name:
        m.weight.data.normal_(1.0, 0.02)
        m.bias.data.fill_(0)


def step_epochs(num_epoch, fractions=(0.5, 0.75, 0.875)):
    return [math.ceil(float(num_epoch * f)) for f in fractions]


def lr_at(epoch, base_lr, steps):
    """adjust_lr: at an epoch listed in `steps` the rate becomes base_lr * 0.1 ** #(steps already reached);
    between them it keeps its value.  Returns None when the epoch is not a step epoch (nothing to change)."""
    if epoch not in steps:
        return None
    return base_lr * (0.1 ** int(np.sum(epoch >= np.array(steps))))


def topk_accuracy(result, labels, k):
    """show_topk: argsort ascending, a hit when the label is among the last k indices; percent"""
    rank = np.asarray(result).argsort()
    hits = [l in rank[i, -k:] for i, l in enumerate(np.asarray(labels))]
    return 100.0 * sum(hits) * 1.0 / len(hits)


def best_checkpoint(work_dir):
    """(epoch, accuracy, path) of the checkpoint to reload.  A file is written only when the accuracy improved, so
    the one with the highest epoch is the best; the reference takes the last name `os.listdir` returns that has an
    underscore (processor.py:142-148), which is that file only when the listing happens to be sorted."""
    found = []
    for name in os.listdir(work_dir):
        m = _CKPT_RE.match(name)
        if m:
            found.append((int(m.group(1)), float(m.group(2)), os.path.join(work_dir, name)))
    if not found:
        raise FileNotFoundError("no epoch*_acc*_model.pth.tar in %s" % work_dir)
    return max(found)


class EpochRunner:
    """Processor(args, data_loader, C, num_classes, graph_dict) of the reference, minus logging.

    `train_set` / `test_set` are (data (N, C, T, V, 1) float tensor, labels (N,) long tensor); they are moved to
    `device` once.  `model` is a step_amd Classifier (or anything with the same call surface) already built."""

    def __init__(self, model, train_set, test_set, work_dir, num_epoch, batch_size=8, base_lr=0.1,
                 weight_decay=1e-4, step=(0.5, 0.75, 0.875), optimizer="Adam", nesterov=True, topk=(1,),
                 eval_interval=1, start_epoch=0, device="cuda:0", seed=0, init=weights_init, log=None):
        self.model = model.to(device)
        if init is not None:
            self.model.apply(init)                                  # processor.py:235
        self.device = device
        self.train_set = tuple(t.to(device) for t in train_set)
        self.test_set = tuple(t.to(device) for t in test_set)
        self.work_dir = work_dir
        os.makedirs(work_dir, exist_ok=True)
        self.num_epoch, self.start_epoch = num_epoch, start_epoch
        self.batch_size, self.base_lr = batch_size, base_lr
        self.topk, self.eval_interval = tuple(topk), eval_interval
        self.loss = nn.CrossEntropyLoss()
        self.steps = step_epochs(num_epoch, step)
        if optimizer == "SGD":
            self.optimizer = torch.optim.SGD(self.model.parameters(), lr=base_lr, momentum=0.9, nesterov=nesterov,
                                             weight_decay=weight_decay)
        elif optimizer == "Adam":
            self.optimizer = torch.optim.Adam(self.model.parameters(), lr=base_lr, weight_decay=weight_decay)
        else:
            raise ValueError(optimizer)
        self.lr = base_lr
        self.best_accuracy = np.zeros((1, max(self.topk)))
        self.accuracy_updated = False
        self.best_epoch = None
        self.best_loss = math.inf
        self.meta = {"epoch": 0, "iter": 0}
        self.history = []                                           # one dict per epoch
        self.result = self.label = None
        self._gen = torch.Generator().manual_seed(seed)
        self._log = log or (lambda s: None)

    # main.py:78-88: shuffle=True, drop_last=True on both splits
    def _batches(self, split):
        data, labels = split
        order = torch.randperm(data.shape[0], generator=self._gen).to(data.device)
        for i in range(0, data.shape[0] - self.batch_size + 1, self.batch_size):
            idx = order[i:i + self.batch_size]
            yield data[idx], labels[idx]

    def adjust_lr(self):
        lr = lr_at(self.meta["epoch"], self.base_lr, self.steps)
        if lr is not None:
            for group in self.optimizer.param_groups:
                group["lr"] = lr
            self.lr = lr

    def per_train(self):
        self.model.train()
        self.adjust_lr()
        losses = []
        for data, label in self._batches(self.train_set):
            output, _ = self.model(data.float())
            loss = self.loss(output, label.long())
            self.optimizer.zero_grad()
            loss.backward()
            self.optimizer.step()
            losses.append(loss.item())
            self.meta["iter"] += 1
        return float(np.mean(losses))

    def per_test(self):
        self.model.eval()
        losses, results, labels = [], [], []
        for data, label in self._batches(self.test_set):
            with torch.no_grad():
                output, _ = self.model(data.float())
            results.append(output.cpu().numpy())
            losses.append(self.loss(output, label.long()).item())
            labels.append(label.cpu().numpy())
        self.result, self.label = np.concatenate(results), np.concatenate(labels)
        accs = {}
        for k in self.topk:
            acc = topk_accuracy(self.result, self.label, k)
            accs[k] = acc
            if acc > self.best_accuracy[0, k - 1]:
                self.best_accuracy[0, k - 1] = acc
                self.accuracy_updated = True
            else:
                self.accuracy_updated = False
        return float(np.mean(losses)), accs

    def checkpoint_name(self, epoch):
        return os.path.join(self.work_dir, CKPT_FORMAT.format(epoch, self.best_accuracy.item()))

    def train(self):
        for epoch in range(self.start_epoch, self.num_epoch):
            self.meta["epoch"] = epoch
            rec = {"epoch": epoch, "train_loss": self.per_train(), "lr": self.lr}
            if epoch % self.eval_interval == 0 or epoch + 1 == self.num_epoch:
                rec["test_loss"], rec["accuracy"] = self.per_test()
            if self.accuracy_updated:                       # (stays set from the last evaluated epoch, as in the reference)
                path = self.checkpoint_name(epoch)
                torch.save(self.model.state_dict(), path)
                rec["saved"] = path
                if rec.get("test_loss", math.inf) < self.best_loss:
                    self.best_loss = rec["test_loss"]
                self.best_epoch = epoch
            self.history.append(rec)
            self._log("epoch %d: %s" % (epoch, rec))
        return self.history

    def load_best_model(self):
        if self.best_epoch is None:
            self.best_epoch, acc, path = best_checkpoint(self.work_dir)
        else:
            path = self.checkpoint_name(self.best_epoch)
        self.model.load_state_dict(torch.load(path, map_location=self.device, weights_only=True))
        return path


def synthetic_gaits(n=342, frames=75, num_classes=4, seed=0, test_size=0.1):
    """Stand-in for the 342 real Emotion-Gait sequences (an external download, absent here): root-relative
    16-joint walks at T = `frames` in the real data's range (about [-0.7, 0.5], info.txt:3-4) whose stride amplitude
    and frequency depend on the class, so that a classifier has something to learn.  Returns
    ((train_x, train_y), (test_x, test_y)) with x (N, 3, T, 16, 1), split 90 / 10 like main.py:68-72."""
    g = torch.Generator().manual_seed(seed)
    labels = torch.arange(n) % num_classes
    t = torch.arange(frames, dtype=torch.float32)[None, :, None]
    phase = torch.rand(n, 1, 16, generator=g) * 6.2832
    freq = (0.18 + 0.05 * labels.float())[:, None, None]
    amp = (0.10 + 0.06 * labels.float())[:, None, None]
    base = torch.randn(1, 1, 16, 3, generator=g) * 0.15
    x = base.expand(n, frames, 16, 3).clone()
    x[..., 0] += amp * torch.sin(freq * t + phase)
    x[..., 1] += 0.5 * amp * torch.cos(freq * t + phase)
    x += 0.02 * torch.randn(n, frames, 16, 3, generator=g)
    x = x - x[:, :, :1, :]                                          # root-relative (common.py:105-112)
    data = x.permute(0, 3, 1, 2).unsqueeze(-1).contiguous()         # (N, 3, T, 16, 1), loader.py:65-66
    order = torch.randperm(n, generator=g)
    n_test = int(math.ceil(n * test_size))
    te, tr = order[:n_test], order[n_test:]
    return (data[tr], labels[tr]), (data[te], labels[te])
