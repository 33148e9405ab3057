"""Per-step helpers of the reference's utils/__init__.py:6-49 (Optimizers, Metric, accuracy).

Host-side bookkeeping only; values equal the reference's, but nothing here forces a device
synchronisation per step: Metric accumulates on the device and is read when `.avg` is asked for.
"""
import torch


class Optimizers(object):
    """Ordered list of optimizers stepped together (utils/__init__.py:6-27): SGD first, Adam second."""

    def __init__(self):
        self.optimizers, self.lrs = [], []

    def add(self, optimizer, lr):
        self.optimizers.append(optimizer)
        self.lrs.append(lr)

    def step(self):
        for opt in self.optimizers:
            opt.step()

    def zero_grad(self):
        for opt in self.optimizers:
            opt.zero_grad()

    def __getitem__(self, index):
        return self.optimizers[index]

    def __setitem__(self, index, value):
        self.optimizers[index] = value


class Metric(object):
    """Running weighted mean (utils/__init__.py:30-43).  `sum`/`n` follow the device of the values
    they are fed, so updating with a device scalar does not synchronise."""

    def __init__(self, name):
        self.name = name
        self.sum = torch.tensor(0.)
        self.n = torch.tensor(0.)

    def update(self, val, num):
        if isinstance(val, torch.Tensor):
            val = val.detach()
            if self.sum.device != val.device:
                self.sum, self.n = self.sum.to(val.device), self.n.to(val.device)
        self.sum = self.sum + val * num
        self.n = self.n + num

    @property
    def avg(self):
        return self.sum / self.n


def classification_accuracy(output, target):
    """Top-1 accuracy of a batch (utils/__init__.py:46-49); stays on the device (no .cpu() stall)."""
    pred = output.max(1, keepdim=True)[1]
    return pred.eq(target.view_as(pred)).float().mean()
