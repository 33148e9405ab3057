"""Per-step helpers of the reference's utils/__init__.py:6-49 (Optimizers, Metric, accuracy).

Host-side bookkeeping only; values equal the reference's, but nothing here forces a device
synchronisation per step: Metric accumulates on the device and is read when `.avg` is asked for.
"""
import gc

import torch


def settle_host_gc():
    """Collect once, then move everything alive -- the model, its masks, torch itself: about a million tracked objects -- into the
    collector's permanent generation (`gc.freeze`).  The step loop allocates a few thousand container objects per step (autograd
    contexts, argument tuples), so CPython starts a full generation-2 pass every few dozen steps, and that pass walks every tracked
    object: 80 ms measured on the GPU box's host (tools/step_times.py), most of a VGG16 step, during which no kernel is enqueued.
    After the freeze the same pass only visits what was allocated since.  PROCESS-WIDE side effects, hence opt-in: every object alive
    in the embedding application is frozen too (cyclic garbage among them -- e.g. an old model's tensors in a reference cycle -- is only
    reclaimed at the next call: unfreeze + collect), and a freeze the application did itself is undone.  Called by
    CPGSession(freeze_gc=True) whenever it has (re)built a model, and by bench.py after its warm-up (disclosed in its JSON line)."""
    gc.unfreeze()
    gc.collect()
    gc.freeze()


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

    def all_reduce_(self, group=None):
        """Sum the running totals over the ranks of a data-parallel run: `avg` then is the mean over the GLOBAL batches, what the
        reference's single-process nn.DataParallel computes on the gathered output (utils/manager.py:60-62) -- and identical on
        every rank, so decisions taken on it (grow, stop the sweep, early stop) cannot diverge between ranks."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return self
        both = torch.stack([self.sum.float().reshape(()), self.n.float().reshape(())])
        if dist.get_backend(group) == 'nccl' and not both.is_cuda:
            both = both.cuda()
        dist.all_reduce(both, op=dist.ReduceOp.SUM, group=group)
        self.sum, self.n = both[0], both[1]
        return self


def classification_accuracy(output, target):
    """Top-1 accuracy of a batch (utils/__init__.py:46-49); stays on the device (no .cpu() stall)."""
    pred = output.max(1, keepdim=True)[1]
    return pred.eq(target.view_as(pred)).float().mean()
