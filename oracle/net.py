"""CPU oracle, part 2: a whole VGG16-BN train / prune / validate step.  TEST INFRASTRUCTURE ONLY
(same import rule as oracle/ops.py).

Restates, on torch-CPU fp32, the per-minibatch op order of the reference's
Manager.train / Manager.validate (utils/manager.py:39-121) on the reference's VGG16-BN
topologies (models/vgg.py:95-154) using the per-layer functions of oracle/ops.py for
everything mask-related.  Used (a) to pin the step order against the golden trajectory
fixtures, (b) as the checker for the HIP path in tests/ and smoke(), (c) as the timed
"port" CPU baseline in bench.py.

Parity status: pinned against tests/golden/trajectory_*.npz and first_forward_vgg*.npz
(outputs of the reference run in the build container).
"""

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops

VGG16_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']


class _STEBinarize(torch.autograd.Function):
    """models/layers.py:11-23: hard threshold forward, identity backward."""

    @staticmethod
    def forward(ctx, pm, thr):
        return torch.from_numpy(ops.binarize(pm.detach().numpy(), thr))

    @staticmethod
    def backward(ctx, g):
        return g, None


class _Masked(nn.Module):
    def effective(self):
        if self.piggymask is None:
            return self.weight
        return _STEBinarize.apply(self.piggymask, self.threshold) * self.weight


class MaskedConv(_Masked):
    """CPU stand-in for SharableConv2d (models/layers.py:43-109)."""

    def __init__(self, cin, cout, k, stride=1, padding=0, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None
        self.piggymask = None
        self.threshold = ops.DEFAULT_THRESHOLD
        self.stride, self.padding = stride, padding

    def forward(self, x):
        return F.conv2d(x, self.effective(), self.bias, self.stride, self.padding)


class MaskedLinear(_Masked):
    """CPU stand-in for SharableLinear (models/layers.py:147-194)."""

    def __init__(self, fin, fout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(fout, fin))
        self.bias = nn.Parameter(torch.empty(fout))
        self.piggymask = None
        self.threshold = ops.DEFAULT_THRESHOLD

    def forward(self, x):
        return F.linear(x, self.effective(), self.bias)


class _Flatten(nn.Module):
    def __init__(self, n):
        super().__init__()
        self.n = n

    def forward(self, x):
        return x.view(-1, self.n)


class OracleVGG(nn.Module):
    """VGG16-BN with per-task heads; `variant` = 'cifar100' (32x32, models/vgg.py:95-122) or
    'imagenet' (224x224 + Dropout, models/vgg.py:124-154).  Module indices inside `features`
    equal the reference's, so parameter names line up with its state_dict."""

    def __init__(self, width=1.0, variant='cifar100', cfg=VGG16_CFG):
        super().__init__()
        seq, cin = [], 3
        for v in cfg:
            if v == 'M':
                seq.append(nn.MaxPool2d(2, 2))
            else:
                c = int(v * width)
                seq += [MaskedConv(cin, c, 3, padding=1, bias=False), nn.BatchNorm2d(c), nn.ReLU(inplace=True)]
                cin = c
        f = int(4096 * width)
        if variant == 'cifar100':
            flat = int(512 * width)
            seq += [_Flatten(flat), MaskedLinear(flat, f), nn.ReLU(True), MaskedLinear(f, f), nn.ReLU(True)]
        else:
            flat = int(512 * width) * 7 * 7
            seq += [_Flatten(flat), MaskedLinear(flat, f), nn.ReLU(True), nn.Dropout(),
                    MaskedLinear(f, f), nn.ReLU(True), nn.Dropout()]
        self.features = nn.Sequential(*seq)
        self.width = width
        self.datasets, self.classifiers = [], nn.ModuleList()
        self.head = None
        # init order and distributions of models/vgg.py:59-70
        for m in self.modules():
            if isinstance(m, MaskedConv):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, MaskedLinear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.constant_(m.bias, 0)

    def add_dataset(self, name, num_classes):          # models/vgg.py:81-88
        if name not in self.datasets:
            self.datasets.append(name)
            head = nn.Linear(int(4096 * self.width), num_classes)
            nn.init.normal_(head.weight, 0, 0.01)
            nn.init.constant_(head.bias, 0)
            self.classifiers.append(head)

    def set_dataset(self, name):                       # models/vgg.py:90-93
        self.head = self.classifiers[self.datasets.index(name)]

    def forward(self, x):
        return self.head(self.features(x))

    def masked_layers(self):
        return [(n, m) for n, m in self.named_modules() if isinstance(m, _Masked)]


class OraclePruner:
    """utils/prune.py SparsePruner over an OracleVGG; owners keyed by bare module name."""

    def __init__(self, model, owners, mode, cur, inference_idx, begin, end, frequency,
                 initial_sparsity, target_sparsity, weight_decay, width_mult=1.0):
        self.model, self.owners, self.mode = model, owners, mode
        self.cur, self.inference_idx = cur, inference_idx
        self.begin, self.end, self.frequency = begin, end, frequency
        self.initial, self.target, self.wd = initial_sparsity, target_sparsity, weight_decay
        self.width_mult = width_mult
        self.last_prune_step = begin

    def claim_free(self):                               # make_finetuning_mask
        self.cur += 1
        for n, _ in self.model.masked_layers():
            self.owners[n] = ops.claim_free(self.owners[n], self.cur)

    def route_torch(self):
        """Same routing with torch-CPU ops, statement for statement as utils/prune.py:195-211 runs on a
        CPU host (multi-threaded elementwise kernels) -- used by the timed CPU baseline."""
        for n, m in self.model.masked_layers():
            owner = torch.from_numpy(self.owners[n])
            if m.weight.grad is not None:
                m.weight.grad.data.add_(m.weight.data, alpha=self.wd)
                m.weight.grad.data[owner.ne(self.cur)] = 0
            if m.piggymask is not None and m.piggymask.grad is not None:
                if self.mode == 'finetune':
                    m.piggymask.grad.data[owner.eq(0) | owner.ge(self.cur)] = 0
                elif self.mode == 'prune':
                    m.piggymask.grad.data.fill_(0)

    def route(self):                                    # do_weight_decay_and_make_grads_zero
        for n, m in self.model.masked_layers():
            if m.weight.grad is None:
                continue
            gpm = None if m.piggymask is None or m.piggymask.grad is None else m.piggymask.grad.numpy()
            gw, gpm2 = ops.route_grads(m.weight.grad.numpy(), m.weight.data.numpy(), self.owners[n],
                                       self.cur, self.wd, gpm, self.mode)
            m.weight.grad.copy_(torch.from_numpy(gw))
            if gpm2 is not None:
                m.piggymask.grad.copy_(torch.from_numpy(gpm2))

    def gradually_prune(self, step):                    # utils/prune.py:78-92
        if ops.time_to_update(step, self.begin, self.end, self.last_prune_step, self.frequency):
            self.last_prune_step = step
            ratio = ops.adjust_sparsity(step, self.begin, self.end, self.initial, self.target)
            for n, m in self.model.masked_layers():
                self.owners[n], _, _ = ops.rank_prune(m.weight.data.numpy(), self.owners[n], self.cur, ratio)
            return ratio
        return ops.adjust_sparsity(self.last_prune_step, self.begin, self.end, self.initial, self.target)

    def apply_mask(self):
        for n, m in self.model.masked_layers():
            m.weight.data.copy_(torch.from_numpy(ops.apply_mask(m.weight.data.numpy(), self.owners[n], self.inference_idx)))

    def sparsity(self):
        return ops.sparsity([self.owners[n] for n, _ in self.model.masked_layers()], self.inference_idx)


def train_step(model, pruner, optimizer, x, target, prune_step=None, torch_routing=False):
    """One iteration of utils/manager.py:50-75.  Returns (logits, loss, prune_ratio or None)."""
    optimizer.zero_grad()
    out = model(x)
    loss = F.cross_entropy(out, target)
    loss.backward()
    if torch_routing:
        pruner.route_torch()
    else:
        pruner.route()
    optimizer.step()
    ratio = None
    if pruner.mode == 'prune':
        ratio = pruner.gradually_prune(prune_step)
    return out.detach(), float(loss.detach()), ratio


def make_task1(width, variant, mode, num_classes=5, lr=1e-2, begin=0, end=8, frequency=3,
               initial=0.0, target=0.3, wd=4e-5, seed=1):
    """Task-1 set-up of CPG_cifar100_main_normal.py:184-346 (model, owners, pruner, SGD-nesterov)."""
    torch.manual_seed(seed)
    model = OracleVGG(width, variant)
    model.add_dataset('t1', num_classes)
    model.set_dataset('t1')
    owners = {n: np.zeros(tuple(m.weight.shape), dtype=np.uint8) for n, m in model.masked_layers()}
    if mode == 'finetune':
        pruner = OraclePruner(model, owners, mode, 0, 1, begin, end, frequency, initial, target, wd, width)
        pruner.claim_free()
    else:
        for n in owners:
            owners[n][...] = 1
        pruner = OraclePruner(model, owners, mode, 1, 1, begin, end, frequency, initial, target, wd, width)
    opt = torch.optim.SGD(list(model.parameters()), lr=lr, momentum=0.9, nesterov=True, weight_decay=0.0)
    return model, pruner, opt
