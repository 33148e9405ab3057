"""CPU oracle, part 2: a whole VGG16-BN (and, round 5, ResNet-50 / SphereNet-20) train / prune / validate step.
TEST INFRASTRUCTURE ONLY (same import rule as oracle/ops.py).

Restates, on torch-CPU fp32, the per-minibatch op order of the reference's
Manager.train / Manager.validate (utils/manager.py:39-121) on the reference's VGG16-BN
topologies (models/vgg.py:95-154) using the per-layer functions of oracle/ops.py for
everything mask-related.  Used (a) to pin the step order against the golden trajectory
fixtures, (b) as the checker for the HIP path in tests/ and smoke(), (c) as the timed
"port" CPU baseline in bench.py.

Parity status: pinned against tests/golden/trajectory_*.npz and first_forward_vgg*.npz
(outputs of the reference run in the build container); OracleResNet / OracleSphereNet against
first_forward_{resnet50,spherenet20}.npz, full_width_logits_*.npz and train_steps_*.npz
(tests/test_oracle_golden.py).
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


class _TaskHeads(nn.Module):
    """Per-task heads shared by the ResNet / SphereNet restatements (models/resnet.py:195-205, models/spherenet.py:176-199)."""

    def _init_heads(self):
        self.datasets, self.classifiers = [], nn.ModuleList()
        self.head = None

    def set_dataset(self, name):
        self.head = self.classifiers[self.datasets.index(name)]

    def masked_layers(self):
        return [(n, m) for n, m in self.named_modules() if isinstance(m, _Masked)]


class _OracleBottleneck(nn.Module):
    """models/resnet.py:60-100: 1x1 -> 3x3 (stride) -> 1x1 (x 4), BatchNorm after each, ReLU, residual add before the last ReLU."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        width = int(planes * (64 / 64.)) * 1                                   # (:68, base_width 64, groups 1)
        self.conv1 = MaskedConv(int(inplanes), width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = MaskedConv(width, width, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = MaskedConv(width, int(planes * self.expansion), 1, bias=False)
        self.bn3 = nn.BatchNorm2d(int(planes * self.expansion))
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        out += identity
        return self.relu(out)


class OracleResNet(_TaskHeads):
    """ResNet-50 of models/resnet.py:103-222 (Bottleneck, [3, 4, 6, 3]) on the masked CPU layers; attribute names equal the
    reference's (conv1, bn1, layer1.0.conv1, ..., layerN.0.downsample.0/1, classifiers.i), so state_dicts line up."""

    def __init__(self, width=1.0, depths=(3, 4, 6, 3)):
        super().__init__()
        m = width
        self.width = m
        self.inplanes = int(64 * m)                                             # (:115)
        self.conv1 = MaskedConv(3, self.inplanes, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(self.inplanes)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(m * 64, depths[0])
        self.layer2 = self._make_layer(m * 128, depths[1], stride=2)
        self.layer3 = self._make_layer(m * 256, depths[2], stride=2)
        self.layer4 = self._make_layer(m * 512, depths[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self._init_heads()
        for mod in self.modules():                                              # (:145-151)
            if isinstance(mod, MaskedConv):
                nn.init.normal_(mod.weight, 0, 0.001)
            elif isinstance(mod, nn.BatchNorm2d):
                nn.init.constant_(mod.weight, 1)
                nn.init.constant_(mod.bias, 0)

    def _make_layer(self, planes, blocks, stride=1):                            # (:170-193)
        downsample = None
        result_planes = int(planes * _OracleBottleneck.expansion)
        if stride != 1 or self.inplanes != result_planes:
            downsample = nn.Sequential(MaskedConv(self.inplanes, result_planes, 1, stride=stride, bias=False), nn.BatchNorm2d(result_planes))
        stack = [_OracleBottleneck(self.inplanes, planes, stride, downsample)]
        self.inplanes = result_planes
        for _ in range(1, blocks):
            stack.append(_OracleBottleneck(self.inplanes, planes))
        return nn.Sequential(*stack)

    def add_dataset(self, name, num_classes):                                   # (:195-201)
        if name not in self.datasets:
            self.datasets.append(name)
            head = nn.Linear(int(2048 * self.width), num_classes)
            nn.init.normal_(head.weight, 0, 0.01)
            nn.init.constant_(head.bias, 0)
            self.classifiers.append(head)

    def forward(self, x):                                                       # (:207-222)
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = self.avgpool(x)
        return self.head(x.view(x.size(0), -1))


class OracleAngleLinear(nn.Module):
    """models/spherenet.py:64-98 (m = 4): (|x| cos(theta), |x| phi(theta)), phi by the Chebyshev polynomial of cos and
    k = floor(m theta / 3.14159265)."""

    def __init__(self, in_features, out_features, m=4):
        super().__init__()
        self.m = m
        self.weight = nn.Parameter(torch.Tensor(in_features, out_features))
        self.weight.data.uniform_(-1, 1).renorm_(2, 1, 1e-5).mul_(1e5)

    def forward(self, x):
        ww = self.weight.renorm(2, 1, 1e-5).mul(1e5)
        xlen = x.pow(2).sum(1).pow(0.5)
        wlen = ww.pow(2).sum(0).pow(0.5)
        cos_theta = (x.mm(ww) / xlen.view(-1, 1) / wlen.view(1, -1)).clamp(-1, 1)
        cos_m_theta = 8 * cos_theta ** 4 - 8 * cos_theta ** 2 + 1              # mlambda[4]
        theta = cos_theta.detach().acos()
        k = (self.m * theta / 3.14159265).floor()
        phi_theta = ((k * 0.0 - 1) ** k) * cos_m_theta - 2 * k
        return cos_theta * xlen.view(-1, 1), phi_theta * xlen.view(-1, 1)


class OracleAngleLoss(nn.Module):
    """models/spherenet.py:24-61: stateful (`it` advances once per call), lambda = max(5, 1500 / (1 + 0.1 it))."""

    def __init__(self):
        super().__init__()
        self.it = 0

    def forward(self, inp, target):
        self.it += 1
        cos_theta, phi_theta = inp
        target = target.view(-1, 1)
        index = torch.zeros_like(cos_theta).scatter_(1, target, 1.0)
        lamb = max(5.0, 1500.0 / (1 + 0.1 * self.it))
        output = cos_theta * 1.0
        output = output - cos_theta * index * (1.0 + 0) / (1 + lamb)
        output = output + phi_theta * index * (1.0 + 0) / (1 + lamb)
        logpt = F.log_softmax(output, dim=1).gather(1, target).view(-1)
        return (-1 * logpt).mean()                                               # (gamma = 0)


class OracleSphereNet(_TaskHeads):
    """SphereNet-20 of models/spherenet.py:101-251: 20 biased 3x3 convs with PReLU, residual pairs, four stride-2 stages; head
    nn.Linear, or Linear(512) + AngleLinear for 'face_verification' (:176-192).  forward_to_embeddings: :136-151."""

    PLAN = [('1', 64, 1), ('2', 128, 2), ('3', 256, 4), ('4', 512, 1)]          # stage, channels, residual pairs

    def __init__(self, width=1.0):
        super().__init__()
        self.width = width
        cin = 3
        self.order = []
        for stage, ch, pairs in self.PLAN:
            c = int(ch * width)
            for i in range(1, 2 * pairs + 2):
                conv = MaskedConv(cin if i == 1 else c, c, 3, stride=2 if i == 1 else 1, padding=1, bias=True)
                setattr(self, 'conv%s_%d' % (stage, i), conv)
                setattr(self, 'relu%s_%d' % (stage, i), nn.PReLU(c))
                self.order.append((stage, i))
            cin = c
        self.flat = int(width * 512) * 7 * 7
        self._init_heads()
        for mod in self.modules():                                              # (:153-161)
            if isinstance(mod, MaskedConv):
                nn.init.kaiming_normal_(mod.weight, mode='fan_out')
                nn.init.constant_(mod.bias, 0)
            elif isinstance(mod, nn.PReLU):
                nn.init.constant_(mod.weight, 0.25)

    def add_dataset(self, name, num_classes):
        if name in self.datasets:
            return
        self.datasets.append(name)
        if 'face_verification' in name:
            head = nn.Sequential(nn.Linear(self.flat, 512), OracleAngleLinear(512, num_classes))
            nn.init.normal_(head[0].weight, 0, 0.01)
            nn.init.constant_(head[0].bias, 0)
            nn.init.normal_(head[1].weight, 0, 0.01)
        else:
            head = nn.Linear(self.flat, num_classes)
            nn.init.normal_(head.weight, 0, 0.01)
            nn.init.constant_(head.bias, 0)
        self.classifiers.append(head)

    def features(self, x):
        for stage, _, pairs in self.PLAN:
            f = lambda i, t: getattr(self, 'relu%s_%d' % (stage, i))(getattr(self, 'conv%s_%d' % (stage, i))(t))
            x = f(1, x)
            for p in range(pairs):
                x = x + f(2 * p + 3, f(2 * p + 2, x))
        return x.view(-1, self.flat)

    def forward(self, x):
        return self.head(self.features(x))

    def forward_to_embeddings(self, x):
        return self.head[0](self.features(x))


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


def train_step(model, pruner, optimizer, x, target, prune_step=None, torch_routing=False, criterion=None):
    """One iteration of utils/manager.py:50-75.  Returns (logits, loss, prune_ratio or None).  criterion: the Manager's loss
    (utils/manager.py:24-33: AngleLoss for 'face_verification', cross entropy otherwise -- the default)."""
    optimizer.zero_grad()
    out = model(x)
    loss = F.cross_entropy(out, target) if criterion is None else criterion(out, target)
    loss.backward()
    if torch_routing:
        pruner.route_torch()
    else:
        pruner.route()
    optimizer.step()
    ratio = None
    if pruner.mode == 'prune':
        ratio = pruner.gradually_prune(prune_step)
    return (out[0].detach() if isinstance(out, tuple) else out.detach()), float(loss.detach()), ratio


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


def make_task1_net(arch, dataset, num_classes, mode='finetune', width=1.0, lr=1e-2, begin=0, end=8, frequency=3, initial=0.0,
                   target=0.3, wd=4e-5, seed=1):
    """make_task1 for any of the three topologies ('vgg16' = the 224 x 224 VGG16-BN, 'resnet50', 'spherenet20'):
    (model, pruner, optimizer, criterion) as CPG_imagenet_main.py / CPG_face_main.py set a first task up."""
    torch.manual_seed(seed)
    model = {'vgg16': lambda: OracleVGG(width, 'imagenet'), 'resnet50': lambda: OracleResNet(width),
             'spherenet20': lambda: OracleSphereNet(width)}[arch]()
    model.add_dataset(dataset, num_classes)
    model.set_dataset(dataset)
    owners = {n: np.zeros(tuple(m.weight.shape), dtype=np.uint8) for n, m in model.masked_layers()}
    if mode == 'finetune':
        pruner = OraclePruner(model, owners, mode, 0, 1, begin, end, frequency, initial, target, wd, width)
        pruner.claim_free()
    else:
        for n in owners:
            owners[n][...] = 1
        pruner = OraclePruner(model, owners, mode, 1, 1, begin, end, frequency, initial, target, wd, width)
    opt = torch.optim.SGD(list(model.parameters()), lr=lr, momentum=0.9, nesterov=True, weight_decay=0.0)
    crit = OracleAngleLoss() if 'face_verification' in dataset else None
    return model, pruner, opt, crit
