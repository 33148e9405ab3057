"""SphereNet-20 + A-Softmax head of CPG on the HIP masked layers (counterpart of models/spherenet.py).

20 x SharableConv2d(3x3, bias=True) with PReLU residual units (four stride-2 stages); the small
dense/elementwise AngleLinear / AngleLoss ops stay in stock torch.
Input is 112x112 (flatten expects 512*m*7*7).
"""

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.parameter import Parameter

from . import layers as nl
from .fused_bn import conv_prelu, conv_prelu_skip
from .vgg import View

__all__ = ['SphereNet', 'spherenet20', 'AngleLoss', 'AngleLinear']


class AngleLoss(nn.Module):
    """A-Softmax loss with iteration-annealed lambda (models/spherenet.py:24-61).  Stateful: `it`
    advances once per call."""

    def __init__(self, gamma=0):
        super().__init__()
        self.gamma = gamma
        self.it = 0
        self.LambdaMin, self.LambdaMax, self.lamb = 5.0, 1500.0, 1500.0

    def forward(self, input, target):
        self.it += 1
        cos_theta, phi_theta = input
        target = target.view(-1, 1)
        onehot = torch.zeros_like(cos_theta).scatter_(1, target, 1.0)
        self.lamb = max(self.LambdaMin, self.LambdaMax / (1 + 0.1 * self.it))
        scale = (1.0 + 0) / (1 + self.lamb)
        output = cos_theta * 1.0
        output = output - cos_theta * onehot * scale
        output = output + phi_theta * onehot * scale
        logpt = F.log_softmax(output, dim=1).gather(1, target).view(-1)
        pt = logpt.detach().exp()
        return (-1 * (1 - pt) ** self.gamma * logpt).mean()


class AngleLinear(nn.Module):
    """Angular-margin head, m = 4 (models/spherenet.py:64-98): returns (|x| cos(theta), |x| phi(theta))."""

    _CHEBYSHEV = [lambda x: x ** 0, lambda x: x ** 1, lambda x: 2 * x ** 2 - 1, lambda x: 4 * x ** 3 - 3 * x,
                  lambda x: 8 * x ** 4 - 8 * x ** 2 + 1, lambda x: 16 * x ** 5 - 20 * x ** 3 + 5 * x]

    def __init__(self, in_features, out_features, m=4):
        super().__init__()
        self.in_features, self.out_features, self.m = in_features, out_features, m
        self.weight = Parameter(torch.Tensor(in_features, out_features))
        self.weight.data.uniform_(-1, 1).renorm_(2, 1, 1e-5).mul_(1e5)

    def forward(self, input):
        ww = self.weight.renorm(2, 1, 1e-5).mul(1e5)
        xlen = input.pow(2).sum(1).pow(0.5)
        wlen = ww.pow(2).sum(0).pow(0.5)
        cos_theta = (input.mm(ww) / xlen.view(-1, 1) / wlen.view(1, -1)).clamp(-1, 1)
        cos_m_theta = self._CHEBYSHEV[self.m](cos_theta)
        theta = cos_theta.detach().acos()
        k = (self.m * theta / 3.14159265).floor()
        phi_theta = ((k * 0.0 - 1) ** k) * cos_m_theta - 2 * k
        return cos_theta * xlen.view(-1, 1), phi_theta * xlen.view(-1, 1)


# (stage, units): stage s starts with a stride-2 conv s_1, followed by `units` residual pairs
_STAGES = [(1, 64, 1), (2, 128, 2), (3, 256, 4), (4, 512, 1)]


class SphereNet(nn.Module):
    def __init__(self, dataset_history, dataset2num_classes, network_width_multiplier=1.0, shared_layer_info={},
                 init_weights=True):
        super().__init__()
        self.network_width_multiplier = network_width_multiplier
        self.make_feature_layers()
        self.shared_layer_info = shared_layer_info
        self.datasets = dataset_history
        self.classifiers = nn.ModuleList()
        self.dataset2num_classes = dataset2num_classes
        if self.datasets:
            self._reconstruct_classifiers()
        if init_weights:
            self._initialize_weights()

    def make_feature_layers(self):
        """conv{s}_{i} / relu{s}_{i} in the reference's registration order (models/spherenet.py:201-251)."""
        ext = self.network_width_multiplier
        cin = 3
        for stage, base, units in _STAGES:
            c = int(base * ext)
            setattr(self, 'conv%d_1' % stage, nl.SharableConv2d(cin, c, 3, 2, 1))
            setattr(self, 'relu%d_1' % stage, nn.PReLU(c))
            for i in range(2, 2 * units + 2):
                setattr(self, 'conv%d_%d' % (stage, i), nl.SharableConv2d(c, c, 3, 1, 1))
                setattr(self, 'relu%d_%d' % (stage, i), nn.PReLU(c))
            cin = c
        self.flatten = View(-1, int(ext * 512) * 7 * 7)

    def _trunk(self, x):
        for stage, _, units in _STAGES:
            x = conv_prelu(getattr(self, 'conv%d_1' % stage), getattr(self, 'relu%d_1' % stage), x)
            for u in range(units):
                a, b = 2 * u + 2, 2 * u + 3
                # x feeds conv a and the unit's sum: both gradients of x meet in conv a's input-gradient epilogue (conv_prelu_skip)
                y, skip = conv_prelu_skip(getattr(self, 'conv%d_%d' % (stage, a)), getattr(self, 'relu%d_%d' % (stage, a)), x)
                x = conv_prelu(getattr(self, 'conv%d_%d' % (stage, b)), getattr(self, 'relu%d_%d' % (stage, b)), y, res=skip)
        return self.flatten(x)

    def forward(self, x):
        return self.classifier(self._trunk(x))

    def forward_to_embeddings(self, x):
        return self.classifier[0](self._trunk(x))

    def _initialize_weights(self):
        for m in self.modules():
            if isinstance(m, nl.SharableConv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out')
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.PReLU):
                nn.init.constant_(m.weight, 0.25)

    def _head(self, width, dataset, num_classes):
        flat = int(width * 512) * 7 * 7
        if 'face_verification' in dataset:
            return nn.Sequential(nl.HeadLinear(flat, 512), AngleLinear(512, num_classes))
        return nl.HeadLinear(flat, num_classes)

    def _reconstruct_classifiers(self):
        for dataset, num_classes in self.dataset2num_classes.items():
            self.classifiers.append(self._head(self.shared_layer_info[dataset]['network_width_multiplier'], dataset, num_classes))

    def add_dataset(self, dataset, num_classes):
        if dataset in self.datasets:
            return
        self.datasets.append(dataset)
        self.dataset2num_classes[dataset] = num_classes
        head = self._head(self.network_width_multiplier, dataset, num_classes)
        self.classifiers.append(head)
        if isinstance(head, nn.Sequential):
            nn.init.normal_(head[0].weight, 0, 0.01)
            nn.init.constant_(head[0].bias, 0)
            nn.init.normal_(head[1].weight, 0, 0.01)
        else:
            nn.init.normal_(head.weight, 0, 0.01)
            nn.init.constant_(head.bias, 0)

    def set_dataset(self, dataset):
        assert dataset in self.datasets
        self.classifier = self.classifiers[self.datasets.index(dataset)]


def spherenet20(dataset_history=[], dataset2num_classes={}, network_width_multiplier=1.0, shared_layer_info={}, **kwargs):
    return SphereNet(dataset_history, dataset2num_classes, network_width_multiplier, shared_layer_info, **kwargs)
