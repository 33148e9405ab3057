"""ResNet family of CPG on the HIP masked layers (counterpart of models/resnet.py).

All convolutions are bias-free SharableConv2d (7x7 s2 stem, 1x1 s1/s2, 3x3 s1/s2); widths scale with
network_width_multiplier with the reference's int() placement; heads are per-task nn.Linear
(`classifiers`).  Names and init order follow the reference so mask keys and seeded weights match
(tests/golden/topology.json, first_forward_resnet50.npz).
"""
import torch.nn as nn

from . import layers as nl
from .fused_bn import FusedSequential, conv_bn_act, conv_bn_act_pool, conv_bn_act_skip, conv_bn_add_act

__all__ = ['ResNet', 'BasicBlock', 'Bottleneck', 'resnet18', 'resnet34', 'resnet50', 'resnet101', 'resnet152',
           'resnext50_32x4d', 'resnext101_32x8d']


def conv3x3(in_planes, out_planes, stride=1, groups=1, dilation=1):
    return nl.SharableConv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=dilation,
                             groups=groups, bias=False, dilation=dilation)


def conv1x1(in_planes, out_planes, stride=1):
    return nl.SharableConv2d(in_planes, out_planes, kernel_size=1, stride=stride, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1,
                 norm_layer=None):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        if groups != 1 or base_width != 64:
            raise ValueError('BasicBlock only supports groups=1 and base_width=64')
        if dilation > 1:
            raise NotImplementedError('Dilation > 1 not supported in BasicBlock')
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = norm_layer(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = norm_layer(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        out = conv_bn_act(self.conv1, self.bn1, self.relu, x)
        return conv_bn_add_act(self.conv2, self.bn2, self.relu, out, x if self.downsample is None else self.downsample(x))


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1,
                 norm_layer=None):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        width = int(planes * (base_width / 64.)) * groups
        self.conv1 = conv1x1(int(inplanes), width)
        self.bn1 = norm_layer(width)
        self.conv2 = conv3x3(width, width, stride, groups, dilation)
        self.bn2 = norm_layer(width)
        self.conv3 = conv1x1(width, int(planes * self.expansion))
        self.bn3 = norm_layer(int(planes * self.expansion))
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        # the block input feeds conv1 and the second branch (identity or downsample conv): routed through conv1's autograd node, which
        # then adds the second branch's gradient in its own input-gradient epilogue (no separate add pass)
        out, skip = conv_bn_act_skip(self.conv1, self.bn1, self.relu, x)
        identity = skip if self.downsample is None else self.downsample(skip)
        out = conv_bn_act(self.conv2, self.bn2, self.relu, out)
        return conv_bn_add_act(self.conv3, self.bn3, self.relu, out, identity)


class ResNet(nn.Module):
    def __init__(self, block, layers, dataset_history, dataset2num_classes, network_width_multiplier,
                 shared_layer_info, num_classes=1000, zero_init_residual=False, groups=1, width_per_group=64,
                 replace_stride_with_dilation=None, norm_layer=None):
        super().__init__()
        self._norm_layer = norm_layer or nn.BatchNorm2d
        m = network_width_multiplier
        self.network_width_multiplier = m
        self.shared_layer_info = shared_layer_info
        self.inplanes = int(64 * m)
        self.dilation = 1
        rswd = replace_stride_with_dilation or [False, False, False]
        if len(rswd) != 3:
            raise ValueError('replace_stride_with_dilation should be None or a 3-element tuple, got {}'.format(rswd))
        self.groups = groups
        self.base_width = width_per_group
        self.conv1 = nl.SharableConv2d(3, self.inplanes, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = self._norm_layer(self.inplanes)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, m * 64, layers[0])
        self.layer2 = self._make_layer(block, m * 128, layers[1], stride=2, dilate=rswd[0])
        self.layer3 = self._make_layer(block, m * 256, layers[2], stride=2, dilate=rswd[1])
        self.layer4 = self._make_layer(block, m * 512, layers[3], stride=2, dilate=rswd[2])
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.datasets, self.classifiers = dataset_history, nn.ModuleList()
        self.dataset2num_classes = dataset2num_classes
        if self.datasets:
            self._reconstruct_classifiers()
        # models/resnet.py:145-151: conv N(0, 1e-3), norm layers 1/0
        for mod in self.modules():
            if isinstance(mod, nl.SharableConv2d):
                nn.init.normal_(mod.weight, 0, 0.001)
            elif isinstance(mod, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(mod.weight, 1)
                nn.init.constant_(mod.bias, 0)
        if zero_init_residual:
            for mod in self.modules():
                if isinstance(mod, Bottleneck):
                    nn.init.constant_(mod.bn3.weight, 0)
                elif isinstance(mod, BasicBlock):
                    nn.init.constant_(mod.bn2.weight, 0)

    def _reconstruct_classifiers(self):
        for dataset, num_classes in self.dataset2num_classes.items():
            width = self.shared_layer_info[dataset]['network_width_multiplier']
            self.classifiers.append(nn.Linear(int(width * 2048), num_classes))

    def _make_layer(self, block, planes, blocks, stride=1, dilate=False):
        norm_layer = self._norm_layer
        downsample = None
        result_planes = int(planes * block.expansion)
        previous_dilation = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        if stride != 1 or self.inplanes != result_planes:
            downsample = FusedSequential(conv1x1(self.inplanes, result_planes, stride), norm_layer(result_planes))
        stack = [block(self.inplanes, planes, stride, downsample, self.groups, self.base_width, previous_dilation,
                       norm_layer)]
        self.inplanes = result_planes
        for _ in range(1, blocks):
            stack.append(block(self.inplanes, planes, groups=self.groups, base_width=self.base_width,
                               dilation=self.dilation, norm_layer=norm_layer))
        return nn.Sequential(*stack)

    def add_dataset(self, dataset, num_classes):
        if dataset in self.datasets:
            return
        self.datasets.append(dataset)
        self.dataset2num_classes[dataset] = num_classes
        head = nn.Linear(int(2048 * self.network_width_multiplier), num_classes)
        self.classifiers.append(head)
        nn.init.normal_(head.weight, 0, 0.01)
        nn.init.constant_(head.bias, 0)

    def set_dataset(self, dataset):
        assert dataset in self.datasets
        self.classifier = self.classifiers[self.datasets.index(dataset)]

    def forward(self, x):
        x = conv_bn_act_pool(self.conv1, self.bn1, self.relu, self.maxpool, x)
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        x = self.avgpool(x)
        return self.classifier(x.view(x.size(0), -1))


def _factory(block, depths, **fixed):
    def build(dataset_history=[], dataset2num_classes={}, **kwargs):
        kwargs.update(fixed)
        return ResNet(block, depths, dataset_history, dataset2num_classes, **kwargs)
    return build


resnet18 = _factory(BasicBlock, [2, 2, 2, 2])
resnet34 = _factory(BasicBlock, [3, 4, 6, 3])
resnet50 = _factory(Bottleneck, [3, 4, 6, 3])
resnet101 = _factory(Bottleneck, [3, 4, 23, 3])
resnet152 = _factory(Bottleneck, [3, 8, 36, 3])
resnext50_32x4d = _factory(Bottleneck, [3, 4, 6, 3], groups=4, width_per_group=32)
resnext101_32x8d = _factory(Bottleneck, [3, 4, 23, 3], groups=8, width_per_group=32)
