"""VGG16-BN topologies of CPG on the HIP masked layers (counterpart of models/vgg.py:33-154,276-282).

Module order, names (`features.{0,3,7,...}`, `classifiers.{i}`), parameter shapes and the
initialisation sequence equal the reference's, so owner-mask dictionary keys, state_dict keys and
`torch.manual_seed(s)` initial weights line up with it (pinned by tests/golden/topology.json and
first_forward_vgg*.npz).  `features` is an nn.Sequential subclass (FusedSequential) that registers the very
same modules but evaluates each BatchNorm2d -> ReLU pair with the fused HIP kernels.  The torchvision-style vgg11..vgg19 factories of the reference are broken
upstream (they omit a required argument) and unused; they are not provided.
"""
import torch.nn as nn

from . import layers as nl
from .fused_bn import FusedSequential

__all__ = ['VGG', 'View', 'custom_vgg', 'custom_vgg_cifar100', 'make_layers', 'make_layers_cifar100']


class View(nn.Module):
    """Reshape as a module so it can sit inside nn.Sequential (models/vgg.py:23-31)."""

    def __init__(self, *shape):
        super().__init__()
        self.shape = shape

    def forward(self, input):
        return input.view(*self.shape)


def _conv_stack(cfg, mult, batch_norm, groups):
    mods, cin = [], 3
    for v in cfg:
        if v == 'M':
            mods.append(nn.MaxPool2d(kernel_size=2, stride=2))
            continue
        cout = int(v * mult)
        # the first conv never uses groups (models/vgg.py:103-106)
        conv = nl.SharableConv2d(cin, cout, kernel_size=3, padding=1, bias=False,
                                 **({} if cin == 3 else {'groups': groups}))
        mods += [conv, nn.BatchNorm2d(cout), nn.ReLU(inplace=True)] if batch_norm else [conv, nn.ReLU(inplace=True)]
        cin = cout
    return mods


def make_layers_cifar100(cfg, network_width_multiplier, batch_norm=False, groups=1):
    """32x32 input: 13 convs, flatten to 512*m, two masked FC layers (models/vgg.py:95-122)."""
    m = network_width_multiplier
    mods = _conv_stack(cfg, m, batch_norm, groups)
    mods += [View(-1, int(512 * m)),
             nl.SharableLinear(int(512 * m), int(4096 * m)), nn.ReLU(True),
             nl.SharableLinear(int(4096 * m), int(4096 * m)), nn.ReLU(True)]
    return FusedSequential(*mods)


def make_layers(cfg, network_width_multiplier, batch_norm=False, groups=1):
    """224x224 input: flatten to 512*m*7*7, Dropout after each FC (models/vgg.py:124-154)."""
    m = network_width_multiplier
    mods = _conv_stack(cfg, m, batch_norm, groups)
    mods += [View(-1, int(512 * m) * 7 * 7),
             nl.SharableLinear(int(512 * m) * 7 * 7, int(4096 * m)), nn.ReLU(True), nn.Dropout(),
             nl.SharableLinear(int(4096 * m), int(4096 * m)), nn.ReLU(True), nn.Dropout()]
    return FusedSequential(*mods)


class VGG(nn.Module):
    """Shared feature trunk + one nn.Linear head per task (models/vgg.py:33-93)."""

    def __init__(self, features, dataset_history, dataset2num_classes, network_width_multiplier=1.0,
                 shared_layer_info={}, init_weights=True, progressive_init=False):
        super().__init__()
        self.features = features
        self.network_width_multiplier = network_width_multiplier
        self.shared_layer_info = shared_layer_info
        self.datasets, self.classifiers = dataset_history, nn.ModuleList()
        self.dataset2num_classes = dataset2num_classes
        if self.datasets:
            self._reconstruct_classifiers()
        if init_weights:
            self._initialize_weights()
        if progressive_init:
            for m in self.modules():
                if isinstance(m, nl.SharableConv2d):
                    nn.init.normal_(m.weight, 0, 0.01)

    def forward(self, x):
        f = self.features(x)
        n_in = getattr(self.classifier, 'in_features', None)
        if n_in is not None and f.dim() == 2 and f.shape[1] > n_in:
            # an earlier (narrower) task served from the grown network without cropping it: its head reads the first
            # 4096 * (its width) features; the rest belong to channels that apply_mask zeroed for this task
            f = f[:, :n_in]
        return self.classifier(f)

    def _initialize_weights(self):
        # same traversal order and distributions as models/vgg.py:59-70 (RNG parity)
        for m in self.modules():
            if isinstance(m, nl.SharableConv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, nl.SharableLinear):
                nn.init.normal_(m.weight, 0, 0.01)
                nn.init.constant_(m.bias, 0)

    def _reconstruct_classifiers(self):
        for dataset, num_classes in self.dataset2num_classes.items():
            width = self.shared_layer_info[dataset]['network_width_multiplier']
            self.classifiers.append(nn.Linear(int(width * 4096), num_classes))

    def add_dataset(self, dataset, num_classes):
        """Append a head for a new task (models/vgg.py:81-88)."""
        if dataset in self.datasets:
            return
        self.datasets.append(dataset)
        self.dataset2num_classes[dataset] = num_classes
        head = nn.Linear(int(4096 * self.network_width_multiplier), num_classes)
        self.classifiers.append(head)
        nn.init.normal_(head.weight, 0, 0.01)
        nn.init.constant_(head.bias, 0)

    def set_dataset(self, dataset):
        """Select the active head (models/vgg.py:90-93)."""
        assert dataset in self.datasets
        self.classifier = self.classifiers[self.datasets.index(dataset)]


def custom_vgg_cifar100(custom_cfg, dataset_history=[], dataset2num_classes={}, network_width_multiplier=1.0,
                        groups=1, shared_layer_info={}, **kwargs):
    return VGG(make_layers_cifar100(custom_cfg, network_width_multiplier, batch_norm=True, groups=groups),
               dataset_history, dataset2num_classes, network_width_multiplier, shared_layer_info, **kwargs)


def custom_vgg(custom_cfg, dataset_history=[], dataset2num_classes={}, network_width_multiplier=1.0,
               groups=1, shared_layer_info={}, **kwargs):
    return VGG(make_layers(custom_cfg, network_width_multiplier, batch_norm=True, groups=groups),
               dataset_history, dataset2num_classes, network_width_multiplier, shared_layer_info, **kwargs)
