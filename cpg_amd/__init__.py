"""cpg_amd -- MI355X (gfx950) implementation of ivclab/CPG's masked-CNN train/prune/retrain hot path.

The package mirrors the reference's module tree for that path so callers switch import roots only:
    models/layers.py   -> cpg_amd.models.layers   (Binarizer, SharableConv2d, SharableLinear)
    models/vgg.py ...  -> cpg_amd.models          (custom_vgg, custom_vgg_cifar100, resnet50, spherenet20)
    utils/prune.py     -> cpg_amd.utils.prune     (SparsePruner)
    utils/manager.py   -> cpg_amd.utils.manager   (Manager)
    utils/__init__.py  -> cpg_amd.utils           (Optimizers, Metric, classification_accuracy)
    nn.DataParallel    -> cpg_amd.dist.DataParallel (one process per GPU, RCCL all-reduce)
All device arithmetic of those classes runs in libcpg_hip.so (cpg_amd/csrc, C ABI in include/cpg_hip.h).
"""
__version__ = '0.1.0'
