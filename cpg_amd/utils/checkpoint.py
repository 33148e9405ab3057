"""Checkpoint / `shared_layer_info` wire format of CPG.

Same on-disk format as the reference's Manager.save_checkpoint / load_checkpoint /
load_checkpoint_only_for_evaluate (utils/manager.py:198-320): one `torch.save`d dict

    {'model_state_dict', 'dataset_history', 'dataset2num_classes', 'masks', 'shared_layer_info'}

where `shared_layer_info[dataset]` stashes the per-task layers (conv/FC bias, BatchNorm statistics and affine,
PReLU slopes, piggymasks) keyed by module name, so checkpoints written by either implementation load in the other.
Host-side file I/O only; tensors are copied with torch ops on whatever device they live on.
"""
import torch
import torch.nn as nn

from ..models import layers as nl

_TASK_KEYS = ('bias', 'bn_layer_running_mean', 'bn_layer_running_var', 'bn_layer_weight', 'bn_layer_bias',
              'prelu_layer_weight', 'piggymask')
# keys of the active-head alias (`self.classifier = self.classifiers[i]`) and of the A-Softmax head pair
_HEAD_ALIAS = ('classifier.weight', 'classifier.bias', 'classifier.0.weight', 'classifier.0.bias', 'classifier.1.weight')


def _root(model):
    return model.module if hasattr(model, 'module') else model


def _masked(m):
    return isinstance(m, (nl.SharableConv2d, nl.SharableLinear))


def collect_task_layers(model, shared_layer_info, dataset):
    """Refresh shared_layer_info[dataset] from the live modules (utils/manager.py:202-221)."""
    info = shared_layer_info.setdefault(dataset, {})
    for k in _TASK_KEYS:
        info.setdefault(k, {})
    for name, module in _root(model).named_modules():
        if _masked(module):
            if module.bias is not None:
                info['bias'][name] = module.bias
            if module.piggymask is not None:
                info['piggymask'][name] = module.piggymask
        elif isinstance(module, nn.BatchNorm2d):
            info['bn_layer_running_mean'][name] = module.running_mean
            info['bn_layer_running_var'][name] = module.running_var
            info['bn_layer_weight'][name] = module.weight
            info['bn_layer_bias'][name] = module.bias
        elif isinstance(module, nn.PReLU):
            info['prelu_layer_weight'][name] = module.weight
    return info


def save_checkpoint(model, masks, shared_layer_info, dataset, filepath):
    collect_task_layers(model, shared_layer_info, dataset)
    root = _root(model)
    torch.save({'model_state_dict': root.state_dict(), 'dataset_history': root.datasets,
                'dataset2num_classes': root.dataset2num_classes, 'masks': masks,
                'shared_layer_info': shared_layer_info}, filepath)


def _corner(dst, src, grow):
    """grow=True: copy src into the top-left corner of a (possibly wider) dst; False: crop a (possibly wider) src."""
    if dst.dim() == 0 or src.dim() != dst.dim():
        dst.copy_(src)
        return
    if grow:
        dst[tuple(slice(0, s) for s in src.shape)].copy_(src)
    else:
        dst.copy_(src[tuple(slice(0, s) for s in dst.shape)])


def load_state(model, state_dict, for_evaluate):
    """utils/manager.py:233-264 (resume training: narrower checkpoint into a possibly widened net) and :266-300
    (inference: possibly wider checkpoint cropped to the task's width).  Piggymasks and the active-head alias are
    restored by the driver, as in the reference."""
    cur = _root(model).state_dict()
    with torch.no_grad():
        for name, param in state_dict.items():
            if 'piggymask' in name or name in _HEAD_ALIAS:
                continue
            if name not in cur:
                raise KeyError('checkpoint tensor %r has no counterpart in the model' % name)
            dst = cur[name]
            if for_evaluate:
                if dst.dim() == 4 or (dst.dim() == 2 and 'features' in name) or dst.dim() == 1:
                    _corner(dst, param, grow=False)
                else:
                    dst.copy_(param)
            else:
                if dst.dim() == 4 or (dst.dim() == 2 and ('features' in name or 'classifiers' in name)) or dst.dim() == 1:
                    _corner(dst, param, grow=True)
                else:
                    dst.copy_(param)


def attach_task_layers(model, shared_layer_info, dataset):
    """Re-attach the task's own bias / BatchNorm / PReLU tensors (utils/manager.py:301-319)."""
    info = shared_layer_info[dataset]
    for name, module in _root(model).named_modules():
        if _masked(module):
            if module.bias is not None:
                module.bias = info['bias'][name]
        elif isinstance(module, nn.BatchNorm2d):
            module.running_mean = info['bn_layer_running_mean'][name]
            module.running_var = info['bn_layer_running_var'][name]
            module.weight = info['bn_layer_weight'][name]
            module.bias = info['bn_layer_bias'][name]
        elif isinstance(module, nn.PReLU):
            module.weight = info['prelu_layer_weight'][name]
