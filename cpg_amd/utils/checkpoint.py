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
from torch.nn.parameter import Parameter

from ..models import layers as nl

_TASK_KEYS = ('bias', 'bn_layer_running_mean', 'bn_layer_running_var', 'bn_layer_weight', 'bn_layer_bias',
              'prelu_layer_weight', 'piggymask')
# keys of the active-head alias (`self.classifier = self.classifiers[i]`) and of the A-Softmax head pair
_HEAD_ALIAS = ('classifier.weight', 'classifier.bias', 'classifier.0.weight', 'classifier.0.bias', 'classifier.1.weight')


def _root(model):
    return model.module if hasattr(model, 'module') else model


def _masked(m):
    return isinstance(m, (nl.SharableConv2d, nl.SharableLinear))


def _snap(t):
    """A detached copy that keeps Parameter-ness (the reference re-attaches these with `module.bias = ...`, which only
    accepts Parameters for registered parameter names)."""
    c = t.detach().clone()
    return Parameter(c, requires_grad=t.requires_grad) if isinstance(t, Parameter) else c


def collect_task_layers(model, shared_layer_info, dataset):
    """Refresh shared_layer_info[dataset] from the live modules (utils/manager.py:202-221).

    The reference stores the live Parameter / buffer objects; that is a snapshot only because each of its phases is a
    separate process with a torch.save in between.  In one process (cpg_amd.driver.CPGSession) the next task keeps
    training the very same BatchNorm / bias / PReLU tensors, so COPIES are stored: `shared_layer_info[task]` is what the
    task looked like when it was collected, whatever trains afterwards (CPG's no-forgetting property depends on it)."""
    info = shared_layer_info.setdefault(dataset, {})
    has_prelu = any(isinstance(m, nn.PReLU) for m in _root(model).modules())
    for k in _TASK_KEYS:
        if k != 'prelu_layer_weight' or has_prelu:      # (only CPG_face_main.py:253-262 creates that key: the file keeps the reference's key set)
            info.setdefault(k, {})
    info['piggymask'] = {}                       # a task whose piggymasks were dropped must not keep stale ones
    for name, module in _root(model).named_modules():
        if _masked(module):
            if module.bias is not None:
                info['bias'][name] = _snap(module.bias)
            if module.piggymask is not None:
                info['piggymask'][name] = _snap(module.piggymask)
        elif isinstance(module, nn.BatchNorm2d):
            info['bn_layer_running_mean'][name] = _snap(module.running_mean)
            info['bn_layer_running_var'][name] = _snap(module.running_var)
            info['bn_layer_weight'][name] = _snap(module.weight)
            info['bn_layer_bias'][name] = _snap(module.bias)
        elif isinstance(module, nn.PReLU):
            info['prelu_layer_weight'][name] = _snap(module.weight)
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


def _put(dst, src):
    """Copy a stored per-task tensor into the live one (top-left corner / crop when the widths differ)."""
    with torch.no_grad():
        if dst.shape == src.shape:
            dst.copy_(src.to(dst.device))
        else:
            _corner(dst, src.to(dst.device), grow=all(a >= b for a, b in zip(dst.shape, src.shape)))


def attach_task_layers(model, shared_layer_info, dataset, piggymasks=False):
    """Give the model the task's own bias / BatchNorm / PReLU values (utils/manager.py:301-319).  The reference swaps
    the stored tensor OBJECTS into the modules; here the values are copied into the live tensors, so optimizers and
    gradient hooks that hold the live Parameters stay valid and the stored snapshot cannot be trained by accident.
    piggymasks=True also restores the task's piggymasks (CPG_cifar100_main_normal.py:281-290): a module gets a
    Parameter holding the stored values, or None when the task has none (task 1)."""
    info = shared_layer_info[dataset]
    for name, module in _root(model).named_modules():
        if _masked(module):
            if module.bias is not None and name in info.get('bias', {}):
                _put(module.bias, info['bias'][name])
            if piggymasks:
                pm = info.get('piggymask', {}).get(name)
                if pm is None:
                    module.piggymask = None
                else:
                    new = torch.full_like(module.weight.detach(), 0.0)
                    _put(new, pm.detach())
                    module.piggymask = Parameter(new)
        elif isinstance(module, nn.BatchNorm2d) and name in info.get('bn_layer_weight', {}):
            _put(module.running_mean, info['bn_layer_running_mean'][name])
            _put(module.running_var, info['bn_layer_running_var'][name])
            _put(module.weight, info['bn_layer_weight'][name])
            _put(module.bias, info['bn_layer_bias'][name])
        elif isinstance(module, nn.PReLU) and name in info.get('prelu_layer_weight', {}):
            _put(module.weight, info['prelu_layer_weight'][name])


def resize_masks(model, masks, mode):
    """Owner masks after the network width changed (CPG_cifar100_main_normal.py:208-249).
    mode 'finetune': the net was widened -- every mask becomes a zero (= free) tensor of the new weight shape with the
    old mask in its top-left corner; mode 'inference': the net is narrower than the checkpoint (an older task evaluated
    at its own width) -- masks are cropped.  Any other mode with mismatching shapes is an error, as the reference's
    asserts say.  Masks are replaced in the dict in place (the dict object is shared with the pruner)."""
    need = False
    for name, module in model.named_modules():
        if _masked(module) and tuple(masks[name].shape) != tuple(module.weight.shape):
            need = True
            wider = all(a <= b for a, b in zip(masks[name].shape, module.weight.shape))
            narrower = all(a >= b for a, b in zip(masks[name].shape, module.weight.shape))
            if not ((wider and mode == 'finetune') or (narrower and mode == 'inference')):
                raise AssertionError('mask %s has shape %s, weight %s: masks only grow in finetune mode and only shrink in '
                                     'inference mode' % (name, tuple(masks[name].shape), tuple(module.weight.shape)))
    if not need:
        return False
    for name, module in model.named_modules():
        if _masked(module):
            old = masks[name]
            new = torch.zeros(module.weight.shape, dtype=torch.uint8, device=module.weight.device)
            if mode == 'finetune':
                new[tuple(slice(0, s) for s in old.shape)].copy_(old.to(new.device))
            else:
                new.copy_(old[tuple(slice(0, s) for s in new.shape)].to(new.device))
            masks[name] = new
    return True
