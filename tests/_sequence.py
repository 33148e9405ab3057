"""Helpers shared by the sequence tests (tests/golden/sequence_{resnet50,spherenet20}.npz: configs[3] / configs[4] as multi-task
sequences, written by running the reference phase by phase -- tests/golden/make_golden.py::gen_sequence_other_nets)."""
import json
import os
import zlib

import numpy as np
import torch
import torch.nn as nn

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
INFO_KEYS = ('bias', 'bn_layer_running_mean', 'bn_layer_running_var', 'bn_layer_weight', 'bn_layer_bias', 'piggymask')


def load(arch):
    return np.load(os.path.join(GOLD, 'sequence_%s.npz' % arch))


def _crc(t):
    return zlib.crc32(t.contiguous().numpy().tobytes()) & 0xFFFFFFFF


def tasks(fx):
    return [(str(n), int(c)) for n, c in zip(fx['tasks'], fx['num_classes'])]


def batches(fx):
    """The fixture's batches, drawn again from the CPU generator the generator script used (the fixture stores their crc32 only).
    Returns {task index: (train list of (x, t), val list of (x, t))}, CPU tensors."""
    g = torch.Generator().manual_seed(int(fx['data_seed']))
    steps, B, shape = int(fx['steps']), int(fx['batch']), tuple(int(v) for v in fx['shape'])
    out = {}
    for ti, (_, ncls) in enumerate(tasks(fx)):
        xs = torch.round(torch.randn(steps, B, *shape, generator=g) * 8.0) / 8.0
        ts = torch.randint(0, ncls, (steps, B), generator=g)
        xv = torch.round(torch.randn(2, B, *shape, generator=g) * 8.0) / 8.0
        tv = torch.randint(0, ncls, (2, B), generator=g)
        assert [_crc(xs), _crc(ts), _crc(xv), _crc(tv)] == [int(v) for v in fx['data_crc/%d' % ti]], 'the generator no longer reproduces the fixture batches'
        out[ti] = ([(xs[i], ts[i]) for i in range(steps)], [(xv[i], tv[i]) for i in range(2)])
    return out


def group(fx, prefix):
    """{name: tensor} of every array stored under `prefix/`."""
    n = len(prefix) + 1
    return {k[n:]: torch.from_numpy(np.asarray(fx[k])) for k in fx.files if k.startswith(prefix + '/')}


def final_checkpoint(fx):
    """The reference's last checkpoint of the sequence as the dict its torch.save wrote (utils/manager.py:223-230)."""
    names = tasks(fx)
    keysets = json.loads(str(fx['final/info_keys']))
    shared = {}
    for dataset, keys in keysets.items():
        shared[dataset] = {}
        for key, val in keys.items():
            if isinstance(val, list):
                shared[dataset][key] = {name: torch.from_numpy(np.asarray(fx['final/info/%s/%s/%s' % (dataset, key, name)])) for name in val}
            else:
                shared[dataset][key] = float(val)
    return {'model_state_dict': group(fx, 'final/state'), 'dataset_history': [n for n, _ in names],
            'dataset2num_classes': {n: c for n, c in names}, 'masks': group(fx, 'final/mask'), 'shared_layer_info': shared}, keysets


def task1_checkpoint(fx):
    """The checkpoint task 2 starts from (task 1 after its prune run): trunk + owner masks in full; task 1's head is not stored (it is
    frozen from here on and plays no part in task 2's steps)."""
    (name, ncls) = tasks(fx)[0]
    info = {k: {} for k in INFO_KEYS}
    info['network_width_multiplier'] = float(fx['width'])
    return {'model_state_dict': group(fx, 't2start'), 'dataset_history': [name], 'dataset2num_classes': {name: ncls},
            'masks': group(fx, 't2start_mask'), 'shared_layer_info': {name: info}}


def apply_pretrained(net, arch):
    """What the generator script put in the place of the pretrained weights of task 1: ResNet-50 -- He re-draw at seed 2 in module order
    + BatchNorm running statistics from generator 29; SphereNet-20 -- nothing (the seeded initialisation as it stands).  Draws on the
    CPU and copies, so a model that already lives on the GPU gets the same numbers."""
    if arch != 'resnet50':
        return
    torch.manual_seed(2)
    for m in net.modules():
        if hasattr(m, 'piggymask') and m.weight.dim() == 4:
            w = torch.empty(m.weight.shape)
            nn.init.kaiming_normal_(w, mode='fan_out', nonlinearity='relu')
            with torch.no_grad():
                m.weight.copy_(w)
    gg = torch.Generator().manual_seed(29)
    for m in net.modules():
        if isinstance(m, nn.BatchNorm2d):
            with torch.no_grad():
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=gg) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=gg) + 0.5)


def rel_err(got, want):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    return float((got - want).abs().max()) / max(float(want.abs().max()), 1e-30)
