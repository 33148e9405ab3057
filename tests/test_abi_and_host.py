"""CPU-side checks: the C-ABI library loads and exports every symbol include/cpg_hip.h declares,
host-side logic (schedule, gate, topology, init parity) matches the golden fixtures, and the
product path refuses to run without a HIP device (no CPU fallback)."""
import json
import os
import re
import types

import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import GOLDEN, ROOT, load_golden

import cpg_amd._lib as L
import cpg_amd.models as M
from cpg_amd.models import layers as nl
from cpg_amd.utils.prune import SparsePruner

VGG_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, 'include', 'cpg_hip.h')).read()
    declared = set(re.findall(r'\b(cpg_[a-z0-9_]+)\s*\(', header))
    declared -= {'cpg_conv_desc', 'cpg_prune_result'}
    assert declared == set(L.EXPORTS), declared ^ set(L.EXPORTS)
    lib = L.lib()                      # raises if the .so is missing or a symbol is absent
    assert lib.cpg_version() == 1
    assert lib.cpg_rank_prune_workspace_bytes() > 0


def test_struct_layouts_match_header():
    import ctypes
    assert ctypes.sizeof(L.ConvDesc) == 14 * 4
    assert ctypes.sizeof(L.PruneResult) == 32
    assert L.PruneResult.cutoff.offset == 24 and L.PruneResult.status.offset == 28


def test_no_cpu_fallback():
    conv = nl.SharableConv2d(3, 4, 3, padding=1, bias=False)
    nn.init.normal_(conv.weight)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        conv(torch.zeros(1, 3, 8, 8))
    lin = nl.SharableLinear(4, 3)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        lin(torch.zeros(2, 4))
    with pytest.raises(NotImplementedError):
        nl.SharableLinear(4, 3, threshold_fn='ternarizer')


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'cpg_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, re.M), os.path.join(dirpath, f)


def build(arch, width, ncls=5):
    torch.manual_seed(1)
    kw = dict(dataset_history=[], dataset2num_classes={}, network_width_multiplier=width, shared_layer_info={})
    m = {'vgg_cifar100': lambda: M.custom_vgg_cifar100(VGG_CFG, **kw), 'vgg': lambda: M.custom_vgg(VGG_CFG, **kw),
         'resnet50': lambda: M.resnet50(**kw), 'spherenet20': lambda: M.spherenet20(**kw)}[arch]()
    m.add_dataset('t1', ncls)
    m.set_dataset('t1')
    return m


class _Wrap(nn.Module):
    def __init__(self, m):
        super().__init__()
        self.module = m


@pytest.mark.parametrize('arch', ['vgg_cifar100', 'vgg', 'resnet50', 'spherenet20'])
def test_topology_names_and_shapes_match_reference(arch):
    topo = json.load(open(os.path.join(GOLDEN, 'topology.json')))[arch]
    m = _Wrap(build(arch, 1.0))
    layers = []
    for name, mod in m.named_modules():
        if isinstance(mod, nl.SharableConv2d):
            layers.append([name, 'conv', list(mod.weight.shape), list(mod.stride), list(mod.padding), mod.bias is not None])
        elif isinstance(mod, nl.SharableLinear):
            layers.append([name, 'linear', list(mod.weight.shape), None, None, mod.bias is not None])
    assert layers == topo['masked_layers']
    assert [[n, list(p.shape)] for n, p in m.named_parameters()] == topo['param_names']
    assert sum(p.numel() for p in m.parameters()) == topo['n_params']


@pytest.mark.parametrize('arch,width,fx', [('vgg_cifar100', 0.125, 'first_forward_vgg_cifar100'), ('vgg', 0.125, 'first_forward_vgg'),
                                           ('resnet50', 0.25, 'first_forward_resnet50'), ('spherenet20', 0.25, 'first_forward_spherenet20')])
def test_seeded_init_matches_reference(arch, width, fx):
    g = load_golden(fx)
    m = build(arch, width, int(g['num_classes']))
    dig = np.array([[float(p.detach().double().sum()), float(p.detach().double().abs().sum())] for p in m.parameters()])
    np.testing.assert_allclose(dig, g['param_digest'], rtol=1e-12, atol=1e-12)
    np.testing.assert_array_equal(next(iter(m.parameters())).detach().reshape(-1)[:64].numpy(), g['first_param_head'])


def test_piggymask_assignment_registers_parameter():
    conv = nl.SharableConv2d(3, 4, 3, padding=1, bias=False)
    assert conv.piggymask is None and 'piggymask' not in dict(conv.named_parameters())
    conv.piggymask = nn.Parameter(torch.full(conv.weight.shape, 0.01))
    assert 'piggymask' in dict(conv.named_parameters()) and 'piggymask' in conv.state_dict()
    assert conv.info['threshold'] == 5e-3


def _pruner(mode, begin, end, freq, init, target):
    net = build('vgg_cifar100', 0.125)
    model = _Wrap(net)
    args = types.SimpleNamespace(mode=mode, dataset='t1', finetune_again=False, target_sparsity=target, initial_sparsity=init,
                                 pruning_frequency=freq, weight_decay=4e-5, network_width_multiplier=0.125)
    return SparsePruner(model, {}, args, begin, end, 1)


def test_schedule_and_gate_match_reference():
    tab = load_golden('schedule')['table']
    pruners = {}
    for begin, end, freq, init, target, step, upd, ratio in tab:
        key = (begin, end, freq, init, target)
        if key not in pruners:
            pruners[key] = _pruner('prune', int(begin), int(end), int(freq), init, target)
        p = pruners[key]
        got = p._time_to_update_masks(int(step))
        assert int(got) == int(upd)
        if got:
            p.last_prune_step = int(step)
        assert p._adjust_sparsity(int(step)) == ratio


def test_current_dataset_idx_rules():
    assert _pruner('prune', 0, 8, 3, 0.0, 0.1).current_dataset_idx == 1
    assert _pruner('finetune', 0, 8, 3, 0.0, 0.1).current_dataset_idx == 0      # +1 in make_finetuning_mask
    with pytest.raises(SystemExit):
        _pruner('bogus', 0, 8, 3, 0.0, 0.1)


def test_angle_head_matches_reference():
    """AngleLinear / AngleLoss (config 5 head, stock torch ops on any device) against the reference's outputs."""
    from cpg_amd.models.spherenet import AngleLinear, AngleLoss
    g = load_golden('angle_head')
    lin = AngleLinear(16, 10)
    lin.weight.data.copy_(torch.from_numpy(g['w']))
    x = torch.from_numpy(g['x']).requires_grad_(True)
    t = torch.from_numpy(g['t'])
    cos, phi = lin(x)
    np.testing.assert_allclose(cos.detach().numpy(), g['cos'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(phi.detach().numpy(), g['phi'], rtol=1e-5, atol=1e-5)
    crit = AngleLoss()
    for k in range(3):
        lin.zero_grad()
        loss = crit(lin(x), t)
        loss.backward()
        assert abs(float(loss) - g['losses'][k]) < 1e-5
        np.testing.assert_allclose(lin.weight.grad.numpy(), g['gw'][k], rtol=1e-4, atol=1e-6)
    assert crit.it == 3
