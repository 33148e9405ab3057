#!/usr/bin/env python3
"""Diagnostic: step-0 gradients of the train_steps_* fixtures (reference, CPU) against the HIP path, per watched layer,
with the fused BatchNorm kernels on and off.  python tools/attic/diag_train_steps.py [resnet50|spherenet20]"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cpg_amd.models as M                       # noqa: E402
from cpg_amd.models import fused_bn              # noqa: E402
from cpg_amd.models import layers as nl          # noqa: E402
from cpg_amd.models.spherenet import AngleLoss   # noqa: E402

DEV = 'cuda:0'


def run(arch, fused):
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'train_steps_%s.npz' % arch))
    width, ncls = float(g['width']), int(g['num_classes'])
    dataset = 'face_verification' if arch == 'spherenet20' else 't1'
    torch.manual_seed(1)
    kw = dict(dataset_history=[], dataset2num_classes={}, network_width_multiplier=width, shared_layer_info={})
    net = M.resnet50(**kw) if arch == 'resnet50' else M.spherenet20(**kw)
    net.add_dataset(dataset, ncls)
    net.set_dataset(dataset)
    if arch == 'resnet50':
        torch.manual_seed(2)
        for m in net.modules():
            if isinstance(m, nl.SharableConv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
    net = net.to(DEV).train()
    fused_bn.ENABLED = fused
    x = torch.from_numpy(g['x'][0]).to(DEV)
    t = torch.from_numpy(g['t'][0]).to(DEV)
    out = net(x)
    crit = AngleLoss() if dataset == 'face_verification' else nn.CrossEntropyLoss()
    loss = crit(out, t)
    loss.backward()
    print('%s fused_bn=%s loss %.7f (ref %.7f)' % (arch, fused, float(loss), float(g['losses'][0])))
    mods = dict(net.named_modules())
    for n in [str(w) for w in g['watch']]:
        ref = g['grad/' + n][0]
        got = mods[n].weight.grad.cpu().numpy()
        print('   %-18s max|g| %-10.4g  maxdiff/scale %.3g' % (n, np.abs(ref).max(), np.abs(got - ref).max() / np.abs(ref).max()))
    params = dict(net.named_parameters())
    for k in [k for k in g.files if k.startswith('g0/')]:
        ref = g[k]
        got = params[k[3:]].grad.cpu().numpy()
        print('   %-18s max|g| %-10.4g  maxdiff/scale %.3g' % (k[3:], np.abs(ref).max(), np.abs(got - ref).max() / (np.abs(ref).max() + 1e-30)))


if __name__ == '__main__' and not os.environ.get('DIAG_FUSED_VS_STOCK') and not os.environ.get('DIAG_PROBE'):
    for arch in (sys.argv[1:] or ['resnet50', 'spherenet20']):
        for fused in (True, False):
            run(arch, fused)


def fused_vs_stock(arch='resnet50'):
    """All parameter gradients: fused BN kernels against stock torch BN (which matches the reference at ~1e-4), listed
    from the output end of the network backwards -- shows where an error enters the backward pass."""
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'train_steps_%s.npz' % arch))
    width, ncls = float(g['width']), int(g['num_classes'])
    grads = {}
    for fused in (True, False):
        torch.manual_seed(1)
        kw = dict(dataset_history=[], dataset2num_classes={}, network_width_multiplier=width, shared_layer_info={})
        net = M.resnet50(**kw)
        net.add_dataset('t1', ncls)
        net.set_dataset('t1')
        torch.manual_seed(2)
        for m in net.modules():
            if isinstance(m, nl.SharableConv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
        net = net.to(DEV).train()
        fused_bn.ENABLED = fused
        x = torch.from_numpy(g['x'][0]).to(DEV)
        t = torch.from_numpy(g['t'][0]).to(DEV)
        nn.functional.cross_entropy(net(x), t).backward()
        grads[fused] = {n: p.grad.detach().double().cpu() for n, p in net.named_parameters() if p.grad is not None}
    names = list(grads[True])
    for n in reversed(names):
        a, b = grads[True][n], grads[False][n]
        print('   %-32s %-18s max|g| %-10.4g maxdiff/scale %.3g' % (n, tuple(a.shape), float(b.abs().max()),
                                                                      float((a - b).abs().max() / (b.abs().max() + 1e-30))))


if __name__ == '__main__' and os.environ.get('DIAG_FUSED_VS_STOCK'):
    fused_vs_stock()


def probe_block(arch='resnet50', block='layer4.0'):
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'train_steps_%s.npz' % arch))
    width, ncls = float(g['width']), int(g['num_classes'])
    cap = {}
    for fused in (True, False):
        torch.manual_seed(1)
        kw = dict(dataset_history=[], dataset2num_classes={}, network_width_multiplier=width, shared_layer_info={})
        net = M.resnet50(**kw)
        net.add_dataset('t1', ncls)
        net.set_dataset('t1')
        torch.manual_seed(2)
        for m in net.modules():
            if isinstance(m, nl.SharableConv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
        net = net.to(DEV).train()
        fused_bn.ENABLED = fused
        blk = dict(net.named_modules())[block]
        c = cap[fused] = {}

        def pre1(mod, inp):
            c['x_in'] = inp[0].detach().clone()

        def post1(mod, inp, out):
            c['c1'] = out.detach().clone()
            out.register_hook(lambda gr: c.__setitem__('g_c1', gr.detach().clone()))

        def pre2(mod, inp):
            c['a1'] = inp[0].detach().clone()
            inp[0].register_hook(lambda gr: c.__setitem__('g_a1', gr.detach().clone()))
        blk.conv1.register_forward_pre_hook(pre1)
        blk.conv1.register_forward_hook(post1)
        blk.conv2.register_forward_pre_hook(pre2)
        x = torch.from_numpy(g['x'][0]).to(DEV)
        t = torch.from_numpy(g['t'][0]).to(DEV)
        nn.functional.cross_entropy(net(x), t).backward()
        c['dbeta'] = blk.bn1.bias.grad.detach().clone()
    for k in ('x_in', 'c1', 'a1', 'g_a1', 'g_c1', 'dbeta'):
        a, b = cap[True][k].double(), cap[False][k].double()
        print('%-6s shape %-18s max %.4g  fused-vs-stock maxdiff/scale %.3g' % (k, tuple(a.shape), float(b.abs().max()),
                                                                                 float((a - b).abs().max() / (b.abs().max() + 1e-30))))
    c1 = cap[False]['c1'].double()
    var = c1.var(dim=(0, 2, 3), unbiased=False)
    mean = c1.mean(dim=(0, 2, 3))
    print('conv1 output per-channel var: min %.3g median %.3g ; |mean| max %.3g' % (float(var.min()), float(var.median()), float(mean.abs().max())))
    mf, ms = cap[True]['a1'] > 0, cap[False]['a1'] > 0
    print('relu mask differences:', int((mf != ms).sum()), 'of', mf.numel())
    d = (cap[True]['dbeta'] - cap[False]['dbeta']).abs()
    worst = torch.topk(d, 5).indices.tolist()
    for ch in worst:
        print(' channel %d: dbeta fused %.5g stock %.5g  var %.3g  mask diffs %d  min|a1|>0 %.3g' % (
            ch, float(cap[True]['dbeta'][ch]), float(cap[False]['dbeta'][ch]), float(var[ch]), int((mf[:, ch] != ms[:, ch]).sum()),
            float(cap[False]['a1'][:, ch][cap[False]['a1'][:, ch] > 0].min()) if bool((cap[False]['a1'][:, ch] > 0).any()) else -1))


if __name__ == '__main__' and os.environ.get('DIAG_PROBE'):
    probe_block(block=os.environ['DIAG_PROBE'])
