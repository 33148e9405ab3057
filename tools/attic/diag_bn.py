#!/usr/bin/env python3
"""Diagnostic: fused BN(+ReLU / +add+ReLU) forward/backward against an fp64 torch reference on small-plane shapes."""
import os
import sys

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cpg_amd.models import fused_bn              # noqa: E402

DEV = 'cuda:0'


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


def one(N, C, H, W, relu, add, seed=0, beta0=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, C, H, W, generator=g) * 2 + torch.randn(1, C, 1, 1, generator=g)
    gy = torch.randn(N, C, H, W, generator=g)
    res = torch.randn(N, C, H, W, generator=g) if add else None
    bn = nn.BatchNorm2d(C)
    if not beta0:
        bn.weight.data = torch.rand(C, generator=g) + 0.5
        bn.bias.data = torch.randn(C, generator=g) * 0.3
    # fp64 reference
    bn64 = nn.BatchNorm2d(C).double()
    bn64.load_state_dict({k: v.double() if v.dtype.is_floating_point else v for k, v in bn.state_dict().items()})
    x64 = x.double().requires_grad_(True)
    r64 = res.double().requires_grad_(True) if add else None
    y64 = bn64(x64)
    if add:
        y64 = y64 + r64
    if relu or add:
        y64 = torch.relu(y64)
    y64.backward(gy.double())
    # fused
    bnd = bn.to(DEV).train()
    xd = x.to(DEV).requires_grad_(True)
    rd = res.to(DEV).requires_grad_(True) if add else None
    act = nn.ReLU(inplace=True)
    if add:
        yd = fused_bn.bn_add_act(bnd, act, xd, rd)
    else:
        yd = fused_bn.bn_act(bnd, act if relu else None, xd)
    yd.backward(gy.to(DEV))
    out = 'N%d C%d %dx%d relu=%d add=%d beta0=%d | y %.2e  dx %.2e  dgamma %.2e  dbeta %.2e' % (
        N, C, H, W, relu, add, beta0, rel(yd.detach().cpu(), y64.detach()), rel(xd.grad.cpu(), x64.grad),
        rel(bnd.weight.grad.cpu(), bn64.weight.grad), rel(bnd.bias.grad.cpu(), bn64.bias.grad))
    if add:
        out += '  dres %.2e' % rel(rd.grad.cpu(), r64.grad)
    print(out)


if __name__ == '__main__':
    for beta0 in (True, False):
        for shape in [(4, 128, 4, 4), (4, 64, 4, 4), (4, 128, 2, 2), (4, 16, 16, 16), (4, 32, 8, 8), (8, 64, 56, 56), (3, 5, 7, 9)]:
            one(*shape, relu=1, add=0, beta0=beta0)
        one(4, 256, 4, 4, relu=0, add=0, beta0=beta0)
        one(4, 256, 4, 4, relu=0, add=1, beta0=beta0)
        one(4, 512, 2, 2, relu=0, add=1, beta0=beta0)
