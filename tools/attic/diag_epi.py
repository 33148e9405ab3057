#!/usr/bin/env python3
"""dump logits of the tiny VGG (fused / unfused) and plain conv outputs to a file, for A/B of two library builds"""
import sys, os
import numpy as np, torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cpg_amd.models as M
VGG_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']
def build(arch, width, ncls=5):
    torch.manual_seed(1)
    kw = dict(dataset_history=[], dataset2num_classes={}, network_width_multiplier=width, shared_layer_info={})
    m = M.custom_vgg_cifar100(VGG_CFG, **kw)
    m.add_dataset('t1', ncls); m.set_dataset('t1')
    return m
from cpg_amd.models.fused_bn import FusedSequential
import cpg_amd.models.layers as nl
DEV = 'cuda:0'
out = {}
torch.manual_seed(0)
net = build('vgg_cifar100', 0.25).to(DEV)
g = torch.Generator().manual_seed(4)
x = torch.randn(16, 3, 32, 32, generator=g).to(DEV)
for fuse in (True, False):
    net.train(); FusedSequential.fuse = fuse
    out['logits_%d' % fuse] = net(x).detach().cpu().numpy()
for (N, C, K, H) in [(16, 16, 16, 32), (16, 32, 32, 16), (16, 64, 64, 8), (16, 128, 128, 4), (4, 64, 128, 16), (2, 32, 48, 8), (16, 128, 128, 2), (16, 128, 128, 8), (16, 64, 128, 4)]:
    conv = nl.SharableConv2d(C, K, 3, padding=1, bias=False).to(DEV)
    gg = torch.Generator().manual_seed(C + K)
    conv.weight.data.copy_(torch.randn(K, C, 3, 3, generator=gg) * 0.1)
    xi = torch.randn(N, C, H, H, generator=gg).to(DEV)
    out['conv_%d_%d_%d_%d' % (N, C, K, H)] = conv(xi).detach().cpu().numpy()
    y, st = nl._MaskedConv2dFn.apply(xi, conv.weight, None, None, 5e-3, (1, 1), (1, 1), (1, 1), 1, True)
    out['convs_%d_%d_%d_%d' % (N, C, K, H)] = y.detach().cpu().numpy()
    out['stats_%d_%d_%d_%d' % (N, C, K, H)] = st.detach().cpu().numpy()
np.savez(sys.argv[1], **out)
