#!/usr/bin/env python3
"""Repeat every 3x3 layer of a VGG16 pass (batch 256) R times -- forward with BatchNorm statistics, input gradient, weight gradient --
and compare every repetition bit for bit with the first: the shared-staging / shared-transform Winograd kernels exchange operands
through double-buffered LDS behind LDS-only barriers, and a hazard there would show up as run-to-run differences at full size and
full occupancy before it shows up in a tolerance test.   usage: python tools/soak_determinism.py [--reps 20] [--hint]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpg_amd import _lib
from cpg_amd.models import layers as nl
from cpg_amd.models import fused_bn

ap = argparse.ArgumentParser()
ap.add_argument('--reps', type=int, default=20)
ap.add_argument('--batch', type=int, default=256)
ap.add_argument('--hint', action='store_true', help='the multi-GPU launch plans (cpg_set_shared_chip_hint(1))')
a = ap.parse_args()
dev = torch.device('cuda', 0)
if a.hint:
    _lib.lib().cpg_set_shared_chip_hint(1)
LAYERS = [(64, 64, 224), (64, 128, 112), (128, 128, 112), (128, 256, 56), (256, 256, 56), (256, 512, 28), (512, 512, 28), (512, 512, 14)]
g = torch.Generator(device=dev).manual_seed(1)
bad = 0
for C, K, H in LAYERS:
    x = torch.randn(a.batch, C, H, H, generator=g, device=dev).relu_().requires_grad_(True)
    gy = torch.randn(a.batch, K, H, H, generator=g, device=dev)
    layer = nl.SharableConv2d(C, K, 3, padding=1, bias=False).to(dev)
    layer.weight.data.normal_(0, 0.05, generator=g)
    bn = torch.nn.BatchNorm2d(K).to(dev)
    first = None
    for r in range(a.reps):
        layer.zero_grad(); x.grad = None
        y, stats = layer.forward_with_bn_stats(x)
        y.backward(gy)
        cur = (y.detach().clone(), stats.detach().clone() if torch.is_tensor(stats) else None, x.grad.clone(), layer.weight.grad.clone())
        if first is None:
            first = cur
        else:
            for name, p, q in zip(('y', 'stats', 'gx', 'gw'), first, cur):
                if p is not None and not torch.equal(p, q):
                    bad += 1
                    print('MISMATCH %d->%d @%d rep %d %s: max |diff| %.3e' % (C, K, H, r, name, float((p - q).abs().max())), flush=True)
    print('%d->%d @%d: %d repetitions' % (C, K, H, a.reps), flush=True)
    del x, gy, layer, first, cur
    torch.cuda.empty_cache()
print('soak: %d mismatches' % bad)
sys.exit(1 if bad else 0)
