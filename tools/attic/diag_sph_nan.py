#!/usr/bin/env python3
"""SphereNet-20 + AngleLoss on bench.py's synthetic data (random images, random labels): where does the loss stop being finite, and does the
SAME run on torch's own ops (MIOpen convolutions, nn.PReLU, nn.Linear; same initial state, data and optimizer) do the same?
   python tools/attic/diag_sph_nan.py [steps] [lr]"""
import os
import sys
import torch
import torch.nn as nn
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench                                             # noqa: E402
from cpg_amd.models import layers as nl, fused_bn        # noqa: E402
from cpg_amd.models.spherenet import AngleLoss           # noqa: E402
from cpg_amd.utils.prune import SparsePruner             # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 80
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-2
dev = torch.device('cuda', 0)
bench.DATASET = bench.ARCHS['spherenet20']['dataset']
g = torch.Generator(device=dev).manual_seed(1)
arch = bench.ARCHS['spherenet20']
pool = [(torch.randn(256, 3, arch['size'], arch['size'], generator=g, device=dev),
         torch.randint(0, arch['classes'], (256,), generator=g, device=dev)) for _ in range(3)]


def run(tag):
    net = bench.build_model(dev, 'spherenet20')
    masks = {n: torch.zeros(m.weight.shape, dtype=torch.uint8, device=dev) for n, m in net.named_modules()
             if isinstance(m, (nl.SharableConv2d, nl.SharableLinear))}
    pruner = SparsePruner(net, masks, bench.make_args('finetune', 1), 0, 0, 1)
    pruner.make_finetuning_mask()
    opt = bench.make_optimizers(net, pruner, lr, None)
    crit = AngleLoss()
    net.train()
    out = []
    for i in range(steps):
        x, t = pool[i % 3]
        opt.zero_grad()
        emb = net.classifier[0](net._trunk(x))
        o = net.classifier[1](emb)
        loss = crit(o, t)
        loss.backward()
        pruner.do_weight_decay_and_make_grads_zero()
        opt.step()
        gmax = max(float(p.grad.abs().max()) for p in net.parameters() if p.grad is not None)
        wmax = max(float(p.detach().abs().max()) for p in net.parameters())
        out.append((float(loss), float(emb.detach().norm(dim=1).min()), float(emb.detach().norm(dim=1).max()), gmax, wmax))
        if not all(map(lambda v: v == v and abs(v) != float('inf'), out[-1])):
            break
    print('== %s: %d steps run' % (tag, len(out)))
    for i, r in enumerate(out):
        if i < 6 or i % 5 == 0 or i >= len(out) - 6:
            print('%s step %3d loss %.5g  |emb| min %.4g max %.4g  max|grad| %.4g  max|w| %.4g' % ((tag, i) + r))
    return out


a = run('hip')
fused_bn.ENABLED = False
nl.SharableConv2d.forward = lambda self, input, layer_info=None, name=None, **kw: F.conv2d(input, self.weight, self.bias, self.stride,
                                                                                         self.padding, self.dilation, self.groups)
nl.SharableConv2d.forward_with_skip = lambda self, input, **kw: (F.conv2d(input, self.weight, self.bias, self.stride, self.padding,
                                                                          self.dilation, self.groups), None, input)
nl.HeadLinear.forward = nn.Linear.forward
b = run('torch')
n = min(len(a), len(b))
print('first steps where the losses differ by more than 1e-3 relative:', [i for i in range(n) if abs(a[i][0] - b[i][0]) > 1e-3 * abs(b[i][0])][:5])
