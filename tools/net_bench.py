#!/usr/bin/env python3
"""Train-step time (forward + backward + SGD) of the other topologies (configs 4 / 5 of BASELINE.json: ResNet-50 224x224,
SphereNet-20 112x112) through the masked layers, batch 256 by default.  Not the headline bench -- a check that the kernels
behind those parity-test configurations are not pathological."""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cpg_amd.models as M  # noqa: E402

FLOP_PER_IMG = {'resnet50': 3 * 2 * 4.087e9, 'spherenet20': 3 * 2 * 2.029e9, 'vgg16': 92.62e9}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--arch', default='resnet50')
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--math', default='fp32', choices=['fp32', 'bf16', 'bf16x3'], help='arithmetic of the 3x3 s1 p1 convolutions (opt-in modes)')
    ap.add_argument('--piggymask', action='store_true', help='task >= 2: a piggymask on every masked layer + Adam on them')
    a = ap.parse_args()
    dev = 'cuda:0'
    from cpg_amd.models import layers as _nl
    _nl.set_conv_math(a.math)
    torch.manual_seed(1)
    kw = dict(dataset_history=[], dataset2num_classes={}, network_width_multiplier=1.0, shared_layer_info={})
    vgg_cfg = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']
    net = M.custom_vgg(vgg_cfg, **kw) if a.arch == 'vgg16' else getattr(M, a.arch)(**kw)
    net.add_dataset('t', 10)
    net.set_dataset('t')
    net = net.to(dev).train()
    shape = (a.batch, 3, 112, 112) if a.arch.startswith('sphere') else (a.batch, 3, 224, 224)
    x = torch.randn(*shape, device=dev)
    t = torch.randint(0, 10, (a.batch,), device=dev)
    opts = []
    if a.piggymask:
        from cpg_amd.models import layers as nl
        pms = []
        for m in net.modules():
            if isinstance(m, (nl.SharableConv2d, nl.SharableLinear)):
                m.piggymask = torch.nn.Parameter(torch.rand_like(m.weight) * 0.012)
                pms.append(m.piggymask)
        opts.append(torch.optim.Adam(pms, lr=1e-4))
        opts.append(torch.optim.SGD([p for p in net.parameters() if all(p is not q for q in pms)], lr=1e-3, momentum=0.9))
    else:
        opts.append(torch.optim.SGD(net.parameters(), lr=1e-3, momentum=0.9))

    def step():
        for o in opts:
            o.zero_grad(set_to_none=True)
        out = net(x)
        loss = F.cross_entropy(out, t)
        loss.backward()
        for o in opts:
            o.step()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    print('%s batch %d (%s): %.2f ms/step, %.0f img/s, %.1f TFLOP/s (conv MACs only)' % (
        a.arch, a.batch, a.math, ms, a.batch / ms * 1e3, a.batch * FLOP_PER_IMG.get(a.arch, 0) / ms / 1e9))


if __name__ == '__main__':
    main()
