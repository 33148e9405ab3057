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

FLOP_PER_IMG = {'resnet50': 3 * 2 * 4.087e9, 'spherenet20': 3 * 2 * 2.029e9}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--arch', default='resnet50')
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--steps', type=int, default=10)
    a = ap.parse_args()
    dev = 'cuda:0'
    torch.manual_seed(1)
    net = getattr(M, a.arch)(dataset_history=[], dataset2num_classes={}, network_width_multiplier=1.0, shared_layer_info={})
    net.add_dataset('t', 10)
    net.set_dataset('t')
    net = net.to(dev).train()
    shape = (a.batch, 3, 224, 224) if a.arch.startswith('res') else (a.batch, 3, 112, 112)
    x = torch.randn(*shape, device=dev)
    t = torch.randint(0, 10, (a.batch,), device=dev)
    opt = torch.optim.SGD(net.parameters(), lr=1e-3, momentum=0.9)

    def step():
        opt.zero_grad(set_to_none=True)
        out = net(x)
        loss = F.cross_entropy(out, t)
        loss.backward()
        opt.step()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    print('%s batch %d: %.1f ms/step, %.0f img/s, %.1f TFLOP/s (conv MACs only)' % (
        a.arch, a.batch, ms, a.batch / ms * 1e3, a.batch * FLOP_PER_IMG.get(a.arch, 0) / ms / 1e9))


if __name__ == '__main__':
    main()
