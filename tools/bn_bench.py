#!/usr/bin/env python3
"""Microbenchmark of the fused BatchNorm -> ReLU (-> MaxPool 2x2) kernels on the 13 VGG16 activation shapes at batch 256:
ms per call (forward = stats + apply, backward = reduce + apply) and the HBM rate over the algorithmic bytes."""
import argparse
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpg_amd.models import fused_bn  # noqa: E402

# (C, H, pooled, multiplicity)
VGG = [(64, 224, False, 1), (64, 224, True, 1), (128, 112, False, 1), (128, 112, True, 1), (256, 56, False, 2), (256, 56, True, 1),
       (512, 28, False, 2), (512, 28, True, 1), (512, 14, False, 2), (512, 14, True, 1)]


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--iters', type=int, default=5)
    a = ap.parse_args()
    dev = 'cuda:0'
    tf = tb = 0.0
    print('%-22s %9s %9s %9s %9s' % ('shape', 'fwd ms', 'TB/s', 'bwd ms', 'TB/s'))
    for C, H, pooled, mult in VGG:
        x = torch.randn(a.batch, C, H, H, device=dev, requires_grad=True)
        bn = nn.BatchNorm2d(C).to(dev).train()
        fn = fused_bn.bn_relu_pool if pooled else fused_bn.bn_relu
        y = fn(x, bn)
        gy = torch.randn_like(y)
        n_in, n_out = x.numel() * 4.0, y.numel() * 4.0
        fwd_bytes = 2 * n_in + n_out                  # stats read, apply read + write
        bwd_bytes = 2 * (n_in + n_out) + n_in         # reduce: x + g; apply: x + g + dx
        t_f = timeit(lambda: fn(x, bn), a.iters)

        def bwd():
            x.grad = None
            y.backward(gy, retain_graph=True)
        t_b = timeit(bwd, a.iters)
        print('%-22s %9.3f %9.2f %9.3f %9.2f' % ('%dx%dx%d%s' % (C, H, H, ' pool' if pooled else ''), t_f, fwd_bytes / t_f / 1e9,
                                                  t_b, bwd_bytes / t_b / 1e9), flush=True)
        tf += t_f * mult
        tb += t_b * mult
    print('TOTAL (13 layers)  fwd %.3f ms  bwd %.3f ms' % (tf, tb))


if __name__ == '__main__':
    main()
