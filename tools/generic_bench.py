#!/usr/bin/env python3
"""Microbenchmark of the generic (non 3x3-s1-p1) masked conv kernels on representative ResNet-50 / SphereNet-20 shapes,
batch 256 by default.  TFLOP/s algorithmic per pass."""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpg_amd import _lib                      # noqa: E402
from cpg_amd.models.layers import _conv_desc  # noqa: E402

# name, C, K, H, k, stride, pad
SHAPES = [('r50 stem 7x7s2', 3, 64, 224, 7, 2, 3), ('r50 1x1 64>256 @56', 64, 256, 56, 1, 1, 0), ('r50 1x1 256>64 @56', 256, 64, 56, 1, 1, 0),
          ('r50 1x1 256>128 @56', 256, 128, 56, 1, 1, 0), ('r50 3x3s2 128 @56', 128, 128, 56, 3, 2, 1), ('r50 1x1 128>512 @28', 128, 512, 28, 1, 1, 0),
          ('r50 1x1 512>128 @28', 512, 128, 28, 1, 1, 0), ('r50 ds 1x1s2 256>512', 256, 512, 56, 1, 2, 0), ('r50 1x1 256>1024 @14', 256, 1024, 14, 1, 1, 0),
          ('r50 1x1 1024>256 @14', 1024, 256, 14, 1, 1, 0), ('r50 1x1 512>2048 @7', 512, 2048, 7, 1, 1, 0), ('r50 1x1 2048>512 @7', 2048, 512, 7, 1, 1, 0),
          ('sph 3x3s2 3>64 @112', 3, 64, 112, 3, 2, 1), ('sph 3x3s2 64>128 @56', 64, 128, 56, 3, 2, 1), ('sph 3x3s2 256>512 @14', 256, 512, 14, 3, 2, 1),
          ('3x3 512>512 @7 (odd map)', 512, 512, 7, 3, 1, 1), ('sph 3x3 64>64 @56', 64, 64, 56, 3, 1, 1), ('sph 3x3 256>256 @14', 256, 256, 14, 3, 1, 1)]


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
    ap.add_argument('--only', default='', help='substring of the shape name')
    a = ap.parse_args()
    L, dev, st, P = _lib.lib(), 'cuda:0', _lib.stream_ptr(), _lib.dptr
    print('%-24s %8s %8s %8s   (TFLOP/s; ms)' % ('shape', 'fwd', 'dgrad', 'wgrad'))
    for name, C, K, H, k, s, p in SHAPES:
        if a.only and a.only not in name:
            continue
        OH = (H + 2 * p - k) // s + 1
        x = torch.randn(a.batch, C, H, H, device=dev)
        w = torch.randn(K, C, k, k, device=dev) * 0.05
        y = torch.empty(a.batch, K, OH, OH, device=dev)
        gy = torch.randn_like(y)
        gx, gw = torch.empty_like(x), torch.empty_like(w)
        d = _conv_desc(x.shape, w.shape, (s, s), (p, p), (1, 1), 1)
        ws, nb = _lib.workspace(L.cpg_conv2d_workspace_bytes(ctypes.byref(d)), dev)
        flops = 2.0 * a.batch * K * OH * OH * C * k * k
        t = [timeit(lambda: L.cpg_conv2d_fwd(ctypes.byref(d), P(x), P(w), None, 5e-3, None, P(y), P(ws), nb, st), a.iters),
             timeit(lambda: L.cpg_conv2d_dgrad(ctypes.byref(d), P(gy), P(w), None, 5e-3, P(gx), P(ws), nb, st), a.iters),
             timeit(lambda: L.cpg_conv2d_wgrad(ctypes.byref(d), P(x), P(gy), P(w), None, 5e-3, P(gw), None, None, P(ws), nb, st), a.iters)]
        print('%-24s %8.1f %8.1f %8.1f   (%.3f %.3f %.3f)' % ((name,) + tuple(flops / ms / 1e9 for ms in t) + tuple(t)), flush=True)


if __name__ == '__main__':
    main()
