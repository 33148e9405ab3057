#!/usr/bin/env python3
"""Does running a layer's input-gradient and weight-gradient launches on two streams (fork after gy is ready, join before the next
consumer) beat running them back to back?  The two are independent; each is a grid of equally long one-wave-per-SIMD blocks, so
the last round of one leaves CUs idle that the other's blocks could take.

    python tools/fork_bench.py [--batch 256] [--iters 10]
"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpg_amd import _lib                      # noqa: E402
from cpg_amd.models.layers import _conv_desc  # noqa: E402

SHAPES = [('vgg 64>64 @224', 64, 64, 224), ('vgg 128>128 @112', 128, 128, 112), ('vgg 256>256 @56', 256, 256, 56), ('vgg 256>512 @28', 256, 512, 28),
          ('vgg 512>512 @28', 512, 512, 28), ('vgg 512>512 @14', 512, 512, 14), ('sph 64>64 @56', 64, 64, 56), ('sph 128>128 @28', 128, 128, 28),
          ('sph 256>256 @14', 256, 256, 14)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--iters', type=int, default=10)
    a = ap.parse_args()
    L, dev, P = _lib.lib(), 'cuda:0', _lib.dptr
    main_s = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    print('%-20s %9s %9s %9s %9s   %s' % ('shape', 'dgrad', 'wgrad', 'seq', 'forked', 'forked / seq'))
    for name, C, K, H in SHAPES:
        x = torch.randn(a.batch, C, H, H, device=dev)
        w = torch.randn(K, C, 3, 3, device=dev) * 0.05
        gy = torch.randn(a.batch, K, H, H, device=dev)
        gx, gw = torch.empty_like(x), torch.empty_like(w)
        d = _conv_desc(x.shape, w.shape, (1, 1), (1, 1), (1, 1), 1)
        nws = L.cpg_conv2d_workspace_bytes(ctypes.byref(d))
        ws1, nb = _lib.workspace(nws, dev)
        ws2, _ = _lib.workspace(nws, dev)
        st_main, st_side = main_s.cuda_stream, side.cuda_stream

        def dgrad(st=st_main):
            L.cpg_conv2d_dgrad(ctypes.byref(d), P(gy), P(w), None, 5e-3, P(gx), P(ws1), nb, st)

        def wgrad(st=st_main, ws=ws1):
            L.cpg_conv2d_wgrad(ctypes.byref(d), P(x), P(gy), P(w), None, 5e-3, P(gw), None, None, P(ws), nb, st)

        def seq():
            dgrad()
            wgrad()

        def forked():
            side.wait_stream(main_s)
            wgrad(st_side, ws2)
            dgrad()
            main_s.wait_stream(side)

        def timeit(fn):
            fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(a.iters):
                fn()
            e.record()
            torch.cuda.synchronize()
            return s.elapsed_time(e) / a.iters
        t = [timeit(dgrad), timeit(wgrad), timeit(seq), timeit(forked)]
        t += [timeit(seq), timeit(forked)]
        print('%-20s %9.3f %9.3f %9.3f %9.3f   %.3f   (again: %.3f %.3f)' % (name, t[0], t[1], t[2], t[3], t[3] / t[2], t[4], t[5]), flush=True)


if __name__ == '__main__':
    main()
