#!/usr/bin/env python3
"""Winograd weight-gradient kernel (conv3x3_wino_wgrad.hip) against the direct one: max error and time per VGG16 layer
(development tool; calls the internal entry point cpg_conv3x3_wino_wgrad directly).

    python tools/wino_wgrad_bench.py [--batch 256] [--iters 3] [--layers f3,f17]
"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpg_amd import _lib                      # noqa: E402
from cpg_amd._lib import ConvDesc             # noqa: E402
from cpg_amd.models.layers import _conv_desc  # noqa: E402
from tools.conv_bench import VGG, timeit      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--iters', type=int, default=3)
    ap.add_argument('--layers', default='')
    ap.add_argument('--shape', default='', help='"N,C,K,H" instead of the VGG list')
    a = ap.parse_args()
    L = _lib.lib()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    raw.cpg_conv3x3_wino_wgrad_workspace.restype = ctypes.c_size_t
    raw.cpg_conv3x3_wino_wgrad_workspace.argtypes = [ctypes.POINTER(ConvDesc)]
    raw.cpg_conv3x3_wino_wgrad_ok.argtypes = [ctypes.POINTER(ConvDesc)]
    raw.cpg_conv3x3_wino_wgrad.restype = ctypes.c_int
    raw.cpg_conv3x3_wino_wgrad.argtypes = [ctypes.POINTER(ConvDesc)] + [ctypes.c_void_p] * 4 + [ctypes.c_float] + [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_void_p]
    dev = 'cuda:0'
    st = _lib.stream_ptr()
    P = _lib.dptr
    cp = ctypes.c_void_p
    sel = set(a.layers.split(',')) if a.layers else None
    layers = [(n, C, K, H, m, a.batch) for n, C, K, H, m in VGG]
    if a.shape:
        N, C, K, H = (int(v) for v in a.shape.split(','))
        layers = [('x0', C, K, H, 1, N)]
    tot = [0.0, 0.0]
    for name, C, K, H, mult, N in layers:
        if sel and name not in sel:
            continue
        torch.manual_seed(0)
        x = torch.randn(N, C, H, H, device=dev).relu_()
        w = torch.randn(K, C, 3, 3, device=dev) * 0.05
        gy = torch.randn(N, K, H, H, device=dev)
        gw0, gw1 = torch.empty_like(w), torch.full_like(w, float('nan'))
        d = _conv_desc(x.shape, w.shape, (1, 1), (1, 1), (1, 1), 1)
        if not raw.cpg_conv3x3_wino_wgrad_ok(ctypes.byref(d)):
            print('%-5s not eligible' % name)
            continue
        need = L.cpg_conv2d_workspace_bytes(ctypes.byref(d))
        _lib.set_option('CPG_NO_WINO_WGRAD', 1)                 # (the direct kernel's partial sums may need more)
        need = max(need, L.cpg_conv2d_workspace_bytes(ctypes.byref(d)))
        _lib.set_option('CPG_NO_WINO_WGRAD', None)
        ws, nb = _lib.workspace(need, dev)
        nbw = raw.cpg_conv3x3_wino_wgrad_workspace(ctypes.byref(d))
        wsw = torch.empty(nbw // 4 + 64, device=dev)
        flops = 2.0 * N * K * H * H * C * 9

        def f0():
            _lib.set_option('CPG_NO_WINO_WGRAD', 1)
            rc = L.cpg_conv2d_wgrad(ctypes.byref(d), P(x), P(gy), P(w), None, 5e-3, P(gw0), None, None, P(ws), nb, st)
            _lib.set_option('CPG_NO_WINO_WGRAD', None)
            assert rc == 0, (rc, L.cpg_last_error())

        def f1():
            rc = raw.cpg_conv3x3_wino_wgrad(ctypes.byref(d), cp(x.data_ptr()), cp(gy.data_ptr()), cp(w.data_ptr()), None, 5e-3, cp(gw1.data_ptr()), None,
                                            cp(wsw.data_ptr()), nbw, st)
            assert rc == 0, (rc, L.cpg_last_error())
        t0 = timeit(f0, a.iters)
        t1 = timeit(f1, a.iters)
        err = ((gw1 - gw0).abs().max() / gw0.abs().max()).item()
        print('%-5s wgrad  direct %7.3f ms %6.1f TF | winograd %7.3f ms %6.1f TF-equivalent (%.2fx)  max err / max |ref| %.2e  (workspace %.0f MB)'
              % (name, t0, flops / t0 / 1e9, t1, flops / t1 / 1e9, t0 / t1, err, nbw / 1e6), flush=True)
        tot[0] += t0 * mult
        tot[1] += t1 * mult
    print('TOTAL wgrad  direct %8.3f ms | winograd %8.3f ms (%.2fx)' % (tot[0], tot[1], tot[0] / max(tot[1], 1e-9)))


if __name__ == '__main__':
    main()
