#!/usr/bin/env python3
"""Winograd F(2x2,3x3) forward / input-gradient kernel (conv3x3_wino.hip) against the direct kernels: max error and time per
VGG16 layer.  Calls the internal entry point cpg_conv3x3_wino_run directly (development tool).

    python tools/wino_bench.py [--batch 256] [--iters 5] [--layers f3,f17]
"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpg_amd import _lib                      # noqa: E402
from cpg_amd.models.layers import _conv_desc  # noqa: E402
from tools.conv_bench import VGG, timeit      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--layers', default='')
    ap.add_argument('--pm', action='store_true')
    a = ap.parse_args()
    L = _lib.lib()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    raw.cpg_conv3x3_wino_pack_bytes.restype = ctypes.c_size_t
    raw.cpg_conv3x3_wino_pack_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
    raw.cpg_conv3x3_wino_tiles.restype = ctypes.c_int
    raw.cpg_conv3x3_wino_tiles.argtypes = [ctypes.c_int] * 5
    raw.cpg_conv3x3_wino_run.restype = ctypes.c_int
    raw.cpg_conv3x3_wino_run.argtypes = [ctypes.c_int] * 8 + [ctypes.c_void_p] * 3 + [ctypes.c_float] + [ctypes.c_void_p] * 4 + [ctypes.c_size_t, ctypes.c_void_p]
    dev = 'cuda:0'
    st = _lib.stream_ptr()
    P = _lib.dptr
    sel = set(a.layers.split(',')) if a.layers else None
    tot = {}
    for name, C, K, H, mult in VGG:
        if (sel and name not in sel) or C < 16:
            continue
        torch.manual_seed(0)
        x = torch.randn(a.batch, C, H, H, device=dev).relu_()
        w = torch.randn(K, C, 3, 3, device=dev) * (2.0 / (9 * C)) ** 0.5
        pm = torch.rand(K, C, 3, 3, device=dev) * 0.012 if a.pm else None
        gy = torch.randn(a.batch, K, H, H, device=dev)
        y0, y1 = torch.empty(a.batch, K, H, H, device=dev), torch.empty(a.batch, K, H, H, device=dev)
        gx0, gx1 = torch.empty_like(x), torch.empty_like(x)
        d = _conv_desc(x.shape, w.shape, (1, 1), (1, 1), (1, 1), 1)
        ws, nb = _lib.workspace(L.cpg_conv2d_workspace_bytes(ctypes.byref(d)), dev)
        nbw = max(raw.cpg_conv3x3_wino_pack_bytes(C, K), raw.cpg_conv3x3_wino_pack_bytes(K, C))
        wsw = torch.empty(nbw // 4 + 64, device=dev)
        tiles = raw.cpg_conv3x3_wino_tiles(a.batch, C, K, H, H)
        stats = torch.zeros(K * tiles * 2, device=dev)
        flops = 2.0 * a.batch * K * H * H * C * 9
        cp = ctypes.c_void_p

        def chk(rc):
            assert rc == 0, (rc, L.cpg_last_error())

        runs = {
            'fwd': (lambda: chk(L.cpg_conv2d_fwd(ctypes.byref(d), P(x), P(w), P(pm), 5e-3, None, P(y0), P(ws), nb, st)),
                    lambda: chk(raw.cpg_conv3x3_wino_run(0, a.batch, C, K, H, H, K, C, cp(x.data_ptr()), cp(w.data_ptr()), cp(pm.data_ptr()) if a.pm else None, 5e-3,
                                                         None, cp(y1.data_ptr()), None, cp(wsw.data_ptr()), nbw, st)), y0, y1),
            'fwdst': (lambda: chk(L.cpg_conv2d_fwd(ctypes.byref(d), P(x), P(w), P(pm), 5e-3, None, P(y0), P(ws), nb, st)),
                      lambda: chk(raw.cpg_conv3x3_wino_run(0, a.batch, C, K, H, H, K, C, cp(x.data_ptr()), cp(w.data_ptr()), cp(pm.data_ptr()) if a.pm else None, 5e-3,
                                                           None, cp(y1.data_ptr()), cp(stats.data_ptr()), cp(wsw.data_ptr()), nbw, st)), y0, y1),
            'dgrad': (lambda: chk(L.cpg_conv2d_dgrad(ctypes.byref(d), P(gy), P(w), P(pm), 5e-3, P(gx0), P(ws), nb, st)),
                      lambda: chk(raw.cpg_conv3x3_wino_run(1, a.batch, K, C, H, H, K, C, cp(gy.data_ptr()), cp(w.data_ptr()), cp(pm.data_ptr()) if a.pm else None, 5e-3,
                                                           None, cp(gx1.data_ptr()), None, cp(wsw.data_ptr()), nbw, st)), gx0, gx1),
        }
        for k, (f0, f1, r0, r1) in runs.items():
            r1.fill_(float('nan'))
            _lib.set_option('CPG_NO_WINO', 1)          # the public entry points dispatch to the Winograd kernels themselves
            t0 = timeit(f0, a.iters)
            _lib.set_option('CPG_NO_WINO', None)
            t1 = timeit(f1, a.iters)
            err = ((r1 - r0).abs().max() / r0.abs().max()).item()
            extra = ''
            if k == 'fwdst':
                s = stats.view(K, tiles, 2).sum(1)
                e1 = ((s[:, 0] - y1.sum((0, 2, 3))).abs().max() / y1.sum((0, 2, 3)).abs().max()).item()
                e2 = ((s[:, 1] - (y1 * y1).sum((0, 2, 3))).abs().max() / (y1 * y1).sum((0, 2, 3)).abs().max()).item()
                extra = '  stats err %.1e %.1e' % (e1, e2)
            print('%-5s %-6s direct %7.3f ms %6.1f TF | winograd %7.3f ms %6.1f TF-equivalent (%.2fx)  max err / max |ref| %.2e%s'
                  % (name, k, t0, flops / t0 / 1e9, t1, flops / t1 / 1e9, t0 / t1, err, extra), flush=True)
            t = tot.setdefault(k, [0.0, 0.0])
            t[0] += t0 * mult
            t[1] += t1 * mult
    for k, (t0, t1) in tot.items():
        print('TOTAL %-6s direct %8.3f ms | winograd %8.3f ms (%.2fx)' % (k, t0, t1, t0 / t1))


if __name__ == '__main__':
    main()
