#!/usr/bin/env python3
"""Per-layer microbenchmark of the masked conv / linear kernels through the C ABI (HIP events on the
launch stream).  Prints TFLOP/s per (layer, pass).  Used for A/B work on kernel configurations and as
the command profiled with rocprofv3 --pmc.

    python tools/conv_bench.py [--batch 256] [--iters 5] [--only fwd,dgrad,wgrad] [--layers 3,7]
"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpg_amd import _lib                      # noqa: E402
from cpg_amd.models.layers import _conv_desc  # noqa: E402

# (name, Cin, Cout, H, multiplicity) of VGG16 @224 -- SURVEY.md section 8 table (f17/f20, f27/f30, f34/f37/f40 share shapes)
VGG = [('f0', 3, 64, 224, 1), ('f3', 64, 64, 224, 1), ('f7', 64, 128, 112, 1), ('f10', 128, 128, 112, 1), ('f14', 128, 256, 56, 1),
       ('f17', 256, 256, 56, 2), ('f24', 256, 512, 28, 1), ('f27', 512, 512, 28, 2), ('f34', 512, 512, 14, 3)]


def timeit(fn, iters):
    rc = fn()
    if rc:      # (an --ab value that changes the launch plan may outgrow the workspace sized for the default plan: say so, do not time the refusal)
        from cpg_amd import _lib
        raise RuntimeError('launch refused (rc %d): %s' % (rc, _lib.lib().cpg_last_error().decode() if hasattr(_lib.lib(), 'cpg_last_error') else ''))
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
    ap.add_argument('--only', default='fwd,dgrad,wgrad')
    ap.add_argument('--layers', default='')
    ap.add_argument('--pm', action='store_true')
    ap.add_argument('--shape', default='', help='extra 3x3 layer "C,K,H" (name x0) instead of the VGG list')
    ap.add_argument('--ab', default='', help='A/B inside one process: NAME=v1,v2,... toggles that env var between timed runs (median of --reps rounds)')
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--width-multiplier', type=float, default=1.0,
                    help="RAW width multiplier of the reference's command line: the VGG list with int(v * sqrt(m)) channels (1.5: 78 / 156 / 313 / 627)")
    ap.add_argument('--pmc-pass', action='store_true', help='no timing: launch every conv of one VGG16 pass exactly once (for rocprofv3 --pmc)')
    a = ap.parse_args()
    L = _lib.lib()
    dev = 'cuda:0'
    st = _lib.stream_ptr()
    sel = set(a.layers.split(',')) if a.layers else None
    print('%-6s %-6s %9s %9s' % ('layer', 'pass', 'ms', 'TFLOP/s'))
    tot = {}
    layers = VGG
    if a.width_multiplier != 1.0:
        r = a.width_multiplier ** 0.5
        layers = [(n, c if c == 3 else int(c * r), int(k * r), h, m) for n, c, k, h, m in VGG]
    if a.shape:
        c, k, h = (int(v) for v in a.shape.split(','))
        layers = [('x0', c, k, h, 1)]
    for name, C, K, H, mult in layers:
        if sel and name not in sel:
            continue
        x = torch.randn(a.batch, C, H, H, device=dev)
        w = torch.randn(K, C, 3, 3, device=dev) * 0.05
        pm = torch.rand(K, C, 3, 3, device=dev) * 0.012 if a.pm else None
        y = torch.empty(a.batch, K, H, H, device=dev)
        gy = torch.randn(a.batch, K, H, H, device=dev)
        gx = torch.empty_like(x)
        gw = torch.empty_like(w)
        gpm = torch.empty_like(w) if a.pm else None
        d = _conv_desc(x.shape, w.shape, (1, 1), (1, 1), (1, 1), 1)
        nws = L.cpg_conv2d_workspace_bytes(ctypes.byref(d))
        ws, nb = _lib.workspace(nws, dev)
        ws16, nb16 = _lib.workspace(L.cpg_conv2d_bf16_workspace_bytes(ctypes.byref(d)), dev)       # opt-in bf16 path (fwd16 / dgrad16)
        wsw16, nbw16 = _lib.workspace(L.cpg_conv2d_wgrad_bf16_workspace_bytes(ctypes.byref(d)), dev)
        flops = 2.0 * a.batch * K * H * H * C * 9
        P = _lib.dptr
        tiles = L.cpg_conv2d_bnstats_tiles(ctypes.byref(d))
        stats = torch.empty(max(1, K * tiles * 2) * 4, device=dev)       # (x 4: an --ab over kernel variants may change the tile count)
        runs = {'fwdstats': lambda: L.cpg_conv2d_fwd_bnstats(ctypes.byref(d), P(x), P(w), P(pm), 5e-3, None, P(y), P(stats), stats.numel() * 4, P(ws), nb, st),
                'fwd': lambda: L.cpg_conv2d_fwd(ctypes.byref(d), P(x), P(w), P(pm), 5e-3, None, P(y), P(ws), nb, st),
                'dgrad': lambda: L.cpg_conv2d_dgrad(ctypes.byref(d), P(gy), P(w), P(pm), 5e-3, P(gx), P(ws), nb, st),
                'fwd16': lambda: L.cpg_conv2d_fwd_bf16(ctypes.byref(d), P(x), P(w), P(pm), 5e-3, None, P(y), P(ws16), nb16, st),
                'dgrad16': lambda: L.cpg_conv2d_dgrad_bf16(ctypes.byref(d), P(gy), P(w), P(pm), 5e-3, P(gx), P(ws16), nb16, st),
                'fwdx3': lambda: L.cpg_conv2d_fwd_bf16x3(ctypes.byref(d), P(x), P(w), P(pm), 5e-3, None, P(y), P(ws16), nb16, st),
                'dgradx3': lambda: L.cpg_conv2d_dgrad_bf16x3(ctypes.byref(d), P(gy), P(w), P(pm), 5e-3, P(gx), P(ws16), nb16, st),
                'wgradx3': lambda: L.cpg_conv2d_wgrad_bf16x3(ctypes.byref(d), P(x), P(gy), P(w), P(pm), 5e-3, P(gw), P(gpm), P(wsw16), nbw16, st),
                'wgrad16': lambda: L.cpg_conv2d_wgrad_bf16(ctypes.byref(d), P(x), P(gy), P(w), P(pm), 5e-3, P(gw), P(gpm), P(wsw16), nbw16, st),
                'wgrad': lambda: L.cpg_conv2d_wgrad(ctypes.byref(d), P(x), P(gy), P(w), P(pm), 5e-3, P(gw), P(gpm), None, P(ws), nb, st)}
        for k in a.only.split(','):
            if k in ('dgrad', 'dgrad16', 'dgradx3') and name == 'f0':
                continue
            if k in ('wgrad16', 'wgradx3') and not L.cpg_conv2d_wgrad_bf16_supported(ctypes.byref(d)):
                continue
            if k in ('fwd16', 'dgrad16', 'fwdx3', 'dgradx3') and not L.cpg_conv2d_bf16_supported(ctypes.byref(d)):
                continue
            if a.pmc_pass:
                for _ in range(mult):
                    runs[k]()
                torch.cuda.synchronize()
                continue
            if a.ab:
                var, vals = a.ab.split('=')
                vals = vals.split(',')
                res = {v: [] for v in vals}
                for v in vals:          # a value may change the launch plan (units per wave slot, ...): size the workspace for the largest
                    _lib.set_option(var, None if v == '-' else v)
                    need = L.cpg_conv2d_workspace_bytes(ctypes.byref(d))
                    if need > nb:
                        ws, nb = _lib.workspace(need, dev)
                for _ in range(a.reps):
                    for v in vals:
                        _lib.set_option(var, None if v == '-' else v)      # (the library reads the environment only when it is loaded)
                        res[v].append(timeit(runs[k], a.iters))
                _lib.set_option(var, None)
                print('%-6s %-6s ' % (name, k) + '  '.join('%s=%s: %.3f ms %.1f TF' % (var, v, sorted(t)[len(t) // 2], flops / sorted(t)[len(t) // 2] / 1e9)
                                                          for v, t in res.items()), flush=True)
                continue
            ms = timeit(runs[k], a.iters)
            print('%-6s %-6s %9.3f %9.1f' % (name, k, ms, flops / ms / 1e9), flush=True)
            t = tot.setdefault(k, [0.0, 0.0])
            t[0] += ms * mult
            t[1] += flops * mult
    for k, (ms, fl) in tot.items():
        print('TOTAL  %-6s %9.3f %9.1f   (13 convs of one VGG16 pass)' % (k, ms, fl / ms / 1e9))


if __name__ == '__main__':
    main()
