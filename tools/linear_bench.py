#!/usr/bin/env python3
"""TFLOP/s of the masked linear kernels on the VGG16 classifier shapes (batch 256) through the C ABI;
--ab like conv_bench.py (e.g. --ab CPG_DISABLE_PW_GEMM=-,1)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpg_amd import _lib  # noqa: E402


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
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--ab', default='')
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--pm', action='store_true', help='with a piggymask (task >= 2)')
    a = ap.parse_args()
    L, dev, st, P = _lib.lib(), 'cuda:0', _lib.stream_ptr(), _lib.dptr
    for fin, fout in ((25088, 4096), (4096, 4096)):
        x = torch.randn(a.batch, fin, device=dev)
        w = torch.randn(fout, fin, device=dev) * 0.01
        y = torch.empty(a.batch, fout, device=dev)
        gy = torch.randn(a.batch, fout, device=dev)
        gx, gw = torch.empty_like(x), torch.empty_like(w)
        ws, nb = _lib.workspace(max(L.cpg_linear_workspace_bytes(a.batch, fin, fout), 1 << 28), dev)     # (room for the split counts an --ab switch selects)
        flops = 2.0 * a.batch * fin * fout
        pm = torch.rand(fout, fin, device=dev) * 0.012 if a.pm else None
        gpm = torch.empty_like(w) if a.pm else None
        Pm, Pg = (P(pm), P(gpm)) if a.pm else (None, None)
        wbytes = fin * fout * 4 * (2 if a.pm else 1)          # the bytes a weight-streaming pass must move (W, + the piggymask)
        runs = {'fwd': lambda: L.cpg_linear_fwd(P(x), P(w), Pm, 5e-3, None, P(y), a.batch, fin, fout, P(ws), nb, st),
                'dgrad': lambda: L.cpg_linear_dgrad(P(gy), P(w), Pm, 5e-3, P(gx), a.batch, fin, fout, P(ws), nb, st),
                'wgrad': lambda: L.cpg_linear_wgrad(P(x), P(gy), P(w), Pm, 5e-3, P(gw), Pg, None, a.batch, fin, fout, P(ws), nb, st)}
        for k, fn in runs.items():
            if a.ab:
                var, vals = a.ab.split('=')
                out = []
                for v in vals.split(','):
                    ts = []
                    for _ in range(a.reps):
                        _lib.set_option(var, None if v == '-' else v)
                        ts.append(timeit(fn, a.iters))
                    _lib.set_option(var, None)
                    t = sorted(ts)[len(ts) // 2]
                    out.append('%s=%s: %.3f ms %.1f TF %.2f TB/s' % (var, v, t, flops / t / 1e9, wbytes / t / 1e9))
                print('%5d->%-5d %-6s %s' % (fin, fout, k, '  '.join(out)), flush=True)
            else:
                t = timeit(fn, a.iters)
                print('%5d->%-5d %-6s %.3f ms %.1f TF %.2f TB/s' % (fin, fout, k, t, flops / t / 1e9, wbytes / t / 1e9), flush=True)


if __name__ == '__main__':
    main()
