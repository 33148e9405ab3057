#!/usr/bin/env python3
"""What a conv launch costs when another stream's kernels hold some CUs (as RCCL's do during the backward pass of a data-parallel
run): HOGS single-wave spin kernels (torch.cuda._sleep, one stream each) are parked on the chip, then a forward / weight-gradient
launch is timed with different grid policies (CPG_WINO_GRIDS, CPG_WW_UNITS, CPG_STEM_BLOCKS).  A wave of the Winograd kernels
needs a whole SIMD's registers, so every parked wave takes its CU out of the launch.

    python tools/attic/diag_interference.py [--hogs 16]
"""
import argparse, ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cpg_amd import _lib
from cpg_amd.models.layers import _conv_desc

def main():
    ap = argparse.ArgumentParser(); ap.add_argument('--hogs', type=int, default=16); a = ap.parse_args()
    L = _lib.lib(); dev = 'cuda:0'; P = _lib.dptr; st = _lib.stream_ptr()
    hogs = [torch.cuda.Stream() for _ in range(a.hogs)]
    def timed(fn, hog):
        fn(); torch.cuda.synchronize()
        if hog:
            for s in hogs:
                with torch.cuda.stream(s):
                    torch.cuda._sleep(int(60e6))         # ~ 25-30 ms
            time.sleep(0.002)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        t = e0.elapsed_time(e1)
        torch.cuda.synchronize()
        return t
    def layer(N, C, K, H):
        x = torch.randn(N, C, H, H, device=dev).relu_(); w = torch.randn(K, C, 3, 3, device=dev) * 0.05
        gy = torch.randn(N, K, H, H, device=dev); y = torch.empty(N, K, H, H, device=dev); gw = torch.empty_like(w)
        d = _conv_desc(x.shape, w.shape, (1, 1), (1, 1), (1, 1), 1)
        def fwd():
            ws, nb = _lib.workspace(L.cpg_conv2d_workspace_bytes(ctypes.byref(d)), dev)
            assert L.cpg_conv2d_fwd(ctypes.byref(d), P(x), P(w), None, 5e-3, None, P(y), P(ws), nb, st) == 0
        def wgrad():
            ws, nb = _lib.workspace(L.cpg_conv2d_workspace_bytes(ctypes.byref(d)), dev)
            assert L.cpg_conv2d_wgrad(ctypes.byref(d), P(x), P(gy), P(w), None, 5e-3, P(gw), None, None, P(ws), nb, st) == 0
        return fwd, wgrad
    cases = [('forward 512->512 @28 (k_wg3)', layer(256, 512, 512, 28)[0], 'CPG_WINO_GRIDS', ['1', '8']),
             ('forward 64->64 @224 (k_wg3)', layer(128, 64, 64, 224)[0], 'CPG_WINO_GRIDS', ['1', '8']),
             ('weight gradient 512->512 @28 (k_wgw)', layer(256, 512, 512, 28)[1], 'CPG_WW_UNITS', ['2', '4', '8']),
             ('stem forward 3->64 @224', layer(256, 3, 64, 224)[0], 'CPG_STEM_BLOCKS', ['512', '1024'])]
    for name, fn, var, vals in cases:
        for v in vals:
            _lib.set_option(var, v)
            free = min(timed(fn, False) for _ in range(3))
            busy = min(timed(fn, True) for _ in range(3))
            print('%-40s %s=%-5s idle chip %.3f ms   with %d parked waves %.3f ms  (x %.2f)' % (name, var, v, free, a.hogs, busy, busy / free), flush=True)
        _lib.set_option(var, None)

main()
