"""Winograd forward / input-gradient launches with sentinel-filled guard regions around the output and the workspace."""
import ctypes, os, sys, torch
sys.path.insert(0, '.')
from cpg_amd import _lib
from cpg_amd.models.layers import _conv_desc
L = _lib.lib(); dev = 'cuda:0'; st = _lib.stream_ptr(); P = _lib.dptr
G = 1 << 16
for (N, C, K, H) in [(16, 16, 16, 32), (16, 16, 32, 16), (16, 32, 64, 8), (16, 64, 128, 4), (16, 128, 128, 2), (3, 16, 16, 2), (100, 64, 64, 14)]:
    x = torch.randn(N, C, H, H, device=dev); w = torch.randn(K, C, 3, 3, device=dev) * 0.1; gy = torch.randn(N, K, H, H, device=dev)
    d = _conv_desc(x.shape, w.shape, (1, 1), (1, 1), (1, 1), 1)
    nws = L.cpg_conv2d_workspace_bytes(ctypes.byref(d))
    for what in ('fwd', 'dgrad'):
        n_out = N * (K if what == 'fwd' else C) * H * H
        buf = torch.full((G + n_out + G,), 777.0, device=dev)
        out = buf[G:G + n_out]
        wsb = torch.full((G + nws // 4 + 4 + G,), 555.0, device=dev)
        ws = wsb[G:G + nws // 4 + 4]
        if what == 'fwd':
            rc = L.cpg_conv2d_fwd(ctypes.byref(d), P(x), P(w), None, 5e-3, None, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(ws.data_ptr()), nws, st)
        else:
            rc = L.cpg_conv2d_dgrad(ctypes.byref(d), P(gy), P(w), None, 5e-3, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(ws.data_ptr()), nws, st)
        torch.cuda.synchronize()
        ok = bool((buf[:G] == 777).all() and (buf[G + n_out:] == 777).all() and (wsb[:G] == 555).all() and (wsb[G + nws // 4 + 4:] == 555).all())
        print((N, C, K, H), what, 'rc', rc, 'guards intact', ok, 'untouched outputs', int((out == 777).sum()), 'ws bytes', nws)
