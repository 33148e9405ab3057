#!/usr/bin/env python3
"""Per-wave phase times of k_wg1 (needs a library built with -DWG_TIMING: CPG_HIP_LIB=...): entry -> prologue start -> main loop
-> epilogue -> exit in 10 ns ticks of the constant clock, and the gap between a wave's exit and the entry of the next wave on
the same SIMD (development tool)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cpg_amd import _lib
from cpg_amd.models.layers import _conv_desc

def main():
    wgrad = '--wgrad' in sys.argv
    stats = '--stats' in sys.argv
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    N, C, K, H = [int(v) for v in (args[0] if args else '16,64,64,224').split(',')]
    L = _lib.lib(); raw = ctypes.CDLL(_lib.LIB_PATH)
    dev = 'cuda:0'
    x = torch.randn(N, C, H, H, device=dev).relu_(); w = torch.randn(K, C, 3, 3, device=dev) * 0.05
    y = torch.empty(N, K, H, H, device=dev)
    d = _conv_desc(x.shape, w.shape, (1, 1), (1, 1), (1, 1), 1)
    ws, nb = _lib.workspace(L.cpg_conv2d_workspace_bytes(ctypes.byref(d)), dev)
    if wgrad:
        gy = torch.randn(N, K, H, H, device=dev); gw = torch.empty_like(w)
        for _ in range(2):
            rc = L.cpg_conv2d_wgrad(ctypes.byref(d), _lib.dptr(x), _lib.dptr(gy), _lib.dptr(w), None, ctypes.c_float(0), _lib.dptr(gw), None, None,
                                    _lib.dptr(ws), nb, _lib.stream_ptr())
            assert rc == 0, L.cpg_last_error()
    if stats:
        import cpg_amd.models.layers as nl
        for _ in range(2):
            nl._MaskedConv2dFn.apply(x, w, None, None, 5e-3, (1, 1), (1, 1), (1, 1), 1, True)
    for _ in range(0 if (wgrad or stats) else 2):
        rc = L.cpg_conv2d_fwd(ctypes.byref(d), _lib.dptr(x), _lib.dptr(w), None, ctypes.c_float(0), None, _lib.dptr(y), _lib.dptr(ws), nb, _lib.stream_ptr())
        assert rc == 0, L.cpg_last_error()
    torch.cuda.synchronize()
    units = 65536 if wgrad else min(65536, N * (H // 2) * (H // 2) // 32 * ((K + 31) // 32) * 2)
    buf = np.zeros((units, 8), dtype=np.uint64)
    assert (raw.cpg_debug_ww_timing if wgrad else raw.cpg_debug_wg_timing)(buf.ctypes.data_as(ctypes.c_void_p), units) == 0
    t = buf[:, :5].astype(np.int64)
    ok = t[:, 4] > 0
    t = t[ok]; hw = buf[ok, 6]; xcc = buf[ok, 7]
    ph = np.diff(t[:, [0, 1, 2, 3, 4]], axis=1) * 0.01
    print('units %d   phases (us, mean / median): setup %.2f / %.2f   prologue %.2f / %.2f   main loop %.2f / %.2f   epilogue %.2f / %.2f   total %.2f'
          % (len(t), ph[:, 0].mean(), np.median(ph[:, 0]), ph[:, 1].mean(), np.median(ph[:, 1]), ph[:, 2].mean(), np.median(ph[:, 2]),
             ph[:, 3].mean(), np.median(ph[:, 3]), (t[:, 4] - t[:, 0]).mean() * 0.01))
    # slot = (xcc, se, sh?, cu, simd): HW_ID bits: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (gfx9)
    slot = (xcc.astype(np.int64) & 0xf) << 32 | (hw.astype(np.int64) & 0xfff0)
    gaps = []
    for sl in np.unique(slot):
        sel = np.where(slot == sl)[0]
        o = sel[np.argsort(t[sel, 0])]
        gaps.extend(((t[o[1:], 0] - t[o[:-1], 4]) * 0.01).tolist())
    gaps = np.array(gaps)
    print('slots %d   exit -> next entry on the same SIMD (us): mean %.2f median %.2f p90 %.2f   span of the launch %.1f us'
          % (len(np.unique(slot)), gaps.mean(), np.median(gaps), np.percentile(gaps, 90), (t[:, 4].max() - t[:, 0].min()) * 0.01))

main()
