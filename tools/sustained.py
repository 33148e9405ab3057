#!/usr/bin/env python3
"""Does the chip hold its clock under sustained MFMA load?  Times the same conv call in windows of
~0.25 s for several seconds, optionally interleaved with large elementwise passes."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpg_amd import _lib
from cpg_amd.models.layers import _conv_desc
L = _lib.lib(); dev = 'cuda:0'; st = _lib.stream_ptr(); P = _lib.dptr
B, C, K, H = 256, 256, 256, 56
x = torch.randn(B, C, H, H, device=dev); w = torch.randn(K, C, 3, 3, device=dev) * 0.05
y = torch.empty(B, K, H, H, device=dev)
d = _conv_desc(x.shape, w.shape, (1, 1), (1, 1), (1, 1), 1)
ws, nb = _lib.workspace(L.cpg_conv2d_workspace_bytes(ctypes.byref(d)), dev)
flops = 2.0 * B * K * H * H * C * 9
big = torch.randn(256, 64, 224, 224, device=dev)
def window(n, mix):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for _ in range(n):
        s.record()
        L.cpg_conv2d_fwd(ctypes.byref(d), P(x), P(w), None, 5e-3, None, P(y), P(ws), nb, st)
        e.record()
        if mix:
            big.mul_(1.0001)
        torch.cuda.synchronize()
        tot += s.elapsed_time(e)
    return flops * n / tot / 1e9
for mix in (False, True):
    t0 = time.time()
    out = []
    while time.time() - t0 < 6.0:
        out.append(window(30, mix))
    print('mix' if mix else 'pure', ' '.join('%.0f' % v for v in out))
os.system('rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | head -6')
