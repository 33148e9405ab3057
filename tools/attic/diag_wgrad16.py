#!/usr/bin/env python3
"""Diagnostic for cpg_conv2d_wgrad_bf16: structured inputs whose weight gradient can be read off by eye."""
import ctypes
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cpg_amd import _lib                      # noqa: E402
from cpg_amd.models.layers import _conv_desc  # noqa: E402

L = _lib.lib()
dev = 'cuda:0'
P = _lib.dptr


def run(x, gy):
    N, C, H, W = x.shape
    K = gy.shape[1]
    w = torch.zeros(K, C, 3, 3, device=dev)
    gw = torch.empty_like(w)
    d = _conv_desc(x.shape, w.shape, (1, 1), (1, 1), (1, 1), 1)
    assert L.cpg_conv2d_wgrad_bf16_supported(ctypes.byref(d))
    ws, nb = _lib.workspace(L.cpg_conv2d_wgrad_bf16_workspace_bytes(ctypes.byref(d)), dev)
    rc = L.cpg_conv2d_wgrad_bf16(ctypes.byref(d), P(x), P(gy), P(w), None, 5e-3, P(gw), None, P(ws), nb, _lib.stream_ptr())
    _lib.check('wgrad16', rc)
    w64 = torch.zeros(K, C, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.cpu().bfloat16().double(), w64, None, padding=1).backward(gy.cpu().bfloat16().double())
    return gw.cpu().double(), w64.grad


def show(tag, got, want, co=0, ci=0):
    print(tag, 'maxdiff/scale %.3g' % float((got - want).abs().max() / want.abs().max()))
    print('  got ', [round(v, 2) for v in got[co, ci].flatten().tolist()])
    print('  want', [round(v, 2) for v in want[co, ci].flatten().tolist()])


N, C, K, H, W = 1, 16, 16, 2, 64
x = torch.ones(N, C, H, W, device=dev)
gy = torch.ones(N, K, H, W, device=dev)
show('ones', *run(x, gy))
x = torch.arange(W, device=dev, dtype=torch.float32).view(1, 1, 1, W).expand(N, C, H, W).contiguous()
gy = torch.zeros(N, K, H, W, device=dev)
gy[:, :, 0, 10] = 1.0
show('ramp x, delta gy at (0,10)', *run(x, gy))
gy = torch.zeros(N, K, H, W, device=dev)
gy[:, :, 1, 37] = 1.0
x = (torch.arange(H * W, device=dev, dtype=torch.float32).view(1, 1, H, W) % 200).expand(N, C, H, W).contiguous()
show('ramp x (h*W+w), delta gy at (1,37)', *run(x, gy))
x = torch.zeros(N, C, H, W, device=dev)
x[:, 3] = 1.0
gy = torch.zeros(N, K, H, W, device=dev)
gy[:, 5] = 1.0
g, w_ = run(x, gy)
print('channel selectivity: nonzero (co, ci) pairs got', sorted(set((int(a), int(b)) for a, b in (g.abs().sum((2, 3)) > 0).nonzero().tolist()))[:8],
      'want', sorted(set((int(a), int(b)) for a, b in (w_.abs().sum((2, 3)) > 0).nonzero().tolist())))
torch.manual_seed(0)
x = torch.randn(2, 32, 4, 64, device=dev)
gy = torch.randn(2, 32, 4, 64, device=dev)
show('random 2x32x4x64', *run(x, gy), co=3, ci=7)
