"""per-tap error of the Winograd weight gradient against fp64 on a small layer"""
import ctypes, os, sys, torch
sys.path.insert(0, '.')
from cpg_amd import _lib
from cpg_amd._lib import ConvDesc
from cpg_amd.models.layers import _conv_desc
L = _lib.lib(); raw = ctypes.CDLL(_lib.LIB_PATH)
raw.cpg_conv3x3_wino_wgrad_workspace.restype = ctypes.c_size_t
raw.cpg_conv3x3_wino_wgrad_workspace.argtypes = [ctypes.POINTER(ConvDesc)]
raw.cpg_conv3x3_wino_wgrad.argtypes = [ctypes.POINTER(ConvDesc)] + [ctypes.c_void_p] * 4 + [ctypes.c_float] + [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_void_p]
dev = 'cuda:0'; st = _lib.stream_ptr(); cp = ctypes.c_void_p
for (N, C, K, H, mode) in [(1, 32, 32, 28, 'rand'), (1, 32, 32, 28, 'center'), (2, 32, 32, 56, 'rand'), (3, 64, 32, 28, 'rand')]:
    torch.manual_seed(1)
    x = torch.randn(N, C, H, H, device=dev); gy = torch.randn(N, K, H, H, device=dev)
    if mode == 'center':
        x.zero_(); gy.zero_(); x[0, :, 10:14, 10:14] = torch.randn(C, 4, 4, device=dev); gy[0, :, 11:13, 11:13] = torch.randn(K, 2, 2, device=dev)
    w = torch.zeros(K, C, 3, 3, device=dev)
    gw = torch.full_like(w, float('nan'))
    d = _conv_desc(x.shape, w.shape, (1, 1), (1, 1), (1, 1), 1)
    nb = raw.cpg_conv3x3_wino_wgrad_workspace(ctypes.byref(d)); ws = torch.empty(nb // 4 + 64, device=dev)
    rc = raw.cpg_conv3x3_wino_wgrad(ctypes.byref(d), cp(x.data_ptr()), cp(gy.data_ptr()), cp(w.data_ptr()), None, 5e-3, cp(gw.data_ptr()), None, cp(ws.data_ptr()), nb, st)
    torch.cuda.synchronize()
    ref = torch.nn.grad.conv2d_weight(x.double().cpu(), w.shape, gy.double().cpu(), padding=1)
    e = (gw.double().cpu() - ref).abs()
    print((N, C, K, H, mode), 'rc', rc, 'max|ref| %.3f' % ref.abs().max().item())
    print('  per-tap max err:', [['%.2e' % e[:, :, r, s].max().item() for s in range(3)] for r in range(3)])
    print('  per-tap ratio got/ref (median):', [['%.3f' % (gw.double().cpu()[:, :, r, s] / ref[:, :, r, s]).median().item() for s in range(3)] for r in range(3)])
print('---- single pixels: gy[0, k, gy_y, gy_x] = 1 for all k; x[0, c, px_y, px_x] = c + 1')
for (gyp, xp) in [((12, 12), (11, 11)), ((12, 12), (11, 12)), ((12, 12), (11, 13)), ((12, 12), (12, 11)), ((13, 13), (12, 12)), ((12, 13), (11, 12)), ((12, 12), (13, 13)), ((0, 0), (0, 0)), ((0, 1), (0, 0))]:
    N, C, K, H = 1, 32, 32, 28
    x = torch.zeros(N, C, H, H, device=dev); gy = torch.zeros(N, K, H, H, device=dev)
    gy[0, :, gyp[0], gyp[1]] = 1.0
    x[0, :, xp[0], xp[1]] = torch.arange(1, C + 1, device=dev).float()
    w = torch.zeros(K, C, 3, 3, device=dev); gw = torch.full_like(w, float('nan'))
    d = _conv_desc(x.shape, w.shape, (1, 1), (1, 1), (1, 1), 1)
    nb = raw.cpg_conv3x3_wino_wgrad_workspace(ctypes.byref(d)); ws = torch.empty(nb // 4 + 64, device=dev)
    raw.cpg_conv3x3_wino_wgrad(ctypes.byref(d), cp(x.data_ptr()), cp(gy.data_ptr()), cp(w.data_ptr()), None, 5e-3, cp(gw.data_ptr()), None, cp(ws.data_ptr()), nb, st)
    torch.cuda.synchronize()
    ref = torch.nn.grad.conv2d_weight(x.double().cpu(), w.shape, gy.double().cpu(), padding=1)
    g = gw.double().cpu()
    print('gy', gyp, 'x', xp, 'expected tap', (xp[0] - gyp[0] + 1, xp[1] - gyp[1] + 1), ' got[k=0, c=4] taps:', [[round(g[0, 4, r, s].item(), 3) for s in range(3)] for r in range(3)],
          ' max err', '%.2e' % (g - ref).abs().max().item(), ' bad k rows', sorted(set((g - ref).abs().amax((1, 2, 3)).gt(1e-3).nonzero().flatten().tolist()))[:8],
          ' bad c cols', sorted(set((g - ref).abs().amax((0, 2, 3)).gt(1e-3).nonzero().flatten().tolist()))[:8])
