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
N, C, K, H = 1, 32, 32, 28
bad = {}
for py in range(0, 28):
    for px in range(0, 28):
        x = torch.zeros(N, C, H, H, device=dev); gy = torch.ones(N, K, H, H, device=dev)
        x[0, :, py, px] = torch.arange(1, C + 1, device=dev).float()
        w = torch.zeros(K, C, 3, 3, device=dev); gw = torch.full_like(w, float('nan'))
        d = _conv_desc(x.shape, w.shape, (1, 1), (1, 1), (1, 1), 1)
        nb = raw.cpg_conv3x3_wino_wgrad_workspace(ctypes.byref(d)); ws = torch.empty(nb // 4 + 64, device=dev)
        raw.cpg_conv3x3_wino_wgrad(ctypes.byref(d), cp(x.data_ptr()), cp(gy.data_ptr()), cp(w.data_ptr()), None, 5e-3, cp(gw.data_ptr()), None, cp(ws.data_ptr()), nb, st)
        torch.cuda.synchronize()
        ref = torch.nn.grad.conv2d_weight(x.double().cpu(), w.shape, gy.double().cpu(), padding=1)
        e = (gw.double().cpu() - ref).abs()
        cols = e.amax((0, 2, 3)).gt(1e-3).nonzero().flatten().tolist()
        if cols:
            bad[(py, px)] = cols
print('x pixels with a wrong result (all-ones gy): %d of 784' % len(bad))
for k in sorted(bad)[:60]:
    print(k, 'bad input channels', bad[k])
