#!/usr/bin/env python3
"""Does a second read of a buffer hit the 256 MiB Infinity Cache?  Reads a buffer twice back to back (float4 copy-rate reduction
kernels of torch) after evicting everything with a 2 GiB sweep, for several buffer sizes: first-read vs second-read GB/s.
Behind the BatchNorm-backward channel-group experiment (docs/LAB_NOTEBOOK.md section 8)."""
import torch

dev = 'cuda:0'
big = torch.empty(512 * 1024 * 1024, dtype=torch.float32, device=dev)      # 2 GiB evictor


def timed(fn):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e)


for mb in (32, 64, 96, 128, 160, 192, 256, 384, 512, 1024):
    n = mb * 1024 * 1024 // 4
    x = torch.randn(n, device=dev)
    y = torch.empty_like(x)
    res = []
    for rep in range(3):
        big.fill_(1.0)                      # evict
        torch.cuda.synchronize()
        t1 = timed(lambda: x.sum())
        t2 = timed(lambda: x.sum())
        big.fill_(1.0)
        torch.cuda.synchronize()
        t3 = timed(lambda: y.copy_(x))      # read + write
        t4 = timed(lambda: x.sum())         # read after the copy wrote an equally large buffer
        res.append((t1, t2, t3, t4))
    t1, t2, t3, t4 = [min(r[i] for r in res) for i in range(4)]
    gb = mb / 1024.0
    print('%5d MB: first read %6.0f GB/s, second read %6.0f GB/s, copy %6.0f GB/s (r+w), read after copy %6.0f GB/s'
          % (mb, gb / t1 * 1e3, gb / t2 * 1e3, 2 * gb / t3 * 1e3, gb / t4 * 1e3), flush=True)
