#!/usr/bin/env python3
"""Which host-side op is behind every device-to-device copy (and every tiny torch kernel) of ONE train step?

    python tools/trace_copies.py --arch spherenet20 [--batch 32]

torch.profiler with stacks around one Manager.train step of the bench's model; prints, per (op, innermost cpg_amd / bench frame), the
number of Memcpy DtoD / copy kernels it launched and their bytes where known.  rocprofv3 only names the kernel (`__amd_rocclr_copyBuffer`)."""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--arch', default='spherenet20')
    ap.add_argument('--batch', type=int, default=32)
    a = ap.parse_args()
    from cpg_amd import dist as cdist
    from cpg_amd.models import layers as nl
    from cpg_amd.utils.manager import Manager
    dev = torch.device('cuda', 0)
    A = bench.ARCHS[a.arch]
    bench.DATASET, bench.LRS = A['dataset'], A.get('lrs', bench.LRS)
    net = bench.build_model(dev, a.arch)
    model = cdist.DataParallel(net)
    masks = {n: torch.zeros(m.weight.shape, dtype=torch.uint8, device=dev) for n, m in model.named_modules()
             if isinstance(m, (nl.SharableConv2d, nl.SharableLinear))}
    g = torch.Generator(device=dev).manual_seed(1)
    pool = [(torch.randn(a.batch, 3, A['size'], A['size'], generator=g, device=dev), torch.randint(0, A['classes'], (a.batch,), generator=g, device=dev))
            for _ in range(2)]
    mgr = Manager(bench.make_args('finetune', 1), model, {}, masks, pool, None, 0, 0)
    mgr.pruner.make_finetuning_mask()
    opt = bench.make_optimizers(model, mgr.pruner, 1e-3)
    mgr.train(opt, 0, [1e-3], 0)                          # warm: allocator, code loading
    torch.cuda.synchronize()
    mgr.train_loader = pool[:1]
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        mgr.train(opt, 0, [1e-3], 0)
        torch.cuda.synchronize()
    by = collections.Counter()
    for ev in prof.events():
        name = ev.name
        if not (name.startswith('aten::copy_') or name.startswith('aten::clone') or name.startswith('aten::contiguous') or 'Memcpy' in name):
            continue
        if ev.device_type is not None and str(ev.device_type).endswith('CUDA'):
            continue
        frames = [f for f in (ev.stack or []) if 'cpg_amd' in f or 'bench.py' in f or 'autograd' in f]
        where = frames[0] if frames else ((ev.stack or ['?'])[0])
        shapes = str(ev.input_shapes)[:60]
        by[(name, where[-90:], shapes)] += 1
    for (name, where, shapes), n in sorted(by.items(), key=lambda kv: -kv[1])[:40]:
        print('%4d  %-18s %-62s %s' % (n, name, shapes, where))
    kern = collections.Counter()
    for ev in prof.events():
        if str(ev.device_type).endswith('CUDA') and ('copy' in ev.name.lower() or 'memcpy' in ev.name.lower()):
            kern[ev.name[:80]] += 1
    print('device-side copy events:', dict(kern))


if __name__ == '__main__':
    main()
