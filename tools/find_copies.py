#!/usr/bin/env python3
"""Which Python lines issue device-to-device copies in a training step?  (rocprofv3 shows __amd_rocclr_copyBuffer launches; this names
their callers.)  ARCH=vgg16|resnet50|spherenet20 FUSED=1 python tools/find_copies.py"""
import os
import sys
import collections
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                             # noqa: E402
from cpg_amd.models import layers as nl                  # noqa: E402
from cpg_amd.utils.manager import Manager                # noqa: E402
from cpg_amd import dist as cdist                        # noqa: E402

dev = torch.device('cuda', 0)
ARCH = os.environ.get('ARCH', 'vgg16')
bench.DATASET = bench.ARCHS[ARCH]['dataset']
model = cdist.DataParallel(bench.build_model(dev, ARCH))
masks = {n: torch.zeros(m.weight.shape, dtype=torch.uint8, device=dev) for n, m in model.named_modules()
         if isinstance(m, (nl.SharableConv2d, nl.SharableLinear))}
g = torch.Generator(device=dev).manual_seed(1)
B = int(os.environ.get('B', 64))
SZ, NCLS = bench.ARCHS[ARCH]['size'], bench.ARCHS[ARCH]['classes']
pool = [(torch.randn(B, 3, SZ, SZ, generator=g, device=dev), torch.randint(0, NCLS, (B,), generator=g, device=dev)) for _ in range(2)]
mgr = Manager(bench.make_args('finetune', 1), model, {}, masks, None, pool, 0, 0)
mgr.pruner.make_finetuning_mask()
opt = bench.make_optimizers(model, mgr.pruner, 1e-3, None)


def steps(n):
    mgr.train_loader = [pool[i % 2] for i in range(n)]
    mgr.train(opt, 0, [1e-3], 0)


steps(3)
torch.cuda.synchronize()
N = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    steps(N)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ('aten::copy_', 'aten::clone', 'aten::contiguous', 'aten::fill_', 'aten::zero_') and ev.device_time_total > 0:
        stack = [f for f in (ev.stack or []) if '/repo/' in f and 'find_copies' not in f][:3]
        cnt[(ev.name, ' <- '.join(s.split('/repo/')[-1] for s in stack))] += 1
for (name, where), n in cnt.most_common(40):
    print('%5.1f / step  %-16s %s' % (n / N, name, where))
