#!/usr/bin/env python3
"""Host-side cost of one training step: enqueue time vs GPU time, plus a cProfile of the step loop."""
import cProfile, os, pstats, sys, time, types, io
import torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cpg_amd.models import layers as nl
from cpg_amd.utils import Optimizers
from cpg_amd.utils.manager import Manager
dev = torch.device('cuda', 0)
ARCH = os.environ.get('ARCH', 'vgg16')
bench.DATASET = bench.ARCHS[ARCH]['dataset']
net = bench.build_model(dev, ARCH)
from cpg_amd import dist as cdist
model = cdist.DataParallel(net)
masks = {n: torch.zeros(m.weight.shape, dtype=torch.uint8, device=dev) for n, m in model.named_modules()
         if isinstance(m, (nl.SharableConv2d, nl.SharableLinear))}
g = torch.Generator(device=dev).manual_seed(1)
B = int(os.environ.get('B', 256))
SZ, NCLS = bench.ARCHS[ARCH]['size'], bench.ARCHS[ARCH]['classes']
pool = [(torch.randn(B, 3, SZ, SZ, generator=g, device=dev), torch.randint(0, NCLS, (B,), generator=g, device=dev)) for _ in range(2)]
mgr = Manager(bench.make_args('finetune', 1), model, {}, masks, None, pool, 0, 0)
mgr.pruner.make_finetuning_mask()
if os.environ.get('TASK2'):            # the cycle of tasks >= 2: piggymasks on every masked layer, MaskedSGD + MaskedAdam (bench.py --task 2)
    bench.begin_task2(model, masks, bench.ARCHS[ARCH], dev)
    pool = [(x, torch.randint(0, bench.ARCHS[ARCH]['classes2'], (B,), generator=g, device=dev)) for x, _ in pool]
    mgr = Manager(bench.make_args('finetune', 1), model, {}, masks, None, pool, 0, 0)
    mgr.pruner.make_finetuning_mask()                      # the free slots go to task 2
    if os.environ.get('MODE') == 'prune':                  # a prune run with an event EVERY step (the prune window of the K = 20 cycle)
        mgr = Manager(bench.make_args('prune', 1), model, {}, masks, None, pool, -20, 40)
    opt = bench.make_optimizers(model, mgr.pruner, 1e-3, 5e-4)
elif os.environ.get('FUSED'):
    opt = bench.make_optimizers(model, mgr.pruner, 1e-3, None)
else:
    opt = Optimizers(); opt.add(torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.9, nesterov=True), 1e-3)
def steps(n):
    mgr.train_loader = [pool[i % 2] for i in range(n)]
    mgr.train(opt, 0, [1e-3], 0)
steps(3); torch.cuda.synchronize()
# enqueue vs total
t0 = time.perf_counter(); steps(10); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('10 steps: python returned after %.1f ms/step, GPU done after %.1f ms/step' % ((t1 - t0) * 100, (t2 - t0) * 100))
pr = cProfile.Profile(); pr.enable(); steps(10); pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(35); print(s.getvalue()[:6000])
