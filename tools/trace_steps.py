#!/usr/bin/env python3
"""N pure training steps (no validate, no prune) for kernel-trace analysis."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cpg_amd.models import layers as nl
from cpg_amd.utils import Optimizers
from cpg_amd.utils.manager import Manager
from cpg_amd import dist as cdist
dev = torch.device('cuda', 0)
model = cdist.DataParallel(bench.build_model(dev))
masks = {n: torch.zeros(m.weight.shape, dtype=torch.uint8, device=dev) for n, m in model.named_modules()
         if isinstance(m, (nl.SharableConv2d, nl.SharableLinear))}
g = torch.Generator(device=dev).manual_seed(1)
pool = [(torch.randn(256, 3, 224, 224, generator=g, device=dev), torch.randint(0, 5, (256,), generator=g, device=dev)) for _ in range(2)]
mgr = Manager(bench.make_args('finetune', 1), model, {}, masks, None, pool, 0, 0)
mgr.pruner.make_finetuning_mask()
opt = Optimizers(); opt.add(torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.9, nesterov=True), 1e-3)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
mgr.train_loader = [pool[i % 2] for i in range(n)]
torch.cuda.synchronize(); t0 = time.perf_counter()
mgr.train(opt, 0, [1e-3], 0)
torch.cuda.synchronize(); print('ms/step', (time.perf_counter() - t0) * 1000 / n)
