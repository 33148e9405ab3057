#!/usr/bin/env python3
"""Per-call wall time of every Manager.train step / validate of bench.py's cycle (synchronised: a diagnosis of one-off costs inside
the timed region -- first launches, allocations -- not a throughput figure).  usage: python tools/step_times.py --arch A --steps K"""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cpg_amd.utils import manager as M

ap = argparse.ArgumentParser()
ap.add_argument('--arch', default='vgg16')
ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--warmup', type=int, default=5)
a = ap.parse_args()
log = []
orig_train = M.Manager.train


def train(self, opt, epoch, lrs, step):
    acc = None
    loader = list(self.train_loader)
    for item in loader:
        self.train_loader = [item]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        acc, step = orig_train(self, opt, epoch, lrs, step)
        torch.cuda.synchronize(); log.append(('train:' + self.args.mode, 1000 * (time.perf_counter() - t0)))
    return acc, step


import gc
_gc_t = [0.0]


def on_gc(phase, info):
    if phase == 'start':
        _gc_t[0] = time.perf_counter()
    elif info['generation'] == 2 or time.perf_counter() - _gc_t[0] > 2e-3:
        log.append(('gc%d' % info['generation'], 1000 * (time.perf_counter() - _gc_t[0])))


gc.callbacks.append(on_gc)
M.Manager.train = train
bench.Manager = M.Manager
orig_bval = bench.validate


def bval(mgr, epoch):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = orig_bval(mgr, epoch)
    torch.cuda.synchronize(); log.append(('validate', 1000 * (time.perf_counter() - t0)))
    return r


bench.validate = bval
sys.argv = ['bench.py', '--arch', a.arch, '--steps', str(a.steps), '--warmup', str(a.warmup), '--no-cpu-baseline']
bench.main()
print(' '.join('%s=%.1f' % (k.replace('train:', ''), v) for k, v in log))
