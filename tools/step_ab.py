#!/usr/bin/env python3
"""In-process A/B of a class-level switch on the VGG16 224x224 batch-256 forward + backward (no optimizer): alternates the
two settings in blocks of --iters steps, --reps times, and prints the median ms per step of each.

    python tools/step_ab.py cpg_amd.models.fused_bn.FusedSequential.fuse_stats
    python tools/step_ab.py env:CPG_NO_V14
"""
import argparse
import importlib
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cpg_amd.models as M  # noqa: E402

VGG_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('switch', help='dotted path of a boolean class / module attribute')
    ap.add_argument('--iters', type=int, default=6)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--batch', type=int, default=256)
    a = ap.parse_args()
    if a.switch.startswith('env:'):             # env:NAME -- True = variable set to "1", False = unset (read by the library per call)
        class _Env(object):
            def __setattr__(self, name, value):
                from cpg_amd import _lib
                _lib.set_option(name, 1 if value else None)      # (the library reads the environment only when it is loaded)
        owner, attr = _Env(), a.switch[4:]
    else:
        path, attr = a.switch.rsplit('.', 1)
        try:
            owner = importlib.import_module(path)
        except ImportError:
            mod, cls = path.rsplit('.', 1)
            owner = getattr(importlib.import_module(mod), cls)
    torch.manual_seed(1)
    net = M.custom_vgg(VGG_CFG, dataset_history=[], dataset2num_classes={}, network_width_multiplier=1.0, shared_layer_info={})
    net.add_dataset('t', 5)
    net.set_dataset('t')
    net = net.cuda().train()
    x = torch.randn(a.batch, 3, 224, 224, device='cuda')
    t = torch.randint(0, 5, (a.batch,), device='cuda')

    def step():
        net.zero_grad(set_to_none=True)
        F.cross_entropy(net(x), t).backward()
    res = {True: [], False: []}
    for v in (True, False):
        setattr(owner, attr, v)
        step()
    for _ in range(a.reps):
        for v in (True, False):
            setattr(owner, attr, v)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(a.iters):
                step()
            e.record()
            torch.cuda.synchronize()
            res[v].append(s.elapsed_time(e) / a.iters)
    for v in (True, False):
        r = sorted(res[v])
        print('%s = %s: median %.2f ms/step  (min %.2f, max %.2f)' % (a.switch, v, r[len(r) // 2], r[0], r[-1]))


if __name__ == '__main__':
    main()
