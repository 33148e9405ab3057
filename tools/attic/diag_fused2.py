"""(direct, unfused) vs (winograd, unfused): where in the backward pass do the gradients first differ?"""
import os, sys, torch, numpy as np, torch.nn as nn
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from tests.test_hip_parity import build, DEV
from cpg_amd.models.fused_bn import FusedSequential
torch.manual_seed(0)
net = build('vgg_cifar100', 0.25).to(DEV)
g = torch.Generator().manual_seed(4)
x = torch.randn(16, 3, 32, 32, generator=g).to(DEV)
t = torch.randint(0, 5, (16,), generator=g).to(DEV)
sd = {k: v.clone() for k, v in net.state_dict().items()}
FusedSequential.fuse = False
cap = {}
def mk(name, store):
    def hook(mod, gin, gout):
        store[name] = (None if gout[0] is None else gout[0].detach().clone(), None if gin[0] is None else gin[0].detach().clone())
    return hook
def fhook(name, store):
    def hook(mod, inp, out):
        store['F' + name] = out.detach().clone()
        if out.requires_grad:
            out.register_hook(lambda gr: store.__setitem__(name, (gr.detach().clone(), None)))
    return hook
for algo in ('direct', 'winograd'):
    from cpg_amd import _lib as _l
    _l.set_option('CPG_NO_WINO', 1 if algo == 'direct' else None)
    store = cap.setdefault(algo, {})
    hs = []
    for n, m in net.features.named_children():
        hs.append(m.register_forward_hook(fhook(n, store)))
    net.load_state_dict(sd); net.zero_grad(); net.train()
    out = net(x)
    nn.functional.cross_entropy(out, t).backward()
    for h in hs: h.remove()
a, b = cap['direct'], cap['winograd']
def rel(u, v):
    if u is None or v is None: return float('nan')
    return float((u - v).abs().max() / (u.abs().max() + 1e-30))
for n, m in reversed(list(net.features.named_children())):
    print('%-4s %-18s fwd out %.2e   grad_out %.2e   grad_in %.2e' % (n, type(m).__name__, rel(a['F' + n], b['F' + n]), rel(a.get(n, (None,))[0], b.get(n, (None,))[0]), float('nan')))
