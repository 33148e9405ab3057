"""fused / unfused BatchNorm x direct / Winograd conv on the narrow CIFAR VGG: per-parameter gradient differences against the
(direct, unfused) run -- which of the four runs leaves the others, and where."""
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
res = {}
for algo in ('direct', 'winograd'):
    from cpg_amd import _lib as _l
    _l.set_option('CPG_NO_WINO', 1 if algo == 'direct' else None)
    for fuse in (False, True):
        net.load_state_dict(sd); net.zero_grad(); net.train()
        FusedSequential.fuse = fuse
        out = net(x)
        nn.functional.cross_entropy(out, t).backward()
        res[(algo, fuse)] = (out.detach().cpu().numpy(), {n: p.grad.cpu().numpy().copy() for n, p in net.named_parameters() if p.grad is not None})
FusedSequential.fuse = True
base = res[('direct', False)]
for key in res:
    print(key, 'logits maxdiff vs (direct, unfused)', np.abs(res[key][0] - base[0]).max())
names = list(base[1])
print('%-24s' % 'param' + ''.join('%22s' % str(k) for k in res))
for n in names:
    sc = np.abs(base[1][n]).max() + 1e-30
    print('%-24s' % n + ''.join('%22.3e' % (np.abs(res[k][1][n] - base[1][n]).max() / sc) for k in res))
