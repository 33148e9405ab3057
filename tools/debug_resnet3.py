import os, sys, torch, torch.nn as nn, torch.nn.functional as F, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cpg_amd.models as M
from cpg_amd.models import layers as nl
from cpg_amd.models.resnet import Bottleneck
DEV='cuda:0'
torch.manual_seed(1)
net = M.resnet50(dataset_history=[], dataset2num_classes={}, network_width_multiplier=0.25, shared_layer_info={})
net.add_dataset('t1',5); net.set_dataset('t1')
torch.manual_seed(2)
for m in net.modules():
    if isinstance(m, nl.SharableConv2d): nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
net = net.to(DEV).eval()
g = torch.Generator().manual_seed(9)
x = torch.randn(4,3,64,64,generator=g).to(DEV); t = torch.randint(0,5,(4,),generator=g).to(DEV)
rec = {}
def mk(name):
    def fh(mod, inp, out):
        rec[name] = {'x': inp[0].detach().clone()}
        out.register_hook(lambda gr: rec[name].__setitem__('gy', gr.detach().clone()))
    return fh
blocks = [(n, m) for n, m in net.named_modules() if isinstance(m, Bottleneck)]
for n, m in blocks: m.register_forward_hook(mk(n))
net(x).pipe = None if False else None
out = net(x); F.cross_entropy(out, t).backward()
hip_fwd = nl.SharableConv2d.forward
torch_fwd = lambda self, input, layer_info=None, name=None: F.conv2d(input, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
rel = lambda a,b: float((a-b).abs().max()/(b.abs().max()+1e-30))
def run_block(m, X, G, scale=1.0):
    m.zero_grad()
    xi = (X * scale).clone().requires_grad_(True)
    y = m(xi); y.backward(G)
    return y.detach().clone(), xi.grad.clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
for n, m in blocks:
    X, G = rec[n]['x'], rec[n]['gy']
    nl.SharableConv2d.forward = hip_fwd
    y1, gx1, gp1 = run_block(m, X, G)
    y1b, gx1b, gp1b = run_block(m, X, G, 1 + 1e-6)
    nl.SharableConv2d.forward = torch_fwd
    y2, gx2, gp2 = run_block(m, X, G)
    worst = max((rel(gp1[k], gp2[k]), k) for k in gp1)
    sens = max((rel(gp1b[k], gp1[k]), k) for k in gp1)
    print('%-10s y %.1e gx %.1e worst-param %.1e (%s) | HIP self-sensitivity gx %.1e param %.1e (%s)' % (n, rel(y1,y2), rel(gx1,gx2), worst[0], worst[1], rel(gx1b, gx1), sens[0], sens[1]))

print('---- inside layer2.2')
m = dict(blocks)['layer2.2']
X, G = rec['layer2.2']['x'], rec['layer2.2']['gy']
def inner(fwd):
    nl.SharableConv2d.forward = fwd
    m.zero_grad()
    xi = X.clone().requires_grad_(True)
    t = {}
    a1 = m.conv1(xi); t['conv1'] = a1
    b1 = m.relu(m.bn1(a1))
    a2 = m.conv2(b1); t['conv2'] = a2
    b2 = m.relu(m.bn2(a2))
    a3 = m.conv3(b2); t['conv3'] = a3
    pre = m.bn3(a3) + xi; t['pre'] = pre
    for v in t.values(): v.retain_grad()
    y = torch.relu(pre)
    y.backward(G)
    return {k: (v.detach().clone(), v.grad.clone()) for k, v in t.items()}, xi.grad.clone()
h, hgx = inner(hip_fwd)
r, rgx = inner(torch_fwd)
for k in ['pre', 'conv3', 'conv2', 'conv1']:
    print(k, 'value', rel(h[k][0], r[k][0]), 'grad', rel(h[k][1], r[k][1]))
pre_h, pre_r = h['pre'][0], r['pre'][0]
flip = (pre_h > 0) != (pre_r > 0)
print('relu mask flips', int(flip.sum()), 'of', flip.numel(), 'G at flips', G[flip].tolist()[:5], 'pre at flips', pre_h[flip].tolist()[:5], pre_r[flip].tolist()[:5])
print('G abs max', float(G.abs().max()), 'G abs mean', float(G.abs().mean()))
d = (h['conv3'][1] - r['conv3'][1]).abs()
print('conv3 grad diff: max at', np.unravel_index(int(d.argmax()), d.shape), float(d.max()), 'n>1e-6*max:', int((d > 1e-6 * float(r['conv3'][1].abs().max())).sum()))
d = (h['conv2'][1] - r['conv2'][1]).abs()
print('conv2-out grad diff: n bad', int((d > 1e-5 * float(r['conv2'][1].abs().max())).sum()), 'of', d.numel())
