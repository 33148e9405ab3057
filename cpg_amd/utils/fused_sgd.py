"""Fused masked SGD.

`MaskedSGD` is a torch.optim.SGD whose step() handles every masked weight (SharableConv2d / SharableLinear
`.weight`) with ONE libcpg_hip.so pass that also performs the reference's gradient routing
(`SparsePruner.do_weight_decay_and_make_grads_zero`, utils/prune.py:203-205); all other parameters (BN affine,
biases, the task head) take torch's own SGD path.  Results equal routing-then-torch-SGD to 1 ulp (the fused
multiply-adds differ in rounding), including the momentum that keeps moving released / frozen weights for a few
steps after their gradient is zeroed (SURVEY 8a note).

Usage (drop-in for CPG_cifar100_main_normal.py:339-341 + utils/manager.py:67-70):

    optimizer_network = MaskedSGD(params_to_optimize_via_SGD, pruner=manager.pruner, lr=lr, momentum=0.9, nesterov=True)
    ...
    loss.backward(); manager.pruner.do_weight_decay_and_make_grads_zero(); optimizers.step()

While a MaskedSGD is attached, `do_weight_decay_and_make_grads_zero()` skips the weight part (piggymask
gradients are still routed there) and step() does it fused; `.grad` ends up routed exactly as before.
"""
import os

import torch

from .. import _lib

# CPG_MULTI_TENSOR=0: one cpg_sgd_route_step / cpg_adam_route_step launch per layer (the behaviour up to round 5; A/B switch)
MULTI_TENSOR = os.environ.get('CPG_MULTI_TENSOR', '1') not in ('0', '')
from ..models import layers as nl


class MaskedSGD(torch.optim.SGD):
    def __init__(self, params, pruner, lr, momentum=0.9, nesterov=True, **kw):
        if kw.get('weight_decay', 0.0) != 0.0 or kw.get('dampening', 0.0) != 0.0:
            raise ValueError('MaskedSGD mirrors the reference optimizer: weight_decay = dampening = 0 '
                             '(the decay is applied by the gradient routing)')
        super().__init__(params, lr=lr, momentum=momentum, nesterov=nesterov, weight_decay=0.0, dampening=0.0)
        self.pruner = pruner
        pruner.fused_weight_step = True          # tells the pruner to leave masked-weight grads to us
        self._masked = {}                        # id(param) -> mask name

    def _refresh(self):
        self._masked = {}
        for name, module in self.pruner.model.named_modules():
            if isinstance(module, (nl.SharableConv2d, nl.SharableLinear)):
                self._masked[id(module.weight)] = name

    @torch.no_grad()
    def step(self, closure=None):
        if not self._masked:
            self._refresh()
        L = _lib.lib()
        s = _lib.stream_ptr()
        pr = self.pruner
        held = []
        for group in self.param_groups:
            # every masked weight of the group that shares `first` (no momentum buffer yet / has one) goes into ONE multi-tensor launch
            # (cpg_sgd_route_step_multi: 53 layers of ResNet-50 = 2 launches instead of 53)
            batches = {True: [], False: []}
            for p in group['params']:
                name = self._masked.get(id(p))
                if name is None or p.grad is None:
                    continue
                state = self.state[p]
                first = 'momentum_buffer' not in state or state['momentum_buffer'] is None
                if first:
                    state['momentum_buffer'] = torch.empty_like(p, memory_format=torch.contiguous_format)
                owner = pr._owner(name, p.data)
                batches[first].append((_lib.dptr(p.data, name='weight').value, _lib.dptr(p.grad, name='weight.grad').value,
                                       _lib.dptr(state['momentum_buffer'], name='momentum').value,
                                       _lib.dptr(owner, torch.uint8, 'mask').value, p.numel()))
                held.append((p, p.grad))
                p.grad = None                    # hide from torch's SGD for the rest of this step
            for first, rows in batches.items():
                if not rows:
                    continue
                if not MULTI_TENSOR:
                    for w_, g_, b_, o_, n_ in rows:
                        rc = L.cpg_sgd_route_step(w_, g_, b_, o_, int(pr.current_dataset_idx), float(pr.args.weight_decay), float(group['lr']),
                                                  float(group['momentum']), int(bool(group['nesterov'])), int(first), n_, s)
                        _lib.check('cpg_sgd_route_step', rc)
                    continue
                items = (_lib.SgdItem * len(rows))(*rows)
                rc = L.cpg_sgd_route_step_multi(items, len(rows), int(pr.current_dataset_idx), float(pr.args.weight_decay), float(group['lr']),
                                                float(group['momentum']), int(bool(group['nesterov'])), int(first), s)
                _lib.check('cpg_sgd_route_step_multi', rc)
        loss = super().step(closure)
        for p, g in held:
            p.grad = g
        return loss


class MaskedAdam(torch.optim.Adam):
    """torch.optim.Adam for the piggymasks (CPG_cifar100_main_normal.py:342-346) whose step() handles every
    `module.piggymask` with ONE pass that also performs the piggymask part of the gradient routing
    (utils/prune.py:206-210) -- cpg_adam_route_step.  Other parameters in its groups take torch's own path.  While a
    MaskedAdam is attached, `do_weight_decay_and_make_grads_zero()` leaves piggymask gradients alone; `.grad` ends up
    routed exactly as before, and the optimizer state keeps torch's keys (step, exp_avg, exp_avg_sq)."""

    def __init__(self, params, pruner, lr, betas=(0.9, 0.999), eps=1e-8, **kw):
        if kw.get('weight_decay', 0.0) != 0.0 or kw.get('amsgrad', False) or kw.get('maximize', False):
            raise ValueError('MaskedAdam mirrors the reference optimizer: weight_decay = 0, amsgrad = maximize = False')
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=0.0)
        self.pruner = pruner
        pruner.fused_piggymask_step = True
        self._masked = {}

    def _refresh(self):
        self._masked = {}
        for name, module in self.pruner.model.named_modules():
            if isinstance(module, (nl.SharableConv2d, nl.SharableLinear)) and module.piggymask is not None:
                self._masked[id(module.piggymask)] = name

    @torch.no_grad()
    def step(self, closure=None):
        self._refresh()                          # piggymasks are re-created between phases (driver._fresh_piggymasks)
        L = _lib.lib()
        s = _lib.stream_ptr()
        pr = self.pruner
        mode = {'finetune': _lib.MODE_FINETUNE, 'prune': _lib.MODE_PRUNE}.get(pr.args.mode)
        held, idle = [], []
        for group in self.param_groups:
            beta1, beta2 = group['betas']
            batches = {}                         # Adam step count -> rows of one multi-tensor launch
            for p in group['params']:
                name = self._masked.get(id(p))
                if name is None or p.grad is None or mode is None:
                    continue
                state = self.state[p]
                if len(state) == 0:
                    state['step'] = torch.tensor(0.0, dtype=torch.float32)
                    state['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    state['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    # the moments are exactly zero until the kernel below has run once.  The flag lives IN the state (not in a set of
                    # id(p) beside it), so that load_state_dict() / any replacement of the state drops it with the state it described
                    state['_pristine'] = True
                state['step'] += 1
                held.append((p, p.grad))
                if mode == _lib.MODE_PRUNE and state.get('_pristine', False):
                    # Prune mode routes EVERY piggymask gradient to zero (utils/prune.py:209-210), and this parameter's moments are
                    # still exactly zero: Adam's update is then m = v = 0, pm -= step_size * 0 / eps -- nothing changes, bit for bit.
                    # The whole prune phase of the reference (lr_mask 0, a fresh optimizer per phase) is this case: only the routed
                    # gradient (all zeros) has to be left in .grad -- 4 B per element instead of the 37 B of the full pass.
                    idle.append(p.grad)
                    p.grad = None
                    continue
                state.pop('_pristine', None)
                owner = pr._owner(name, p.data)
                batches.setdefault(int(state['step']), []).append(
                    (_lib.dptr(p.data, name='piggymask').value, _lib.dptr(p.grad, name='piggymask.grad').value,
                     _lib.dptr(state['exp_avg']).value, _lib.dptr(state['exp_avg_sq']).value,
                     _lib.dptr(owner, torch.uint8, 'mask').value, p.numel()))
                p.grad = None
            for step, rows in batches.items():   # (cpg_adam_route_step_multi: every piggymask of the group in one launch)
                if not MULTI_TENSOR:
                    for p_, g_, a_, b_, o_, n_ in rows:
                        rc = L.cpg_adam_route_step(p_, g_, a_, b_, o_, int(pr.current_dataset_idx), mode, float(group['lr']), float(beta1),
                                                   float(beta2), float(group['eps']), step, n_, s)
                        _lib.check('cpg_adam_route_step', rc)
                    continue
                items = (_lib.AdamItem * len(rows))(*rows)
                rc = L.cpg_adam_route_step_multi(items, len(rows), int(pr.current_dataset_idx), mode, float(group['lr']), float(beta1),
                                                 float(beta2), float(group['eps']), step, s)
                _lib.check('cpg_adam_route_step_multi', rc)
        if idle:
            torch._foreach_zero_(idle)           # (one multi-tensor kernel)
        if len(held) > len(idle):
            pr._pm_mutations += 1                # the kernel wrote through data_ptr(): tensor._version did not move
        loss = super().step(closure)
        for p, g in held:
            p.grad = g
        return loss
