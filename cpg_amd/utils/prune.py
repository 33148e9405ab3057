"""SparsePruner on MI355X: the reference's utils/prune.py behind the same class API.

Every method keeps the reference's name, arguments and observable effect; the per-layer tensor
work is one libcpg_hip.so call each (include/cpg_hip.h):

    _pruning_mask / gradually_prune / one_shot_prune   -> cpg_rank_prune   (utils/prune.py:30-109)
    calculate_* (4 statistics)                         -> cpg_mask_hist    (utils/prune.py:111-193)
    do_weight_decay_and_make_grads_zero                -> cpg_route_grads  (utils/prune.py:195-211)
    make_pruned_zero / apply_mask                      -> cpg_zero_pruned / cpg_apply_mask (:213-231)
    make_finetuning_mask                               -> cpg_claim_free   (utils/prune.py:233-243)

Differences that are deliberate (values identical, stalls removed):
  * the k-th-value search runs on the device (the reference copies every layer to the host);
    the only host read-back is one 32-byte result record per layer after a prune event, needed to
    reproduce the reference's `sys.exit(2)` when a layer has too few candidates;
  * the four statistics share one histogram pass and are cached until a mask (or, for the shared
    ratio, a piggymask) is mutated -- detected through tensor._version plus an internal counter.
"""
import ctypes
import sys

import torch

from .. import _lib
from ..models import layers as nl


def _masked(module):
    return isinstance(module, (nl.SharableConv2d, nl.SharableLinear))


class SparsePruner(object):
    """Performs pruning on the given model (utils/prune.py:6-28)."""
    _epochs = 0

    def __init__(self, model, masks, args, begin_prune_step, end_prune_step, inference_dataset_idx):
        self.model = model
        self.args = args
        self.sparsity_func_exponent = 3
        self.begin_prune_step = begin_prune_step
        self.end_prune_step = end_prune_step
        self.last_prune_step = begin_prune_step
        self.masks = masks
        datasets = self._root().datasets
        again = bool(getattr(args, 'finetune_again', False))
        if args.mode in ('prune', 'inference') or (args.mode == 'finetune' and again):
            self.current_dataset_idx = datasets.index(args.dataset) + 1
        elif args.mode == 'finetune':
            self.current_dataset_idx = len(datasets) - 1
        else:
            print("We do not support '{}' mode".format(args.mode))
            sys.exit(-1)
        self.inference_dataset_idx = inference_dataset_idx
        self.fused_weight_step = False   # set by utils.fused_sgd.MaskedSGD: it routes masked-weight grads itself
        self.fused_piggymask_step = False   # set by utils.fused_sgd.MaskedAdam: it routes piggymask grads itself
        SparsePruner._epochs += 1
        self._epoch = SparsePruner._epochs      # distinguishes this pruner's mutation counter from an earlier pruner's (plan caches)
        self._mutations = 0          # bumped whenever a kernel of ours rewrites a mask in place
        self._pm_mutations = 0       # bumped when a kernel of ours rewrites a piggymask through its raw pointer (MaskedAdam)
        self._hist_key = None
        self._hist = None
        self.last_prune_records = []
        self.prune_events = 0        # rank-prune events executed (gradually_prune + one_shot_prune)

    # ------------------------------------------------------------------ helpers
    def _root(self):
        return self.model.module if hasattr(self.model, 'module') else self.model

    def _layers(self):
        for name, module in self.model.named_modules():
            if _masked(module):
                yield name, module

    def _owner(self, name, like):
        m = self.masks[name]
        if m.dtype != torch.uint8:
            raise TypeError('mask %s must be uint8 (torch.ByteTensor), got %s' % (name, m.dtype))
        if m.device != like.device:
            # the reference moves masks next to the weights lazily (utils/prune.py:228)
            m = m.to(like.device)
            self.masks[name] = m
        if m.shape != like.shape:
            raise RuntimeError('mask %s has shape %s, weight has %s' % (name, tuple(m.shape), tuple(like.shape)))
        if not m.is_contiguous():
            m = m.contiguous()
            self.masks[name] = m
        return m

    # ------------------------------------------------------------------ rank prune
    def _rank_prune_layers(self, pruning_ratio):
        """Launch cpg_rank_prune for every masked layer, then read the result records once."""
        L = _lib.lib()
        layers = list(self._layers())
        if not layers:
            return []
        dev = layers[0][1].weight.device
        res = torch.zeros(len(layers), _lib.PRUNE_RESULT_BYTES // 8, dtype=torch.int64, device=dev)
        ws, nbytes = _lib.workspace(L.cpg_rank_prune_workspace_bytes(), dev)
        s = _lib.stream_ptr()
        for i, (name, module) in enumerate(layers):
            w = module.weight.data
            owner = self._owner(name, w)
            rc = L.cpg_rank_prune(_lib.dptr(w, name='weight'), _lib.dptr(owner, torch.uint8, 'mask'),
                                  int(self.current_dataset_idx), float(pruning_ratio), w.numel(),
                                  ctypes.c_void_p(res[i].data_ptr()), _lib.dptr(ws), nbytes, s)
            _lib.check('cpg_rank_prune', rc)
        self._mutations += 1
        self.prune_events += 1
        raw = res.cpu().numpy().tobytes()
        recs = []
        for i, (name, _) in enumerate(layers):
            r = _lib.PruneResult.from_buffer_copy(raw[i * _lib.PRUNE_RESULT_BYTES:(i + 1) * _lib.PRUNE_RESULT_BYTES])
            recs.append({'layer': name, 'n_candidates': r.n_candidates, 'k': r.k, 'n_released': r.n_released,
                         'cutoff': r.cutoff, 'status': r.status})
        self.last_prune_records = recs
        if any(r['status'] == _lib.CPG_E_KRANGE for r in recs):
            # utils/prune.py:38-42
            print("Not enough weights for pruning, that is to say, too little space for new task, need expand the network.")
            sys.exit(2)
        return recs

    def _pruning_mask(self, weights, mask, layer_name, pruning_ratio):
        """Rank one layer by magnitude and release the smallest weights of the current task
        (utils/prune.py:30-53).  Mutates and returns `mask`."""
        L = _lib.lib()
        weights = weights.contiguous()
        res = torch.zeros(_lib.PRUNE_RESULT_BYTES // 8, dtype=torch.int64, device=weights.device)
        ws, nbytes = _lib.workspace(L.cpg_rank_prune_workspace_bytes(), weights.device)
        rc = L.cpg_rank_prune(_lib.dptr(weights, name='weights'), _lib.dptr(mask, torch.uint8, 'mask'),
                              int(self.current_dataset_idx), float(pruning_ratio), weights.numel(),
                              ctypes.c_void_p(res.data_ptr()), _lib.dptr(ws), nbytes, _lib.stream_ptr())
        _lib.check('cpg_rank_prune', rc)
        self._mutations += 1
        r = _lib.PruneResult.from_buffer_copy(res.cpu().numpy().tobytes())
        if r.status == _lib.CPG_E_KRANGE:
            print("Not enough weights for pruning, that is to say, too little space for new task, need expand the network.")
            sys.exit(2)
        return mask

    def _adjust_sparsity(self, curr_prune_step):
        """Cubic sparsity schedule (utils/prune.py:55-66); python floats, bit-identical."""
        p = min(1.0, max(0.0, ((curr_prune_step - self.begin_prune_step)
                               / (self.end_prune_step - self.begin_prune_step))))
        return self.args.target_sparsity + \
            (self.args.initial_sparsity - self.args.target_sparsity) * pow(1 - p, self.sparsity_func_exponent)

    def _time_to_update_masks(self, curr_prune_step):
        """Update gate (utils/prune.py:68-76)."""
        in_range = self.begin_prune_step <= curr_prune_step <= self.end_prune_step
        due = (self.last_prune_step + self.args.pruning_frequency) <= curr_prune_step
        return in_range and due

    def gradually_prune(self, curr_prune_step):
        """utils/prune.py:78-92.  Weights are NOT zeroed here (the reference leaves released weights
        stale until the next apply_mask)."""
        if self._time_to_update_masks(curr_prune_step):
            self.last_prune_step = curr_prune_step
            curr_pruning_ratio = self._adjust_sparsity(curr_prune_step)
            self._rank_prune_layers(curr_pruning_ratio)
        else:
            curr_pruning_ratio = self._adjust_sparsity(self.last_prune_step)
        return curr_pruning_ratio

    def one_shot_prune(self, one_shot_prune_perc):
        """utils/prune.py:94-109: fixed-ratio prune, then zero the released weights."""
        print('Pruning for dataset idx: %d' % (self.current_dataset_idx))
        print('Pruning each layer by removing %.2f%% of values' % (100 * one_shot_prune_perc))
        self._rank_prune_layers(one_shot_prune_perc)
        self.make_pruned_zero()

    # ------------------------------------------------------------------ statistics
    def _histogram(self, with_piggymask=False):
        """257-entry count vector over all masked layers: [#owner==id for id in 0..255] + [#picked]."""
        layers = list(self._layers())
        key = (self._mutations, self._pm_mutations if with_piggymask else 0, with_piggymask, self.inference_dataset_idx,
               tuple((id(self.masks[n]), self.masks[n]._version) for n, _ in layers),
               tuple((id(m.piggymask), m.piggymask._version) for _, m in layers if m.piggymask is not None)
               if with_piggymask else ())
        if key == self._hist_key:
            return self._hist
        if not layers:
            return [0] * 257
        L = _lib.lib()
        dev = layers[0][1].weight.device
        hist = torch.zeros(257, dtype=torch.int64, device=dev)
        s = _lib.stream_ptr()
        for name, module in layers:
            owner = self._owner(name, module.weight.data)
            pm = None
            if with_piggymask:
                pm = module.piggymask.data.contiguous()       # AttributeError on None, as in the reference
            rc = L.cpg_mask_hist(_lib.dptr(owner, torch.uint8, 'mask'), _lib.dptr(pm, name='piggymask'),
                                 int(self.inference_dataset_idx), owner.numel(), ctypes.c_void_p(hist.data_ptr()), s)
            _lib.check('cpg_mask_hist', rc)
        self._hist = hist.cpu().tolist()
        self._hist_key = key
        return self._hist

    def _numel(self):
        return sum(self.masks[n].numel() for n, _ in self._layers())

    def calculate_sparsity(self):
        """#free / #(free or owned by the inference task) (utils/prune.py:111-136)."""
        h = self._histogram()
        idx = self.inference_dataset_idx
        total = h[0] + (h[idx] if 0 < idx < 256 else 0)
        return float(h[0]) / float(total) if total != 0 else 0.0

    def calculate_curr_task_ratio(self):
        """utils/prune.py:138-157."""
        h = self._histogram()
        return float(h[self.inference_dataset_idx]) / self._numel() * (self.args.network_width_multiplier ** 2)

    def calculate_zero_ratio(self):
        """utils/prune.py:159-178."""
        h = self._histogram()
        return float(h[0]) / self._numel() * (self.args.network_width_multiplier ** 2)

    def calculate_shared_part_ratio(self):
        """Share of older tasks' weights picked by the piggymask (utils/prune.py:180-193)."""
        h = self._histogram(with_piggymask=True)
        total = sum(h[1:self.inference_dataset_idx])
        return float(h[256]) / float(total) if total != 0 else 0.0

    # ------------------------------------------------------------------ gradient routing
    def do_weight_decay_and_make_grads_zero(self):
        """Sets grads of fixed weights to 0 (utils/prune.py:195-211), one fused pass per layer."""
        assert self.masks
        L = _lib.lib()
        s = _lib.stream_ptr()
        mode = {'finetune': _lib.MODE_FINETUNE, 'prune': _lib.MODE_PRUNE}.get(self.args.mode)
        for name, module in self._layers():
            w = module.weight
            if self.fused_weight_step:
                # weight part deferred to MaskedSGD.step(); only the piggymask gradient is routed here
                pm = module.piggymask
                if pm is None or pm.grad is None or mode is None or self.fused_piggymask_step:
                    continue       # (piggymask gradients are routed inside MaskedAdam.step() when one is attached)
                owner = self._owner(name, w.data)
                scratch = torch.zeros_like(w.data)
                rc = L.cpg_route_grads(_lib.dptr(scratch), _lib.dptr(w.data.contiguous(), name='weight'),
                                       _lib.dptr(owner, torch.uint8, 'mask'), int(self.current_dataset_idx), 0.0,
                                       _lib.dptr(pm.grad.data, name='piggymask.grad'), mode, scratch.numel(), s)
                _lib.check('cpg_route_grads', rc)
                continue
            if w.grad is None:
                # the reference still routes a piggymask grad here; without a weight grad only that part applies
                gw = None
            else:
                gw = w.grad.data
            pm = module.piggymask
            gpm = pm.grad.data if (pm is not None and pm.grad is not None and mode is not None and not self.fused_piggymask_step) else None
            if gw is None and gpm is None:
                continue
            owner = self._owner(name, w.data)
            if gw is None:
                # rare: only the piggymask received a gradient; run the routing on a scratch weight grad
                gw = torch.zeros_like(w.data)
            if not gw.is_contiguous() or (gpm is not None and not gpm.is_contiguous()):
                raise RuntimeError('gradient of %s is not contiguous' % name)
            rc = L.cpg_route_grads(_lib.dptr(gw, name='weight.grad'), _lib.dptr(w.data.contiguous(), name='weight'),
                                   _lib.dptr(owner, torch.uint8, 'mask'), int(self.current_dataset_idx),
                                   float(self.args.weight_decay), _lib.dptr(gpm, name='piggymask.grad'),
                                   mode if mode is not None else _lib.MODE_FINETUNE, gw.numel(), s)
            _lib.check('cpg_route_grads', rc)

    # ------------------------------------------------------------------ mask application
    def make_pruned_zero(self):
        """Makes pruned weights 0 (utils/prune.py:213-221)."""
        assert self.masks
        L = _lib.lib()
        s = _lib.stream_ptr()
        for name, module in self._layers():
            w = module.weight.data
            rc = L.cpg_zero_pruned(_lib.dptr(w, name='weight'), _lib.dptr(self._owner(name, w), torch.uint8, 'mask'),
                                   w.numel(), s)
            _lib.check('cpg_zero_pruned', rc)

    def apply_mask(self):
        """Keep only the weights of tasks <= inference_dataset_idx, destructively (utils/prune.py:223-231)."""
        L = _lib.lib()
        s = _lib.stream_ptr()
        for name, module in self._layers():
            w = module.weight.data
            rc = L.cpg_apply_mask(_lib.dptr(w, name='weight'), _lib.dptr(self._owner(name, w), torch.uint8, 'mask'),
                                  int(self.inference_dataset_idx), w.numel(), s)
            _lib.check('cpg_apply_mask', rc)

    def make_finetuning_mask(self):
        """Hand every free slot to the new task (utils/prune.py:233-243)."""
        assert self.masks
        self.current_dataset_idx += 1
        L = _lib.lib()
        s = _lib.stream_ptr()
        for name, module in self._layers():
            owner = self._owner(name, module.weight.data)
            rc = L.cpg_claim_free(_lib.dptr(owner, torch.uint8, 'mask'), int(self.current_dataset_idx), owner.numel(), s)
            _lib.check('cpg_claim_free', rc)
        self._mutations += 1
