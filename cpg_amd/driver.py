"""Single-process CPG task loop.

The reference runs one `python CPG_*_main.py` per phase, orchestrated by bash through exit codes and checkpoint
files (experiment1/CPG_cifar100_scratch_mul_1.5.sh).  This module keeps the model, owner masks,
`shared_layer_info` and optimizers resident and runs the same phases back to back:

    for each task:  finetune  ->  gradual prune sweep (0 -> 0.1 -> ... )  ->  pick the sparsest ratio that holds the
                    accuracy goal  ->  (task >= 2) piggymask retrain

The set-up steps mirror CPG_cifar100_main_normal.py: model + head (:184-197), owner-mask allocation (:201-207),
piggymask creation for task >= 2 (:251-270), prune window (:308-309), SGD-nesterov for weights + Adam for piggymasks
(:320-346), LR schedule (:431-444).  What is policy rather than hot path is kept deliberately small: "grow the
network" on a missed accuracy goal is reported to the caller (`TaskResult.needs_growth`, the reference's exit code 2)
instead of being re-launched here, and data loading is whatever iterable of (images, labels) the caller provides.
"""
import copy
import types

import torch
import torch.nn as nn
from torch.nn.parameter import Parameter

from . import dist as cdist
from . import models
from .models import layers as nl
from .utils import Optimizers
from .utils.manager import Manager

VGG16_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']


def default_args(**over):
    """The flags the hot path reads (utils/prune.py, utils/manager.py), with experiment1's values."""
    a = dict(mode='finetune', dataset='task1', finetune_again=False, target_sparsity=0.1, initial_sparsity=0.0,
             pruning_frequency=10, pruning_interval=4, weight_decay=4e-5, network_width_multiplier=1.0, cuda=True,
             log_path=None, progress=False, lr=1e-2, lr_mask=5e-4, checkpoint_format='{save_folder}/checkpoint-{epoch}.pth.tar')
    a.update(over)
    return types.SimpleNamespace(**a)


def masked_layers(model):
    return [(n, m) for n, m in model.named_modules() if isinstance(m, (nl.SharableConv2d, nl.SharableLinear))]


class TaskResult(object):
    def __init__(self):
        self.finetune_acc = None
        self.ratio_to_acc = {}          # the reference's record.txt (pruning ratio -> validation accuracy)
        self.chosen_ratio = 0.0
        self.needs_growth = False       # reference exit code 2
        self.no_free_capacity = False   # reference exit code 5
        self.steps = 0


class CPGSession(object):
    """Holds everything the reference passes between processes through checkpoint files."""

    def __init__(self, arch='custom_vgg_cifar100', width=1.0, device='cuda', cfg=VGG16_CFG, data_parallel=True,
                 fused_optimizers=True):
        self.arch, self.width, self.device = arch, width, torch.device(device)
        self.fused_optimizers = fused_optimizers      # MaskedSGD / MaskedAdam: gradient routing fused into the optimizer passes
        kw = dict(dataset_history=[], dataset2num_classes={}, network_width_multiplier=width, shared_layer_info={})
        build = getattr(models, arch)
        self.net = build(cfg, **kw) if 'vgg' in arch else build(**kw)
        self.shared_layer_info = {}
        self.masks = {}
        self.model = None
        self.data_parallel = data_parallel

    # -- per-task set-up (CPG_cifar100_main_normal.py:196-290) -------------------------------------------------
    def start_task(self, dataset, num_classes):
        self.net.add_dataset(dataset, num_classes)
        self.net.set_dataset(dataset)
        if self.model is None:
            self.net.to(self.device)
            self.model = cdist.DataParallel(self.net) if self.data_parallel else self.net
        else:
            self.net.classifiers.to(self.device)
        if not self.masks:
            for name, module in masked_layers(self.model):
                self.masks[name] = torch.zeros(module.weight.shape, dtype=torch.uint8, device=self.device)
        task_id = self.net.datasets.index(dataset) + 1
        if dataset not in self.shared_layer_info:
            self.shared_layer_info[dataset] = {k: {} for k in ('bias', 'bn_layer_running_mean', 'bn_layer_running_var',
                                                               'bn_layer_weight', 'bn_layer_bias', 'piggymask')}
            if task_id > 1:
                self._fresh_piggymasks()
        self.shared_layer_info[dataset]['network_width_multiplier'] = self.width
        if hasattr(self.model, 'refresh_hooks'):
            self.model.refresh_hooks()
        return task_id

    def _fresh_piggymasks(self):
        """Real-valued picker over the older tasks' weights, initialised at 0.01 (:263-270, :272-279)."""
        root = self.net
        prefix = 'module.' if hasattr(self.model, 'module') else ''
        for name, module in root.named_modules():
            if isinstance(module, (nl.SharableConv2d, nl.SharableLinear)):
                pm = torch.full_like(self.masks[prefix + name], 0.01, dtype=torch.float32)
                module.piggymask = Parameter(pm)

    def make_optimizers(self, args, pruner=None):
        """Head of the current task + every non-piggymask parameter -> SGD(nesterov); piggymasks -> Adam (:320-346).
        With a pruner (and fused_optimizers) the masked weights / piggymasks take the fused routing + update passes."""
        idx = self.net.datasets.index(args.dataset)
        sgd_params, adam_params = [], []
        for name, p in self.model.named_parameters():
            if 'classifiers' in name:
                if '.{}.'.format(idx) in name:
                    sgd_params.append(p)
            elif 'piggymask' in name:
                adam_params.append(p)
            else:
                sgd_params.append(p)
        opts = Optimizers()
        if pruner is not None and self.fused_optimizers:
            from .utils.fused_sgd import MaskedAdam, MaskedSGD
            opts.add(MaskedSGD(sgd_params, pruner=pruner, lr=args.lr, momentum=0.9, nesterov=True), args.lr)
            if adam_params:
                opts.add(MaskedAdam(adam_params, pruner=pruner, lr=args.lr_mask), args.lr_mask)
            return opts
        opts.add(torch.optim.SGD(sgd_params, lr=args.lr, weight_decay=0.0, momentum=0.9, nesterov=True), args.lr)
        if adam_params:
            opts.add(torch.optim.Adam(adam_params, lr=args.lr_mask), args.lr_mask)
        return opts

    # -- phases ------------------------------------------------------------------------------------------------
    def _manager(self, args, train_loader, val_loader, begin, end):
        return Manager(args, self.model, self.shared_layer_info, self.masks, train_loader, val_loader, begin, end)

    def finetune(self, args, train_loader, val_loader, epochs, lr_drops=(50, 80)):
        """`--mode finetune` (:386-388, :401-444).  Returns (manager, last train acc, last val acc)."""
        args = copy.copy(args)
        args.mode = 'finetune'
        mgr = self._manager(args, train_loader, val_loader, 0, 0)
        if not args.finetune_again:
            mgr.pruner.make_finetuning_mask()
        opts = self.make_optimizers(args, mgr.pruner)
        lrs = list(opts.lrs)
        stop_lr_mask = mgr.pruner.calculate_curr_task_ratio() != 0.0
        tr = va = 0.0
        step = 0
        for epoch in range(epochs):
            tr, step = mgr.train(opts, epoch, lrs, step)
            va = mgr.validate(epoch)
            if epoch + 1 in lr_drops:
                for g in opts[0].param_groups:
                    g['lr'] *= 0.1
                lrs[0] = opts[0].param_groups[0]['lr']
            if len(opts.lrs) == 2:
                if epoch + 1 == 50:
                    for g in opts[1].param_groups:
                        g['lr'] *= 0.2
                if stop_lr_mask and epoch + 1 == 70:
                    for g in opts[1].param_groups:
                        g['lr'] *= 0.0
        return mgr, tr, va

    def prune(self, args, train_loader, val_loader, initial, target, epochs):
        """One `--mode prune` run initial -> target: the first `pruning_interval` epochs release weights every
        `pruning_frequency` steps, the remaining epochs retrain at the fixed mask (:308-309, :384, :401-404)."""
        args = copy.copy(args)
        args.mode, args.initial_sparsity, args.target_sparsity = 'prune', initial, target
        args.lr, args.lr_mask = 1e-3, 0.0
        steps_per_epoch = len(train_loader)
        mgr = self._manager(args, train_loader, val_loader, 0, args.pruning_interval * steps_per_epoch)
        mgr.validate(-1)
        opts = self.make_optimizers(args, mgr.pruner)
        lrs = list(opts.lrs)
        tr = va = 0.0
        step = 0
        for epoch in range(epochs):
            tr, step = mgr.train(opts, epoch, lrs, step)
            va = mgr.validate(epoch)
        return mgr, tr, va

    def run_task(self, dataset, num_classes, train_loader, val_loader, accuracy_goal=0.0, finetune_epochs=1,
                 prune_epochs=1, sparsities=(0.1, 0.2, 0.3), args=None, min_train_acc=0.0, allow_acc_loss=0.0):
        """finetune -> prune sweep -> choose ratio (tools/choose_appropriate_pruning_ratio_for_next_task.py) ->
        piggymask retrain for task >= 2.  `accuracy_goal` plays baseline_cifar100_acc.txt's role."""
        res = TaskResult()
        args = args or default_args()
        args = copy.copy(args)
        args.dataset, args.network_width_multiplier = dataset, self.width
        task_id = self.start_task(dataset, num_classes)
        mgr, tr, va = self.finetune(args, train_loader, val_loader, finetune_epochs)
        res.finetune_acc = va
        res.ratio_to_acc[0.0] = round(va, 4)
        if va < accuracy_goal:
            res.needs_growth = True                    # reference: sys.exit(2) -> bash widens the network
            return res
        if mgr.pruner.calculate_curr_task_ratio() == 0.0:
            res.no_free_capacity = True                # reference: sys.exit(5)
            return res
        prev = 0.0
        for s in sparsities:
            snapshot = (copy.deepcopy(self.net.state_dict()), {k: v.clone() for k, v in self.masks.items()})
            mgr, tr, va = self.prune(args, train_loader, val_loader, prev, s, prune_epochs)
            if tr < min_train_acc:                     # reference: sys.exit(6), keep the previous sparsity level
                self.net.load_state_dict(snapshot[0])
                for k, v in snapshot[1].items():
                    self.masks[k].copy_(v)
                break
            res.ratio_to_acc[s] = round(va, 4)
            prev = s
        # sparsest ratio whose accuracy holds the goal (the reference walks the record from the sparsest down)
        res.chosen_ratio = 0.0
        for s in sorted((k for k in res.ratio_to_acc if k > 0.0), reverse=True):
            if res.ratio_to_acc[s] + allow_acc_loss >= accuracy_goal:
                res.chosen_ratio = s
                break
        if task_id > 1:
            again = copy.copy(args)
            again.finetune_again, again.lr_mask = True, 1e-4
            self._fresh_piggymasks()
            if hasattr(self.model, 'refresh_hooks'):
                self.model.refresh_hooks()
            self.finetune(again, train_loader, val_loader, 1)
        return res
