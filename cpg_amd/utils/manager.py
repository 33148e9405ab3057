"""Manager: the per-minibatch train / validate loops of the reference's utils/manager.py:16-152,
driving the HIP-backed SparsePruner and masked layers.

Op order per step is the reference's (utils/manager.py:50-75): zero_grad -> forward -> accuracy ->
loss -> backward -> do_weight_decay_and_make_grads_zero -> optimizers.step -> (prune mode)
gradually_prune -> statistics.  `validate` calls apply_mask() FIRST and leaves the weights mutated,
as the reference does (utils/manager.py:105).

What is not reproduced is the reference's per-step host stall: loss / accuracy accumulate on the
device and the progress line (tqdm postfix with the mask statistics) is refreshed on a wall-clock
interval instead of forcing `.item()` + 15 reductions + `.cpu()` every step; the statistics themselves
ARE evaluated every step / validation batch (one cached histogram pass).  Returned values are identical.  Checkpoint save / load (utils/manager.py:198-320) delegate to
utils/checkpoint.py and keep the reference's file format; LFW evaluation (:156-195) needs real face pairs and
sklearn and is out of scope.
"""
import logging
import time

import numpy as np
import torch
import torch.nn as nn

from . import Metric, classification_accuracy
from . import checkpoint as ckpt
from .prune import SparsePruner

try:                                    # progress bar is cosmetic; tqdm is present in the image
    from tqdm import tqdm
except ImportError:                     # pragma: no cover
    tqdm = None


class _NullBar(object):
    def __init__(self, *a, **k):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def set_postfix(self, *a, **k):
        pass

    def update(self, n=1):
        pass


class Manager(object):
    """Handles training and pruning (utils/manager.py:16-37)."""

    def __init__(self, args, model, shared_layer_info, masks, train_loader, val_loader, begin_prune_step, end_prune_step):
        self.args = args
        self.model = model
        self.shared_layer_info = shared_layer_info
        root = model.module if hasattr(model, 'module') else model
        self.inference_dataset_idx = root.datasets.index(args.dataset) + 1
        self.pruner = SparsePruner(self.model, masks, self.args, begin_prune_step, end_prune_step,
                                   self.inference_dataset_idx)
        if hasattr(model, 'set_gradient_filter'):
            # data parallel: exchange only the gradient slots that survive the routing below (it runs right after the sync)
            model.set_gradient_filter(self.pruner)
        self.train_loader = train_loader
        self.val_loader = val_loader
        if getattr(args, 'freeze_gc', False):
            # OPT-IN (args.freeze_gc; no counterpart in the reference): the model, its masks and the pruner exist now -- one full collection,
            # then gc.freeze(): removes the 80 ms generation-2 collector pauses from the step loop (cpg_amd.utils.settle_host_gc: process-wide)
            from . import settle_host_gc
            settle_host_gc()
        self.progress = bool(getattr(args, 'progress', True)) and tqdm is not None
        self.last_stats = {}
        self.postfix_interval = float(getattr(args, 'postfix_interval', 0.5))
        if args.dataset == 'face_verification':
            from ..models.spherenet import AngleLoss
            self.criterion = AngleLoss()
        elif args.dataset == 'emotion':
            counts = torch.from_numpy(np.array([74874, 134415, 25459, 14090, 6378, 3803, 24882]).astype(np.float32))
            weights = (torch.sum(counts) - counts) / counts
            self.criterion = nn.CrossEntropyLoss(weight=weights.cuda() if getattr(args, 'cuda', True) else weights)
        else:
            self.criterion = nn.CrossEntropyLoss()

    def _bar(self, total, desc):
        return tqdm(total=total, desc=desc, ascii=True) if self.progress else _NullBar()

    def _to_device(self, data, target):
        if getattr(self.args, 'cuda', True):
            data, target = data.cuda(non_blocking=True), target.cuda(non_blocking=True)
        return data, target

    def train(self, optimizers, epoch_idx, curr_lrs, curr_prune_step):
        """One epoch of the hot loop (utils/manager.py:39-100). Returns (avg_train_acc, curr_prune_step)."""
        self.model.train()
        train_loss = Metric('train_loss')
        train_accuracy = Metric('train_accuracy')
        last_post = 0.0
        nbatches = len(self.train_loader)
        with self._bar(nbatches, 'Train Ep. #{}: '.format(epoch_idx + 1)) as t:
            for batch_idx, (data, target) in enumerate(self.train_loader):
                data, target = self._to_device(data, target)
                optimizers.zero_grad()
                output = self.model(data)
                num = data.size(0)
                if self.args.dataset != 'face_verification':
                    train_accuracy.update(classification_accuracy(output, target), num)
                loss = self.criterion(output, target)
                train_loss.update(loss, num)
                loss.backward()
                # Set fixed param grads to 0 (after the data-parallel all-reduce hooks have filled .grad)
                if hasattr(self.model, 'finish_gradient_sync'):
                    self.model.finish_gradient_sync()
                self.pruner.do_weight_decay_and_make_grads_zero()
                optimizers.step()
                if self.args.mode == 'prune':
                    self.pruner.gradually_prune(curr_prune_step)
                    curr_prune_step += 1
                # the mask statistic of the progress line is computed EVERY step, as the reference does
                # (utils/manager.py:77-88); it is one histogram pass that is cached until a mask mutates
                self.last_stats = {'sparsity': self.pruner.calculate_sparsity()}
                now = time.time()
                if self.progress and (now - last_post >= self.postfix_interval or batch_idx + 1 == nbatches):
                    last_post = now
                    t.set_postfix({'loss': train_loss.avg.item(),
                                   'accuracy': '{:.2f}'.format(100. * train_accuracy.avg.item()),
                                   'lr': curr_lrs[0],
                                   'sparsity': self.last_stats['sparsity'],
                                   'network_width_mpl': self.args.network_width_multiplier})
                t.update(1)
        if getattr(self.model, '_world', 1) > 1:        # data parallel: metrics of the GLOBAL batches, the same on every rank
            pg = getattr(self.model, 'process_group', None)
            train_loss.all_reduce_(pg)
            if self.args.dataset != 'face_verification':
                train_accuracy.all_reduce_(pg)
        summary = {'loss': '{:.3f}'.format(train_loss.avg.item()),
                   'accuracy': '{:.2f}'.format(100. * train_accuracy.avg.item()),
                   'lr': curr_lrs[0],
                   'sparsity': '{:.3f}'.format(self.pruner.calculate_sparsity()),
                   'network_width_mpl': self.args.network_width_multiplier}
        if getattr(self.args, 'log_path', None):
            logging.info(('In train()-> Train Ep. #{} '.format(epoch_idx + 1)
                          + ', '.join(['{}: {}'.format(k, v) for k, v in summary.items()])))
        return train_accuracy.avg.item(), curr_prune_step

    def validate(self, epoch_idx, biases=None):
        """Evaluation (utils/manager.py:103-152): apply_mask() first, then an eval-mode forward pass."""
        self.pruner.apply_mask()
        if hasattr(self.model, 'sync_buffers'):
            self.model.sync_buffers()            # data parallel: evaluate with rank 0's BN running statistics
        self.model.eval()
        val_loss = Metric('val_loss')
        val_accuracy = Metric('val_accuracy')
        idx = self.inference_dataset_idx
        last_post = 0.0
        nbatches = len(self.val_loader)
        with self._bar(nbatches, 'Val Ep. #{}: '.format(epoch_idx + 1)) as t:
            with torch.no_grad():
                for bi, (data, target) in enumerate(self.val_loader):
                    data, target = self._to_device(data, target)
                    output = self.model(data)
                    num = data.size(0)
                    val_loss.update(self.criterion(output, target), num)
                    val_accuracy.update(classification_accuracy(output, target), num)
                    # statistics per validation batch, as the reference (utils/manager.py:126-136); cached histogram
                    stats = {'sparsity': self.pruner.calculate_sparsity(),
                             'task{} ratio'.format(idx): self.pruner.calculate_curr_task_ratio(),
                             'zero ratio': self.pruner.calculate_zero_ratio()}
                    if idx != 1:
                        stats['shared_ratio'] = self.pruner.calculate_shared_part_ratio()
                    self.last_stats = stats
                    now = time.time()
                    if self.progress and (now - last_post >= self.postfix_interval or bi + 1 == nbatches):
                        last_post = now
                        post = {'loss': val_loss.avg.item(),
                                'accuracy': '{:.2f}'.format(100. * val_accuracy.avg.item())}
                        post.update(stats)
                        post['mpl'] = self.args.network_width_multiplier
                        t.set_postfix(post)
                    t.update(1)
        if getattr(self.model, '_world', 1) > 1:        # data parallel: every rank returns the same accuracy (sharded or replicated loader)
            pg = getattr(self.model, 'process_group', None)
            val_loss.all_reduce_(pg)
            val_accuracy.all_reduce_(pg)
        summary = {'loss': '{:.3f}'.format(val_loss.avg.item()),
                   'accuracy': '{:.2f}'.format(100. * val_accuracy.avg.item()),
                   'sparsity': '{:.3f}'.format(self.pruner.calculate_sparsity()),
                   'task{} ratio'.format(idx): '{:.3f}'.format(self.pruner.calculate_curr_task_ratio()),
                   'zero ratio': '{:.3f}'.format(self.pruner.calculate_zero_ratio()),
                   'mpl': self.args.network_width_multiplier}
        if idx != 1:
            summary['shared_ratio'] = '{:.3f}'.format(self.pruner.calculate_shared_part_ratio())
        if getattr(self.args, 'log_path', None):
            logging.info(('In validate()-> Val Ep. #{} '.format(epoch_idx + 1)
                          + ', '.join(['{}: {}'.format(k, v) for k, v in summary.items()])))
        return val_accuracy.avg.item()

    def eval_embeddings(self, epoch_idx=0):
        """The device half of the reference's evalLFW (utils/manager.py:156-175), which is what CPG_face_main.py runs instead of
        `validate` for the `face_verification` task (:337-341,:370-373,:417): apply_mask() first, eval mode, then
        `forward_to_embeddings` of every validation batch.  Returns the list of embedding tensors (on the device); scoring them as
        LFW pairs (utils/metrics.py: 10-fold ROC on the host with sklearn) needs the real pairs and is out of scope.  A loader
        that yields (a, p, label) pairs, as the reference's LFWDataset does, gives a list of (emb_a, emb_p, label)."""
        self.pruner.apply_mask()
        if hasattr(self.model, 'sync_buffers'):
            self.model.sync_buffers()
        self.model.eval()
        root = self.model.module if hasattr(self.model, 'module') else self.model
        out = []
        with torch.no_grad():
            for batch in self.val_loader:
                if len(batch) == 3:
                    a, p, label = batch
                    a, p = self._to_device(a, p)
                    out.append((root.forward_to_embeddings(a), root.forward_to_embeddings(p), label))
                else:
                    data, _ = self._to_device(*batch)
                    out.append(root.forward_to_embeddings(data))
                self.last_stats = {'sparsity': self.pruner.calculate_sparsity(),
                                   'task{} ratio'.format(self.inference_dataset_idx): self.pruner.calculate_curr_task_ratio()}
        return out

    # ------------------------------------------------------------------ checkpoints (utils/manager.py:198-320)
    def _path(self, folder, epoch):
        return self.args.checkpoint_format.format(save_folder=folder, epoch=epoch)

    def save_checkpoint(self, optimizers, epoch_idx, save_folder):
        """Same dict layout as the reference; optimizer state is not saved (momentum restarts each phase)."""
        ckpt.save_checkpoint(self.model, self.pruner.masks, self.shared_layer_info, self.args.dataset,
                             self._path(save_folder, epoch_idx + 1))

    def load_checkpoint(self, optimizers, resume_from_epoch, save_folder):
        if resume_from_epoch > 0:
            state = torch.load(self._path(save_folder, resume_from_epoch), map_location='cpu', weights_only=False)
            ckpt.load_state(self.model, state['model_state_dict'], for_evaluate=False)

    def load_checkpoint_only_for_evaluate(self, resume_from_epoch, save_folder):
        if resume_from_epoch > 0:
            state = torch.load(self._path(save_folder, resume_from_epoch), map_location='cpu', weights_only=False)
            ckpt.load_state(self.model, state['model_state_dict'], for_evaluate=True)
            ckpt.attach_task_layers(self.model, self.shared_layer_info, self.args.dataset)
