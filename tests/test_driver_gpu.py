"""cpg_amd.driver.CPGSession on the GPU: the single-process replacement of the reference's bash state machine
(experiment1/CPG_cifar100_scratch_mul_1.5.sh + tools/*.py).

What is asserted are CPG's own invariants, which need no trained accuracy:
  * the sweep's chosen stage is what the model is left in (tools/choose_appropriate_pruning_ratio_for_next_task.py copies
    that checkpoint; "no stage holds the goal" goes back to the pre-prune checkpoint);
  * NO FORGETTING: after task 2 (piggymasks, Adam, retrain) and after a network-growth step, evaluating task 1 -- at its
    own width, with its own BatchNorm snapshot -- gives bit-identical logits;
  * growth pads the owner masks with free slots, keeps the old weights in the top-left corner, and the new task trains in
    the widened net; checkpoints round-trip through the reference's file format on the GPU.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _data(seed, n=16, task=0, ncls=5):
    g = torch.Generator().manual_seed(seed)
    t = torch.randint(0, ncls, (n,), generator=g)
    x = 0.5 * torch.randn(n, 3, 32, 32, generator=g)
    for i in range(n):                                   # a class- and task-dependent bright block: learnable, not needed to be learnt
        c = int(t[i])
        r, col, ch = (c * 5 + task * 3) % 16, (c * 6) % 20, (c + task) % 3
        x[i, ch, r:r + 12, col:col + 12] += 2.0
    return x.to(DEV), t.to(DEV)


def _loaders(task):
    return [_data(10 + 50 * task + i, task=task) for i in range(4)], [_data(500 + 50 * task + i, task=task) for i in range(2)]


def _session(width=0.125):
    from cpg_amd.driver import CPGSession, default_args
    sess = CPGSession('custom_vgg_cifar100', width, device=DEV, seed=1)
    args = default_args(lr=5e-2, lr_mask=5e-4, pruning_frequency=1, pruning_interval=1, prune_lr=1e-2)
    return sess, args


def _zero_fraction(sess, owner=None):
    tot = sum(v.numel() for v in sess.masks.values())
    z = sum(int((v == 0).sum()) for v in sess.masks.values())
    return z / tot


def test_run_task_sweep_restores_chosen_stage_and_task2_does_not_forget():
    sess, args = _session()
    tr1, va1 = _loaders(0)
    res = sess.run_task('t1', 5, tr1, va1, accuracy_goal=0.0, finetune_epochs=2, prune_epochs=1, sparsities=(0.2, 0.4),
                        args=args, min_train_acc=-1.0)
    assert set(res.ratio_to_acc) == {0.0, 0.2, 0.4} and res.chosen_ratio == 0.4 and not res.grown_to and not res.needs_growth
    # the model is left in the 0.4 stage: its last rank-prune event ran at step 3 of a 4-step window, where the cubic
    # schedule (utils/prune.py:55-66) stands at 0.4 - 0.2 / 64
    assert abs(_zero_fraction(sess) - (0.4 - 0.2 / 64)) < 1e-3
    acc1, logits1 = sess.evaluate('t1', va1)
    assert abs(acc1 - res.ratio_to_acc[0.4]) < 1e-4                                 # ... whose recorded accuracy it reproduces
    w1 = {n: m.weight.detach().clone() for n, m in sess.net.named_modules() if hasattr(m, 'piggymask')}
    own1 = {k: (v == 1) for k, v in sess.masks.items()}

    # a sweep in which NO stage holds the goal goes back to the pre-prune ("scratch") checkpoint
    sess_b, args_b = _session()
    res_b = sess_b.run_task('t1', 5, tr1, va1, accuracy_goal=0.0, finetune_epochs=2, prune_epochs=1, sparsities=(0.3,),
                            args=args_b, min_train_acc=-1.0, allow_acc_loss=-10.0)
    assert res_b.chosen_ratio == 0.0 and 0.3 in res_b.ratio_to_acc
    assert _zero_fraction(sess_b) == 0.0                                            # masks of the finetuned model: nothing released
    del sess_b

    # ---- task 2: piggymasks over task 1's weights, free slots trained, prune, piggymask retrain
    tr2, va2 = _loaders(1)
    res2 = sess.run_task('t2', 5, tr2, va2, accuracy_goal=0.0, finetune_epochs=2, prune_epochs=1, sparsities=(0.3,),
                         args=args, min_train_acc=-1.0, retrain_epochs=2)
    assert res2.chosen_ratio == 0.3 and res2.retrain_kept in (True, False) and res2.retrain_acc is not None
    for k, v in sess.masks.items():
        assert torch.equal(v == 1, own1[k]), k                                       # task 1's slots are still task 1's
        assert int((v == 2).sum()) > 0
    for n, m in sess.net.named_modules():
        if hasattr(m, 'piggymask'):
            keep = own1['module.' + n]
            assert torch.equal(m.weight.detach()[keep], w1[n][keep]), n             # ... and were not moved by a single ulp
            assert m.piggymask is not None
    acc1b, logits1b = sess.evaluate('t1', va1)
    assert acc1b == acc1
    for a, b in zip(logits1, logits1b):
        assert torch.equal(a, b), 'task-1 logits changed after learning task 2'
    acc2, _ = sess.evaluate('t2', va2)
    assert abs(acc2 - res2.retrain_acc) < 1e-6                                       # the kept model is the better of pruned / retrained
    assert sess.shared_layer_info['t1']['piggymask'] == {} and len(sess.shared_layer_info['t2']['piggymask']) == 15


def test_growth_pads_masks_keeps_old_weights_and_old_task_logits(tmp_path):
    from cpg_amd.utils import checkpoint as ckpt
    sess, args = _session(0.125)
    tr1, va1 = _loaders(0)
    sess.run_task('t1', 5, tr1, va1, accuracy_goal=0.0, finetune_epochs=2, prune_epochs=1, sparsities=(0.5,), args=args,
                  min_train_acc=-1.0)
    acc1, logits1 = sess.evaluate('t1', va1)
    old_masks = {k: v.clone() for k, v in sess.masks.items()}
    old_w = {k: v.detach().clone() for k, v in sess.net.state_dict().items()}
    # task 2 with an unreachable accuracy goal: exit code 2 -> widen by 0.125 (cap 0.25) and finetune again from task 1's state
    tr2, va2 = _loaders(1)
    res2 = sess.run_task('t2', 5, tr2, va2, accuracy_goal=2.0, finetune_epochs=1, prune_epochs=1, sparsities=(0.3,), args=args,
                         min_train_acc=-1.0, max_width_multiplier=0.0625, width_step=0.046875, retrain_epochs=1, total_num_tasks=2)   # raw 1/64 -> 1/16: widths 0.125 -> 0.25
    assert res2.grown_to == [0.0625] and sess.width == 0.25 and sess.width_multiplier == 0.0625 and res2.needs_growth      # (goal 2.0 stays missed at the cap)
    assert sess.shared_layer_info['t1']['network_width_multiplier'] == 0.125
    assert sess.shared_layer_info['t2']['network_width_multiplier'] == 0.25
    for name, m in sess.model.named_modules():
        if hasattr(m, 'piggymask'):
            mk, o = sess.masks[name], old_masks[name]
            assert mk.shape == m.weight.shape and mk.shape != o.shape
            corner = tuple(slice(0, s) for s in o.shape)
            assert torch.equal(mk[corner] == 1, o == 1)                              # task 1's slots kept, in the corner
            outside = torch.ones_like(mk, dtype=torch.bool)
            outside[corner] = False
            assert int((mk[outside] == 1).sum()) == 0 and int((mk[outside] == 2).sum()) > 0     # new slots: free -> task 2
            k = name[len('module.'):] + '.weight'
            keep = (o == 1)
            assert torch.equal(m.weight.detach()[corner][keep], old_w[k][keep]), name
    acc1b, logits1b = sess.evaluate('t1', va1)                                       # evaluated at width 0.125: cropped net + masks
    assert acc1b == acc1
    for a, b in zip(logits1, logits1b):
        assert torch.equal(a, b), 'task-1 logits changed after growing the network'
    # ---- checkpoint round trip in the reference's file format, on the GPU
    from cpg_amd.utils.manager import Manager
    from cpg_amd.driver import default_args
    a = default_args(mode='inference', dataset='t2', network_width_multiplier=sess.width)
    mgr = Manager(a, sess.model, sess.shared_layer_info, sess.masks, None, va2, 0, 0)
    acc2 = mgr.validate(0)
    mgr.save_checkpoint(None, 0, str(tmp_path))
    state = torch.load(os.path.join(tmp_path, 'checkpoint-1.pth.tar'), map_location='cpu', weights_only=False)
    assert set(state) == {'model_state_dict', 'dataset_history', 'dataset2num_classes', 'masks', 'shared_layer_info'}
    assert state['dataset_history'] == ['t1', 't2']
    from cpg_amd.driver import CPGSession
    fresh = CPGSession('custom_vgg_cifar100', 0.25, device=DEV, seed=3)
    fresh.load(state)
    acc2b, _ = fresh.evaluate('t2', va2)
    assert acc2b == acc2
    acc1c, logits1c = fresh.evaluate('t1', va1)
    assert acc1c == acc1
    for a_, b_ in zip(logits1, logits1c):
        assert torch.equal(a_, b_), 'task-1 logits changed through a checkpoint round trip'


def test_serving_old_task_from_grown_network_skips_dead_channels():
    """One resident wide network for all tasks: task 1 (learnt at width 0.5) evaluated on the network grown to 1.0 for task 2.
    apply_mask kills every slot of task 2 -> half of every layer's output channels and the trailing half of its input channels
    are dead; the inference conv kernels skip them (tile / chunk skip counters) and the logits equal the cropped model's."""
    from cpg_amd.models.fused_bn import FusedSequential
    sess, args = _session(0.5)
    tr1, va1 = _loaders(0)
    sess.run_task('t1', 5, tr1, va1, accuracy_goal=0.0, finetune_epochs=1, prune_epochs=1, sparsities=(0.5,), args=args,
                  min_train_acc=-1.0)
    tr2, va2 = _loaders(1)
    sess.run_task('t2', 5, tr2, va2, accuracy_goal=2.0, finetune_epochs=1, prune_epochs=1, sparsities=(0.3,), args=args,
                  min_train_acc=-1.0, max_width_multiplier=1.0, width_step=0.75, retrain_epochs=1, total_num_tasks=2)   # raw 0.25 -> 1.0: widths 0.5 -> 1.0
    assert sess.width == 1.0
    acc_c, logits_c = sess.evaluate('t1', va1)                       # the reference's way: a cropped width-0.5 model
    FusedSequential.skip_log = []
    try:
        acc_w, logits_w = sess.evaluate('t1', va1, crop=False)       # the resident width-1.0 model
        log = [t.cpu().tolist() for t in FusedSequential.skip_log]
    finally:
        FusedSequential.skip_log = None
    assert acc_w == acc_c
    for a, b in zip(logits_c, logits_w):
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=1e-5, atol=1e-5 * float(a.abs().max()))
    assert log, 'no fused inference conv ran'
    skipped = sum(s for _, s in log)
    assert skipped > 0, log                                            # whole output tiles were skipped ...
    widths = [c for c, _ in log]
    # ... and every layer after the stem stopped at the last live input channel = the width-0.5 channel count
    names = [m for m in sess.net.features if hasattr(m, 'piggymask') and getattr(m, 'kernel_size', None) == (3, 3)]
    per_call = {m.in_channels: None for m in names}
    assert any(c > 0 and c <= max(per_call) // 2 for c in widths), (widths, sorted(per_call))


def test_prune_run_that_runs_out_of_candidates_is_dropped_like_the_reference_process_that_exits_2():
    """At a 0.1 target every task hands on a tenth of what it got: by task 3 the smallest layer of the narrow net (216 weights) holds two
    slots of the task and the rank k = round(ratio * candidates) is 0 -- utils/prune.py:38-42 prints "Not enough weights for pruning" and
    exits with code 2.  The reference's process dies without saving and bash carries on with what earlier runs recorded; the session does
    the same: the run leaves no stage, the model goes back to the pre-prune state, earlier tasks still answer bit-identically."""
    sess, args = _session()
    kept = {}
    for t in range(3):
        tr, va = _loaders(t)
        res = sess.run_task('t%d' % (t + 1), 5, tr, va, accuracy_goal=0.0, finetune_epochs=1, prune_epochs=1, sparsities=(0.1,), args=args,
                            min_train_acc=-1.0, retrain_epochs=1)
        kept['t%d' % (t + 1)] = (va, sess.evaluate('t%d' % (t + 1), va))
        for name, (v, (acc0, outs0)) in kept.items():
            acc, outs = sess.evaluate(name, v)
            assert acc == acc0 and all(torch.equal(a, b) for a, b in zip(outs0, outs)), name
    assert res.prune_exit2 == 0.1 and res.chosen_ratio == 0.0 and set(res.ratio_to_acc) == {0.0}
    assert int(sum((m == 0).sum() for m in sess.masks.values())) == 0                  # nothing of task 3 was released
