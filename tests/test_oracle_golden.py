"""Pins oracle/ (the CPU restatement) to outputs of the reference itself (tests/golden/*.npz)."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden
from oracle import ops
from oracle import net as onet


def test_binarizer_matches_reference():
    g = load_golden('binarizer')
    y = ops.binarize(g['x'], float(g['threshold']))
    np.testing.assert_array_equal(np.isnan(y), np.isnan(g['y']))
    np.testing.assert_array_equal(np.nan_to_num(y, nan=7.0), np.nan_to_num(g['y'], nan=7.0))
    # straight-through backward
    np.testing.assert_array_equal(g['grad_in'], g['grad_out'])


CONV_FILES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, 'conv_*.npz')))


@pytest.mark.parametrize('name', CONV_FILES)
def test_conv_matches_reference(name):
    g = load_golden(name)
    N, C, H, W, M, k, s, p, d, has_bias = [int(v) for v in g['cfg']]
    pm = g['pm'] if 'pm' in g.files else None
    b = g['b'] if has_bias else None
    y = ops.conv2d_forward(g['x'], g['w'], pm, b, s, p, d)
    np.testing.assert_allclose(y, g['y'], rtol=1e-5, atol=1e-5)
    r = ops.conv2d_backward(g['x'], g['w'], g['gy'], pm, bool(has_bias), s, p, d)
    np.testing.assert_allclose(r['gx'], g['gx'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(r['gw'], g['gw'], rtol=1e-5, atol=1e-5)
    if pm is not None:
        np.testing.assert_allclose(r['gpm'], g['gpm'], rtol=1e-5, atol=1e-5)
    if has_bias:
        np.testing.assert_allclose(r['gb'], g['gb'], rtol=1e-5, atol=1e-5)


LIN_FILES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, 'linear_*.npz')))


@pytest.mark.parametrize('name', LIN_FILES)
def test_linear_matches_reference(name):
    g = load_golden(name)
    pm = g['pm'] if 'pm' in g.files else None
    y = ops.linear_forward(g['x'], g['w'], pm, g['b'])
    np.testing.assert_allclose(y, g['y'], rtol=1e-5, atol=1e-6)
    r = ops.linear_backward(g['x'], g['w'], g['gy'], pm)
    for key in ('gx', 'gw', 'gb') + (('gpm',) if pm is not None else ()):
        np.testing.assert_allclose(r[key], g[key], rtol=1e-5, atol=1e-6)


ROUTE_FILES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, 'route_*.npz')))


@pytest.mark.parametrize('name', ROUTE_FILES)
def test_route_matches_reference(name):
    g = load_golden(name)
    mode = str(g['mode'])
    for layer in ('conv', 'fc'):
        gpm_in = g['gpm_in_' + layer] if ('gpm_in_' + layer) in g.files else None
        gw, gpm = ops.route_grads(g['gw_in_' + layer], g['w_' + layer], g['owner_' + layer], int(g['cur']),
                                  float(g['wd']), gpm_in, mode)
        # masks (zero pattern) exact, values to 1 ulp (torch's add_ may or may not fuse the multiply-add)
        np.testing.assert_array_equal(gw == 0, g['gw_out_' + layer] == 0)
        np.testing.assert_allclose(gw, g['gw_out_' + layer], rtol=3e-7, atol=0)
        if gpm_in is not None:
            np.testing.assert_array_equal(gpm, g['gpm_out_' + layer])


RANK_CASES = ['rand_t1', 'multi_t2', 'multi_t3', 'ties', 'round_2p5', 'round_1p5', 'k_zero', 'no_cand',
              'all', 'special', 'layer']


@pytest.mark.parametrize('tag', RANK_CASES)
def test_rank_prune_matches_reference(tag):
    g = load_golden('rank_prune')
    w, owner = g[tag + '_w'], g[tag + '_owner']
    cur, ratio, status = int(g[tag + '_cur']), float(g[tag + '_ratio']), int(g[tag + '_status'])
    if status == 2:
        with pytest.raises(ops.NotEnoughWeights):
            ops.rank_prune(w, owner, cur, ratio)
        return
    out, k, cutoff = ops.rank_prune(w, owner, cur, ratio)
    np.testing.assert_array_equal(out, g[tag + '_out'])


def test_schedule_matches_reference():
    tab = load_golden('schedule')['table']
    last = {}
    for begin, end, freq, init, target, step, upd, ratio in tab:
        key = (begin, end, freq, init, target)
        lp = last.setdefault(key, begin)
        got = ops.time_to_update(step, begin, end, lp, freq)
        assert int(got) == int(upd), (key, step)
        if got:
            last[key] = step
        assert ops.adjust_sparsity(step, begin, end, init, target) == ratio   # bit-exact fp64


def test_stats_and_mask_ops_match_reference():
    g = load_golden('stats_masks')
    for i in range(3):
        t = 'case%d_' % i
        owners = [g[t + 'owner_conv'], g[t + 'owner_fc']]
        pms = [g[t + 'pm_conv'], g[t + 'pm_fc']]
        idx, width = int(g[t + 'inference_idx']), float(g[t + 'width'])
        assert ops.sparsity(owners, idx) == float(g[t + 'sparsity'])
        assert ops.curr_task_ratio(owners, idx, width) == float(g[t + 'curr_task_ratio'])
        assert ops.zero_ratio(owners, width) == float(g[t + 'zero_ratio'])
        assert ops.shared_part_ratio(owners, pms, idx) == float(g[t + 'shared_part_ratio'])
        for layer in ('conv', 'fc'):
            np.testing.assert_array_equal(ops.apply_mask(g[t + 'w_' + layer], g[t + 'owner_' + layer], idx), g[t + 'applied_' + layer])
            np.testing.assert_array_equal(ops.zero_pruned(g[t + 'w_' + layer], g[t + 'owner_' + layer]), g[t + 'zeroed_' + layer])
            np.testing.assert_array_equal(ops.claim_free(g[t + 'owner_' + layer], int(g[t + 'claimed_idx'])), g[t + 'claimed_' + layer])
    two = [np.full((3, 3), 2, np.uint8)]
    assert ops.sparsity(two, 1) == float(g['empty_sparsity']) == 0.0
    assert ops.shared_part_ratio(two, [np.ones((3, 3), np.float32)], 1) == float(g['empty_shared']) == 0.0


@pytest.mark.parametrize('variant,fixture', [('cifar100', 'first_forward_vgg_cifar100'), ('imagenet', 'first_forward_vgg')])
def test_oracle_vgg_init_and_forward(variant, fixture):
    """Same seed + same init order => same weights and logits as the reference topology."""
    g = load_golden(fixture)
    torch.manual_seed(1)
    m = onet.OracleVGG(float(g['width']), variant)
    m.add_dataset('t1', int(g['num_classes']))
    m.set_dataset('t1')
    digest = np.array([[float(p.double().sum()), float(p.double().abs().sum())] for p in m.parameters()])
    np.testing.assert_allclose(digest, g['param_digest'], rtol=1e-12, atol=1e-12)
    m.eval()
    with torch.no_grad():
        y = m(torch.from_numpy(g['x'])).numpy()
    np.testing.assert_allclose(y, g['y'], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize('mode', ['prune', 'finetune'])
def test_oracle_trajectory_matches_reference(mode):
    g = load_golden('trajectory_' + mode)
    model, pruner, opt = onet.make_task1(float(g['width']), 'cifar100', mode, lr=float(g['lr']),
                                         begin=int(g['begin']), end=int(g['end']), frequency=int(g['freq']),
                                         initial=float(g['initial']), target=float(g['target']), wd=float(g['wd']))
    # start from the fixture's initial state (also checks that seed-1 init reproduced it)
    sd = model.state_dict()
    for k in sd:
        ref = g['init/' + k.replace('head.', 'classifier.')] if ('init/' + k.replace('head.', 'classifier.')) in g.files else None
        if ref is not None and sd[k].dtype.is_floating_point:
            np.testing.assert_array_equal(sd[k].numpy(), ref, err_msg=k)
    model.train()
    xs, ts = torch.from_numpy(g['x']), torch.from_numpy(g['t'])
    for s in range(xs.shape[0]):
        out, loss, ratio = onet.train_step(model, pruner, opt, xs[s], ts[s], prune_step=s)
        np.testing.assert_allclose(out.numpy(), g['logits'][s], rtol=1e-4, atol=1e-6, err_msg='step %d' % s)
        assert abs(loss - g['losses'][s]) < 1e-5
        if mode == 'prune':
            assert ratio == g['ratios'][s]
        assert abs(pruner.sparsity() - g['sparsities'][s]) < 1e-12
    for n, _ in model.masked_layers():
        np.testing.assert_array_equal(pruner.owners[n], g['mask/module.' + n], err_msg=n)
    pruner.apply_mask()
    model.eval()
    with torch.no_grad():
        ev = model(xs[0]).numpy()
    np.testing.assert_allclose(ev, g['eval_logits'], rtol=1e-4, atol=1e-6)


def test_one_shot_prune_matches_reference():
    """utils/prune.py:94-109: rank prune at a fixed ratio, then zero the released weights."""
    g = load_golden('one_shot_prune')
    for i in range(2):
        t = 'case%d_' % i
        for layer in ('conv', 'fc'):
            out, _, _ = ops.rank_prune(g[t + 'w_' + layer], g[t + 'owner_' + layer], int(g[t + 'cur']), float(g[t + 'perc']))
            np.testing.assert_array_equal(out, g[t + 'mask_' + layer])
            np.testing.assert_array_equal(ops.zero_pruned(g[t + 'w_' + layer], out), g[t + 'wout_' + layer])


@pytest.mark.parametrize('mode', ['finetune', 'prune'])
def test_oracle_matches_reference_manager_train_and_validate(mode):
    """The fixture was produced by the reference's OWN Manager.train + Manager.validate (utils/manager.py:39-152); the
    oracle's step order, its validate (apply_mask first, weights left mutated) and the returned accuracies must agree."""
    g = load_golden('manager_' + mode)
    model, pruner, opt = onet.make_task1(float(g['width']), 'cifar100', mode, lr=float(g['lr']), begin=int(g['begin']),
                                         end=int(g['end']), frequency=int(g['freq']), initial=float(g['initial']),
                                         target=float(g['target']), wd=float(g['wd']))
    sd = model.state_dict()
    for k in sd:
        if sd[k].dtype.is_floating_point:
            np.testing.assert_array_equal(sd[k].numpy(), g['init/' + k.replace('head.', 'classifier.')], err_msg=k)
    model.train()
    xs, ts = torch.from_numpy(g['x']), torch.from_numpy(g['t'])
    correct = 0
    for s in range(xs.shape[0]):
        out, loss, ratio = onet.train_step(model, pruner, opt, xs[s], ts[s], prune_step=s)
        np.testing.assert_allclose(out.numpy(), g['logits'][s], rtol=1e-4, atol=1e-6, err_msg='step %d' % s)
        correct += int((out.argmax(1) == ts[s]).sum())
    assert abs(correct / ts.numel() - float(g['train_acc'])) < 1e-6
    for n, m in model.masked_layers():
        np.testing.assert_array_equal(pruner.owners[n], g['mask/module.' + n], err_msg=n)
        np.testing.assert_allclose(m.weight.detach().numpy(), g['pre/' + n + '.weight'], rtol=1e-4, atol=1e-7, err_msg=n)
    # validate from the reference's pre-validate state: identical inputs -> tight comparison
    model.load_state_dict({k.replace('classifier.', 'head.'): torch.from_numpy(g['pre/' + k.replace('head.', 'classifier.')])
                           for k in sd}, strict=True)
    pruner.apply_mask()
    for n, m in model.masked_layers():
        np.testing.assert_array_equal(m.weight.detach().numpy(), g['post/' + n + '.weight'], err_msg=n)
    model.eval()
    xv, tv = torch.from_numpy(g['xv']), torch.from_numpy(g['tv'])
    with torch.no_grad():
        ev = torch.stack([model(xv[i]) for i in range(xv.shape[0])])
    np.testing.assert_allclose(ev.numpy(), g['eval_logits'], rtol=1e-5, atol=1e-6)
    assert abs(float((ev.argmax(2) == tv).float().mean()) - float(g['val_acc'])) < 1e-6
    assert abs(pruner.sparsity() - float(g['sparsity'])) < 1e-12


# --------------------------------------------------------------------------- round 5: the ResNet-50 / SphereNet-20 restatements
def _oracle_net(arch, width, ncls, dataset='t1', he_seed=None):
    torch.manual_seed(1)
    m = onet.OracleResNet(width) if arch == 'resnet50' else onet.OracleSphereNet(width)
    m.add_dataset(dataset, ncls)
    m.set_dataset(dataset)
    if he_seed is not None:                       # make_golden.reinit_resnet: He-normal drawn in module order
        torch.manual_seed(he_seed)
        for mod in m.modules():
            if isinstance(mod, onet.MaskedConv):
                torch.nn.init.kaiming_normal_(mod.weight, mode='fan_out', nonlinearity='relu')
    return m


@pytest.mark.parametrize('arch,fixture,width,he', [('resnet50', 'first_forward_resnet50', 0.25, None), ('spherenet20', 'first_forward_spherenet20', 0.25, None),
                                                   ('resnet50', 'full_width_logits_resnet50', 1.0, 2), ('spherenet20', 'full_width_logits_spherenet20', 1.0, None)])
def test_oracle_resnet_spherenet_init_and_forward(arch, fixture, width, he):
    """OracleResNet / OracleSphereNet (models/resnet.py:60-222, models/spherenet.py:101-251) against the reference's own
    seed-1 initialisation (sum / abs-sum per parameter) and eval-mode logits, narrow and at width 1.0."""
    g = load_golden(fixture)
    m = _oracle_net(arch, width, int(g['num_classes']), he_seed=he)
    digest = np.array([[float(p.detach().double().sum()), float(p.detach().double().abs().sum())] for p in m.parameters()])
    np.testing.assert_allclose(digest, g['param_digest'], rtol=1e-12, atol=0)
    for name, mod in m.named_modules():
        if isinstance(mod, torch.nn.BatchNorm2d) and 'bn_mean/' + name in g.files:
            mod.running_mean.copy_(torch.from_numpy(g['bn_mean/' + name]))
            mod.running_var.copy_(torch.from_numpy(g['bn_var/' + name]))
    m.eval()
    with torch.no_grad():
        y = m(torch.from_numpy(g['x'])).numpy()
    np.testing.assert_allclose(y, g['y'], rtol=1e-5, atol=1e-6 * max(1.0, float(np.abs(g['y']).max())))


@pytest.mark.parametrize('arch', ['resnet50', 'spherenet20'])
def test_oracle_train_steps_match_reference(arch):
    """Three TRAIN-mode steps of the narrow ResNet-50 / SphereNet-20 (+ AngleLinear head, AngleLoss) through the oracle's own pieces --
    forward, loss, backward, OraclePruner.route, SGD-nesterov, rank-prune events -- against what the reference's modules produced
    (train_steps_*.npz): logits and losses of every step, the raw gradients of the four watched convs, the prune ratios, the owner
    masks and the final weights.  Same torch-CPU ops in the same order: held to 1e-5."""
    g = load_golden('train_steps_' + arch)
    width, ncls = float(g['width']), int(g['num_classes'])
    dataset = 'face_verification' if arch == 'spherenet20' else 't1'
    m = _oracle_net(arch, width, ncls, dataset, he_seed=2 if arch == 'resnet50' else None)
    digest = np.array([[float(p.detach().double().sum()), float(p.detach().double().abs().sum())] for p in m.parameters()])
    np.testing.assert_allclose(digest, g['param_digest'], rtol=1e-12, atol=1e-12)
    owners = {n: np.ones(tuple(l.weight.shape), np.uint8) for n, l in m.masked_layers()}
    pruner = onet.OraclePruner(m, owners, 'prune', 1, 1, 0, 2, 1, 0.0, 0.3, float(g['wd']), width)
    opt = torch.optim.SGD(list(m.parameters()), lr=float(g['lr']), momentum=0.9, nesterov=True, weight_decay=0.0)
    crit = onet.OracleAngleLoss() if dataset == 'face_verification' else torch.nn.CrossEntropyLoss()
    xs, ts = torch.from_numpy(g['x']), torch.from_numpy(g['t'])
    watch = [str(w) for w in g['watch']]
    mods = dict(m.named_modules())
    m.train()
    for s in range(xs.shape[0]):
        opt.zero_grad()
        out = m(xs[s])
        loss = crit(out, ts[s])
        loss.backward()
        o1 = out[0] if isinstance(out, tuple) else out
        sc = float(np.abs(g['logits'][s]).max())
        np.testing.assert_allclose(o1.detach().numpy(), g['logits'][s], rtol=1e-5, atol=1e-5 * sc, err_msg='logits step %d' % s)
        if isinstance(out, tuple):
            np.testing.assert_allclose(out[1].detach().numpy(), g['logits2'][s], rtol=1e-5, atol=1e-5 * float(np.abs(g['logits2'][s]).max()))
        assert abs(float(loss.detach()) - g['losses'][s]) <= 1e-5 * max(1.0, abs(g['losses'][s]))
        for n in watch:
            ref = g['grad/' + n][s]
            np.testing.assert_allclose(mods[n].weight.grad.numpy(), ref, rtol=1e-5, atol=1e-5 * float(np.abs(ref).max()), err_msg='grad %s step %d' % (n, s))
        pruner.route()
        opt.step()
        ratio = pruner.gradually_prune(s)
        assert ratio == float(g['ratios'][s])
    names = [n for n, _ in m.masked_layers()]
    assert [int((pruner.owners[n] == 0).sum()) for n in names] == [int(v) for v in g['mask_zero_counts']]
    for n in watch:
        np.testing.assert_array_equal(pruner.owners[n], g['mask/module.' + n])
        np.testing.assert_allclose(mods[n].weight.detach().numpy(), g['final/' + n], rtol=1e-5, atol=1e-7)
    assert abs(pruner.sparsity() - float(g['sparsity'])) < 1e-12


# --------------------------------------------------------------------------- round 6: configs[3] / configs[4] as SEQUENCES
def _oracle_phase(m, owners, mode, task_idx, train, val, lr, lr_mask, target, wd, width, dataset):
    """One phase of a task >= 2 on the oracle: the set-up of CPG_imagenet_main.py:320-346 / CPG_face_main.py:318-345 (SGD-nesterov over the
    shared tensors + the task's head, Adam over the piggymasks), Manager's loss choice (utils/manager.py:29-36) and its train loop."""
    cur = task_idx + 1
    pruner = onet.OraclePruner(m, owners, mode, cur - 1 if mode == 'finetune' else cur, cur, 0, len(train), 1, 0.0, target, wd, width)
    head = m.classifiers[task_idx]
    sgd = [p for n, p in m.named_parameters() if 'classifiers' not in n and 'piggymask' not in n] + list(head.parameters())
    adam = [l.piggymask for _, l in m.masked_layers()]
    opts = [torch.optim.SGD(sgd, lr=lr, momentum=0.9, nesterov=True, weight_decay=0.0), torch.optim.Adam(adam, lr=lr_mask)]
    if dataset == 'face_verification':
        crit = onet.OracleAngleLoss()
    elif dataset == 'emotion':
        counts = torch.tensor([74874, 134415, 25459, 14090, 6378, 3803, 24882], dtype=torch.float32)
        crit = torch.nn.CrossEntropyLoss(weight=(counts.sum() - counts) / counts)
    else:
        crit = torch.nn.CrossEntropyLoss()
    pre = None
    if mode == 'finetune':
        pruner.claim_free()
    else:
        pruner.apply_mask()                                   # the "Before pruning:" validate
        m.eval()
        with torch.no_grad():
            pre = torch.stack([m(x) for x, _ in val])
    logits, losses = [], []
    m.train()
    for s, (x, t) in enumerate(train):
        for o in opts:
            o.zero_grad()
        out = m(x)
        loss = crit(out, t)
        loss.backward()
        pruner.route()
        for o in opts:
            o.step()
        if mode == 'prune':
            pruner.gradually_prune(s)
        logits.append(out.detach())
        losses.append(float(loss.detach()))
    pruner.apply_mask()
    m.eval()
    with torch.no_grad():
        ev = torch.stack([m(x) for x, _ in val])
    return torch.stack(logits), np.array(losses), ev, pre, pruner


@pytest.mark.parametrize('arch', ['resnet50', 'spherenet20'])
def test_oracle_sequence_tasks_after_the_first(arch):
    """The oracle through every phase AFTER task 1 of the reference-run sequences (sequence_*.npz: ResNet-50 imagenet -> cubs_cropped;
    SphereNet-20 face_verification -> gender -> emotion), starting from the reference's own task-1 checkpoint: finetune with piggymasks
    (Adam) over task 1's frozen weights, the head / loss switch (nn.Linear + CE; class-weighted CE for `emotion`), the prune run with its
    'before pruning' validate.  Same torch-CPU ops in the same order as the reference's processes: per-step logits and losses 1e-5,
    owner histograms and piggymask-off counts equal."""
    import _sequence as sq
    fx = sq.load(arch)
    width, wd, lr_mask = float(fx['width']), float(fx['wd']), float(fx['lr_mask'])
    names = sq.tasks(fx)
    data = sq.batches(fx)
    torch.manual_seed(1)
    m = onet.OracleResNet(width) if arch == 'resnet50' else onet.OracleSphereNet(width)
    m.add_dataset(*names[0])
    missing, unexpected = m.load_state_dict(sq.group(fx, 't2start'), strict=False)
    assert not unexpected and all(k.startswith('classifiers.0') for k in missing), (missing, unexpected)
    owners = {n: sq.group(fx, 't2start_mask')['module.' + n].numpy().copy() for n, _ in m.masked_layers()}
    layer_names = [n for n, _ in m.masked_layers()]
    for ti in range(1, len(names)):
        dataset, ncls = names[ti]
        m.add_dataset(dataset, ncls)
        m.set_dataset(dataset)
        m.classifiers[ti].load_state_dict(sq.group(fx, 'head_init/%d' % ti))
        for _, l in m.masked_layers():
            l.piggymask = torch.nn.Parameter(torch.full(l.weight.shape, 0.01))       # (CPG_face_main.py:263-270)
        train, val = data[ti]
        for mode, lr, target in (('finetune', float(fx['lr_finetune'][ti]), 0.0), ('prune', float(fx['lr_prune'][ti]), float(fx['targets'][ti]))):
            tag = 't%d_%s' % (ti + 1, mode)
            logits, losses, ev, pre, pruner = _oracle_phase(m, owners, mode, ti, train, val, lr, lr_mask if mode == 'finetune' else 0.0,
                                                            target, wd, width, dataset)
            sc = float(np.abs(fx[tag + '/logits']).max())
            np.testing.assert_allclose(logits.numpy(), fx[tag + '/logits'], rtol=1e-5, atol=2e-5 * sc, err_msg=tag)
            np.testing.assert_allclose(losses, fx[tag + '/losses'], rtol=2e-5, atol=1e-5, err_msg=tag)
            np.testing.assert_allclose(ev.numpy(), fx[tag + '/val'], rtol=1e-5, atol=2e-5 * float(np.abs(fx[tag + '/val']).max()), err_msg=tag)
            if pre is not None:
                np.testing.assert_allclose(pre.numpy(), fx[tag + '/pre_val'], rtol=1e-5, atol=2e-5 * float(np.abs(fx[tag + '/pre_val']).max()))
            hist = np.array([[int((pruner.owners[n] == k).sum()) for k in range(5)] for n in layer_names])
            np.testing.assert_array_equal(hist, fx[tag + '/owner_hist'], err_msg=tag)
            mods = dict(m.named_modules())
            off = np.array([int((mods[n].piggymask.detach() <= 0.005).sum()) for n in layer_names])
            np.testing.assert_array_equal(off, fx[tag + '/pm_off'], err_msg=tag)
            assert abs(pruner.sparsity() - float(fx[tag + '/stats'][0])) < 1e-12
