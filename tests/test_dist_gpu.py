"""The data-parallel composition on the GPU: HIP model + cpg_amd.dist.DataParallel + MaskedSGD + rank prune.

A GPU box handed to the tests has ONE MI355X, so two ranks share it and talk through gloo (RCCL needs one device per
rank); everything else is the product path: the masked conv / linear HIP kernels, the gradient hooks and
finish_gradient_sync() inside Manager.train, the fused routing + SGD step, cpg_rank_prune.  Expected:

  * both ranks end with bit-identical weights and owner masks (replicated state, no collective besides the gradients);
  * they equal ONE process training on the full batch -- BatchNorm is frozen in eval mode here, because per-replica batch
    statistics (nn.DataParallel semantics, kept on purpose) would make a sharded run differ from a full-batch run.
"""
import os
import socket
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VGG_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']
WIDTH, B, STEPS = 0.125, 8, 3


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _batches():
    g = torch.Generator().manual_seed(77)
    return [(torch.randn(B, 3, 32, 32, generator=g), torch.randint(0, 5, (B,), generator=g)) for _ in range(STEPS)]


def _run(rank, world, data_parallel, scenario='task1_prune', dev='cuda:0'):
    """scenario 'task1_prune': 3 prune-mode steps (rank-prune events after steps 1 and 2) through Manager.train.
    scenario 'task2_finetune': task 2 of two -- 70 % of the slots belong to task 1 (frozen, picked through piggymasks), the free
    ones are claimed and trained; MaskedSGD + MaskedAdam; the data-parallel exchange sends only the slots that survive routing.
    Returns (weights, masks, wrapper active, payload of the last step)."""
    import torch.nn as nn
    import cpg_amd.models as M
    from cpg_amd import dist as cdist
    from cpg_amd.models import layers as nl
    from cpg_amd.utils import Optimizers
    from cpg_amd.utils.fused_sgd import MaskedSGD
    from cpg_amd.utils.manager import Manager
    torch.manual_seed(1)
    net = M.custom_vgg_cifar100(VGG_CFG, dataset_history=[], dataset2num_classes={}, network_width_multiplier=WIDTH, shared_layer_info={})
    net.add_dataset('t1', 5)
    net.set_dataset('t1')
    net = net.to(dev)
    for m in net.modules():                       # frozen BatchNorm: .train() from Manager.train must not re-enable batch statistics
        if isinstance(m, nn.BatchNorm2d):
            m.eval()
            m.train = lambda mode=True, _m=m: _m
    model = cdist.DataParallel(net, large_numel=1 << 12) if data_parallel else _Wrap(net)
    loader = []
    for x, t in _batches():
        xs, ts = cdist.shard_batch(x, t, rank, world)
        loader.append((xs.to(dev), ts.to(dev)))
    names = [(n, m) for n, m in model.named_modules() if isinstance(m, (nl.SharableConv2d, nl.SharableLinear))]
    if scenario == 'task1_prune':
        masks = {n: torch.ones(m.weight.shape, dtype=torch.uint8, device=dev) for n, m in names}
        args = types.SimpleNamespace(mode='prune', dataset='t1', finetune_again=False, target_sparsity=0.3, initial_sparsity=0.0,
                                     pruning_frequency=1, weight_decay=4e-5, network_width_multiplier=WIDTH, cuda=True, log_path=None,
                                     progress=False)
        mgr = Manager(args, model, {}, masks, loader, None, 0, 2)
        opts = Optimizers()
        opts.add(MaskedSGD(list(model.parameters()), pruner=mgr.pruner, lr=1e-2, momentum=0.9, nesterov=True), 1e-2)
        mgr.train(opts, 0, [1e-2], 0)
        assert mgr.pruner.prune_events == 2
    else:
        from torch.nn.parameter import Parameter
        from cpg_amd.utils.fused_sgd import MaskedAdam
        net.add_dataset('t2', 5)
        net.set_dataset('t2')
        net.classifiers.to(dev)
        g = torch.Generator().manual_seed(5)
        masks = {n: (torch.rand(m.weight.shape, generator=g) < 0.7).to(torch.uint8).to(dev) for n, m in names}      # 1 = task 1, 0 = free
        for n, m in names:
            m.piggymask = Parameter(torch.full_like(m.weight.detach(), 0.01))
        if hasattr(model, 'refresh_hooks'):
            model.refresh_hooks()
            model.compact_below = 0.9                      # also compact the piggymask gradients (70 % of their slots survive)
        args = types.SimpleNamespace(mode='finetune', dataset='t2', finetune_again=False, target_sparsity=0.3, initial_sparsity=0.0,
                                     pruning_frequency=1, weight_decay=4e-5, network_width_multiplier=WIDTH, cuda=True, log_path=None,
                                     progress=False)
        mgr = Manager(args, model, {}, masks, loader, None, 0, 0)
        mgr.pruner.make_finetuning_mask()                  # free slots -> task 2
        assert mgr.pruner.current_dataset_idx == 2
        sgd = [p for n, p in model.named_parameters() if 'piggymask' not in n and 'classifiers.0.' not in n]
        adam = [p for n, p in model.named_parameters() if 'piggymask' in n]
        opts = Optimizers()
        opts.add(MaskedSGD(sgd, pruner=mgr.pruner, lr=1e-2, momentum=0.9, nesterov=True), 1e-2)
        opts.add(MaskedAdam(adam, pruner=mgr.pruner, lr=5e-4), 5e-4)
        mgr.train(opts, 0, [1e-2, 5e-4], 0)
    torch.cuda.synchronize()
    return ({k: v.detach().cpu() for k, v in net.state_dict().items()}, {k: v.cpu() for k, v in masks.items()},
            bool(getattr(model, '_active', False)), dict(getattr(model, 'last_payload', {})))


class _Wrap(torch.nn.Module):
    def __init__(self, m):
        super().__init__()
        self.module = m

    def forward(self, x):
        return self.module(x)


def _worker(rank, world, port, out_dir, scenario, backend='gloo'):
    """backend 'gloo': every rank on cuda:0 (a one-GPU box); 'nccl' (= RCCL): rank r on cuda:r (tests/test_dist_rccl.py)."""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dev = 'cuda:0'
    if backend == 'nccl':
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dev = 'cuda:%d' % rank
        torch.cuda.set_device(rank)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        sd, masks, active, payload = _run(rank, world, True, scenario, dev)
        assert active
        torch.save({'sd': sd, 'masks': masks, 'payload': payload}, os.path.join(out_dir, 'rank%d.pt' % rank))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _two_ranks(tmp_path, scenario, world=2, backend='gloo'):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), scenario, backend), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, 'rank0.pt'))
    for r in range(1, world):
        r1 = torch.load(os.path.join(tmp_path, 'rank%d.pt' % r))
        for k in r0['sd']:
            assert torch.equal(r0['sd'][k], r1['sd'][k]), 'ranks 0 and %d diverged in %s' % (r, k)
        for k in r0['masks']:
            assert torch.equal(r0['masks'][k], r1['masks'][k]), 'ranks 0 and %d diverged in mask %s' % (r, k)
    return r0


def check_task2_against_single_process(r0):
    pay = r0['payload']
    assert 0 < pay['sent_elems'] < 0.75 * pay['dense_elems'], pay          # ~30 % of the weight slots + ~70 % of the piggymask slots
    sd, masks, _, _ = _run(0, 1, False, 'task2_finetune')
    for k, v in sd.items():
        if v.dtype.is_floating_point:
            a, b = r0['sd'][k].numpy(), v.numpy()
            np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-6 * (float(np.abs(b).max()) + 1e-30), err_msg=k)
    for k in masks:
        assert torch.equal(r0['masks'][k], masks[k]), k


def test_two_ranks_task2_piggymasks_compacted_gradient_exchange(tmp_path):
    """Task 2 under data parallelism: only the gradient slots that survive routing are exchanged (cpg_pack_owned), and the
    result still equals one process on the full batch -- weights of task 1 untouched, task-2 slots and piggymasks trained."""
    check_task2_against_single_process(_two_ranks(tmp_path, 'task2_finetune'))


def check_task1_against_single_process(r0):
    sd, masks, _, _ = _run(0, 1, False)                    # one process, the full batch
    for k, v in sd.items():
        if v.dtype.is_floating_point:
            a, b = r0['sd'][k].numpy(), v.numpy()
            np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-6 * (float(np.abs(b).max()) + 1e-30), err_msg=k)
    mism = sum(int((r0['masks'][k] != masks[k]).sum()) for k in masks)
    total = sum(v.numel() for v in masks.values())
    assert mism <= max(2, 1e-4 * total), 'owner masks differ from the single-process run in %d of %d slots' % (mism, total)
    released = sum(int((v == 0).sum()) for v in masks.values())
    assert released > 0.05 * total                         # the prune events really released weights


def test_two_ranks_hip_model_masked_sgd_prune_match_single_process(tmp_path):
    check_task1_against_single_process(_two_ranks(tmp_path, 'task1_prune'))


def _driver_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from cpg_amd.driver import CPGSession, default_args
        dev = 'cuda:0'

        def data(seed, n=8):
            g = torch.Generator().manual_seed(seed)
            t = torch.randint(0, 5, (n,), generator=g)
            x = 0.5 * torch.randn(n, 3, 32, 32, generator=g)
            for i in range(n):
                c = int(t[i])
                x[i, c % 3, (c * 5) % 16:(c * 5) % 16 + 12, (c * 6) % 20:(c * 6) % 20 + 12] += 2.0
            return x.to(dev), t.to(dev)
        # every rank its OWN shard of the training batches (rank-dependent seeds): the local train accuracies differ
        train = [data(100 + 10 * i + rank) for i in range(4)]
        val = [data(900 + i) for i in range(2)]
        sess = CPGSession('custom_vgg_cifar100', 0.125, device=dev, seed=1)
        args = default_args(lr=5e-2, lr_mask=5e-4, pruning_frequency=1, pruning_interval=1, prune_lr=1e-2)
        # an unreachable goal at the first width forces the grow branch; min_train_acc sits where shard accuracies can straddle it
        res = sess.run_task('t1', 5, train, val, accuracy_goal=2.0, finetune_epochs=1, prune_epochs=1, sparsities=(0.2, 0.4), args=args,
                            min_train_acc=0.3, max_width_multiplier=0.0625, width_step=0.046875)      # raw 1/64 -> 1/16: widths 0.125 -> 0.25
        torch.cuda.synchronize()
        torch.save({'res': {'grown_to': res.grown_to, 'ratio_to_acc': res.ratio_to_acc, 'chosen_ratio': res.chosen_ratio,
                            'finetune_train_acc': res.finetune_train_acc, 'finetune_acc': res.finetune_acc},
                    'sd': {k: v.detach().cpu() for k, v in sess.net.state_dict().items()},
                    'masks': {k: v.cpu() for k, v in sess.masks.items()}}, os.path.join(out_dir, 'drv_rank%d.pt' % rank))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_driver_decisions_are_rank_identical(tmp_path):
    """CPGSession.run_task branches on the train / validation accuracies (grow, stop the sweep, choose the ratio).
    Under data parallelism every rank must take the SAME branch -- Manager returns metrics over the global batches -- or the ranks'
    collectives mismatch (a hang) and their networks get different widths.  Two ranks with different shards run a task that grows."""
    import torch.multiprocessing as mp
    mp.spawn(_driver_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, 'drv_rank0.pt'))
    r1 = torch.load(os.path.join(tmp_path, 'drv_rank1.pt'))
    assert r0['res'] == r1['res'], (r0['res'], r1['res'])
    assert r0['res']['grown_to'] == [0.0625]                      # the goal was missed at width 0.125: both ranks widened once
    for k in r0['sd']:
        assert torch.equal(r0['sd'][k], r1['sd'][k]), 'ranks diverged in %s' % k
    for k in r0['masks']:
        assert torch.equal(r0['masks'][k], r1['masks'][k]), 'ranks diverged in mask %s' % k


def test_bench_self_launches_two_ranks(tmp_path):
    """`python bench.py --gpus 2` started WITHOUT torch.distributed.run must become two ranks by itself (gloo here: the box
    has one GPU) and print n_gpus = 2 with the multi_gpu block; a world size that contradicts --gpus is refused."""
    import json
    import subprocess
    env = dict(os.environ, CPG_BENCH_BACKEND='gloo')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--batch', '4',
                        '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith('{')][-1]
    out = json.loads(line)
    assert out['n_gpus'] == 2 and out['config']['global_batch'] == 8 and out['config']['parallelism'] == 'dp2'
    mg = out['multi_gpu']
    assert mg['backend'] == 'gloo' and len(mg['per_rank_ms_per_step']) == 2 and mg['replicas_identical_after_cycle'] is True
    assert mg['allreduced_gradient_bytes_per_step'] > 500e6
    # the messages of a step in launch order and the step time a hardware run should show
    kinds = [b['kind'] for b in mg['buckets']]
    assert kinds.count('chunk') == 4 and kinds[-1] == 'coalesced', kinds       # features.45 goes out in 4 row blocks; small tensors last
    assert kinds.index('chunk') < kinds.index('tensor') or 'tensor' in kinds[:2]  # (backward order: the classifier's layers first)
    assert abs(sum(b['bytes'] for b in mg['buckets']) - mg['allreduced_gradient_bytes_per_step']) < 1e6
    assert mg['predicted_ms_per_step'] > mg['single_gpu_ms_per_step'] > 0 and 0 < mg['exposed_ms'] < 5.0
    bad = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1'],
                         env=dict(env, WORLD_SIZE='1', RANK='0'), capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and 'WORLD_SIZE' in (bad.stderr + bad.stdout)
    # the reference's own split (the batch is NOT scaled with the GPUs, CPG_cifar100_main_normal.py:112-114,199): --global-batch
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--global-batch', '8',
                        '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith('{')][-1])
    assert out['scaling'] == 'strong' and out['config']['global_batch'] == 8 and out['config']['per_gpu_batch'] == 4
    odd = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--global-batch', '7'], env=env, capture_output=True,
                         text=True, timeout=300)
    assert odd.returncode != 0 and 'divisible' in (odd.stderr + odd.stdout)


def test_bench_json_line_is_the_last_stdout_line_with_rccl(tmp_path):
    """With a process group on RCCL (here: world 1, CPG_DP_FORCE=1 -- every hook and collective of the N > 1 path) RCCL's version banner
    sits in the C stdio buffer of a piped stdout until the process ends; bench.py must still END its stdout with the one JSON line."""
    import json
    import subprocess
    env = dict(os.environ, CPG_DP_FORCE='1', RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1', '--batch', '4',
                        '--no-cpu-baseline', '--optin-steps', '0'], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert lines[-1].startswith('{'), lines[-5:]
    out = json.loads(lines[-1])
    assert out['n_gpus'] == 1 and out['steps'] == 3 and 'roofline' in out


@pytest.mark.parametrize('arch', ['resnet50', 'vgg16'])
def test_bench_task2_line(arch):
    """`bench.py --task 2` end to end at a tiny batch: the task-1 leg, the switch to task 2 (30 % of every layer free, a new head with ITS OWN
    label range -- ResNet-50's second task has 196 classes where the first has 200 --, piggymasks on every masked layer), the cycle
    with SGD + Adam, the finetune_again leg; the JSON line carries the ratio block."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--arch', arch, '--task', '2', '--steps', '4', '--warmup', '1', '--batch', '8',
                        '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith('{')][-1])
    assert out['config']['task'] == 2 and 'NOT the headline' in out['metric'] and out['n_gpus'] == 1
    t2 = out['task2']
    assert t2['task1_ms_per_step'] > 0 and t2['finetune_again_ms_per_step'] > 0 and abs(t2['free_share_handed_to_task2'] - 0.3) < 0.01
    assert out['config']['cycle']['prune_events'] >= 1 and 'roofline' in out and out.get('cpu_baseline') is None



# ---- a large SharableLinear weight used TWICE in one graph, exchanged as row blocks (_ChunkedGradient.active()'s join path)
def _shared_weight_grads(rank, world, data_parallel, dev='cuda:0'):
    import torch.nn as nn
    from cpg_amd import dist as cdist
    from cpg_amd.models import layers as nl

    class Twice(nn.Module):
        def __init__(self):
            super().__init__()
            self.fc = nl.SharableLinear(96, 96)
            self.head = nn.Linear(96, 5)

        def forward(self, x):
            return self.head(self.fc(torch.relu(self.fc(x))))            # the same masked weight in two places of the graph
    torch.manual_seed(3)
    net = Twice()
    nn.init.normal_(net.fc.weight, 0, 0.1)
    nn.init.zeros_(net.fc.bias)
    net = net.to(dev)
    model = cdist.DataParallel(net, large_numel=1 << 10, chunk_numel=1 << 10, nchunks=4) if data_parallel else net
    g = torch.Generator().manual_seed(9)
    x, t = torch.randn(16, 96, generator=g), torch.randint(0, 5, (16,), generator=g)
    out = {}
    for step in range(2):                                  # second step: nothing may be left pending from the first
        xs, ts = cdist.shard_batch(x, t, rank, world)
        net.zero_grad(set_to_none=True)
        torch.nn.functional.cross_entropy(model(xs.to(dev)), ts.to(dev)).backward()
        if data_parallel:
            model.finish_gradient_sync()
            assert not net.fc.weight._cpg_dp_chunk.pending
            out['chunks_step%d' % step] = [k for k, _ in model.last_bucket_log].count('chunk')
        out['g%d' % step] = {n: p.grad.detach().cpu().clone() for n, p in net.named_parameters()}
    torch.cuda.synchronize()
    return out


def _shared_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        torch.save(_shared_weight_grads(rank, world, True), os.path.join(out_dir, 'shared%d.pt' % rank))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_chunked_exchange_of_a_weight_used_twice_equals_single_process(tmp_path):
    """The first backward call of the shared weight sends its gradient as 4 row blocks; the second call arrives while they are on the
    wire: active() joins them and falls back to the whole-tensor path, autograd sums the two, the hook all-reduces mean(g1) + g2_local.
    Result: every rank holds the single-process full-batch gradient, on both steps, and no row block is left pending."""
    import torch.multiprocessing as mp
    mp.spawn(_shared_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(os.path.join(tmp_path, 'shared0.pt')), torch.load(os.path.join(tmp_path, 'shared1.pt'))
    ref = _shared_weight_grads(0, 1, False)
    for step in range(2):
        assert r0['chunks_step%d' % step] == 4, r0                     # the row-block path really ran
        for n, want in ref['g%d' % step].items():
            assert torch.equal(r0['g%d' % step][n], r1['g%d' % step][n]), n
            np.testing.assert_allclose(r0['g%d' % step][n].numpy(), want.numpy(), rtol=1e-4, atol=1e-6 * float(want.abs().max()), err_msg=n)


@pytest.mark.parametrize('arch', ['spherenet20', 'resnet50'])
def test_bench_task_sequence_line(arch):
    """`bench.py --task-sequence 2 --arch A` end to end at a tiny batch: configs[3] / configs[4]'s flow through CPGSession at FULL width --
    pass-through task 1 + prune run, task 2 with piggymasks (SphereNet-20: AngleLinear / AngleLoss -> nn.Linear / CE, embeddings as task 1's
    evaluation) -- and the line's own assertions: weights finite, every earlier task bit-identical after the later one."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--arch', arch, '--task-sequence', '2', '--steps', '22', '--batch', '4'],
                       env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith('{')][-1])
    assert out['valid'] is True and out['weights_finite'] and out['earlier_tasks_bit_identical'] == {'checked': 3, 'bit_identical': True}
    t1, t2 = out['tasks']
    assert t1['pass_through'] is True and t2['pass_through'] is False and 'NOT the headline' in out['metric']
    assert set(t2['owner_histogram']) >= {'1', '2'} and t2['shared_ratio'] is not None and t1['first_rank_prune_event']['k_total'] > 0
