"""Data-parallel path on CPU: world_size-2 gloo processes through cpg_amd.dist.DataParallel.

The wrapper is device-agnostic (it only touches .grad tensors and torch.distributed), so its
collective logic -- per-parameter async all-reduce of large tensors, coalesced small tensors, mean
over ranks, parameter / buffer broadcast, batch sharding -- is exercised here with the gloo backend on
a CPU model (the oracle's narrow VGG16-BN in eval mode, so BatchNorm does not couple the shards).
Expected: gradients after finish_gradient_sync() == gradients of ONE process on the full batch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    from cpg_amd import dist as cdist
    from oracle import net as onet

    torch.manual_seed(100 + rank)                      # ranks start from DIFFERENT weights on purpose
    net = onet.OracleVGG(0.125, 'cifar100')
    net.add_dataset('t1', 5)
    net.set_dataset('t1')
    for b in net.buffers():
        if b.dtype.is_floating_point:
            b.add_(float(rank))                        # and different BN statistics
    model = cdist.DataParallel(net, large_numel=1 << 12)       # small threshold: both code paths are taken
    model.sync_buffers()
    model.eval()
    g = torch.Generator().manual_seed(7)
    x = torch.randn(8, 3, 32, 32, generator=g)
    t = torch.randint(0, 5, (8,), generator=g)
    xs, ts = cdist.shard_batch(x, t)
    assert xs.shape[0] == 4
    loss = F.cross_entropy(model(xs), ts)
    loss.backward()
    model.finish_gradient_sync()
    grads = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    params = {n: p.detach().clone() for n, p in net.named_parameters()}
    bufs = {n: b.clone() for n, b in net.named_buffers()}
    torch.save({'grads': grads, 'params': params, 'bufs': bufs, 'x': x, 't': t}, os.path.join(out_dir, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_gradients_match_single_process(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, 'rank0.pt'))
    r1 = torch.load(os.path.join(tmp_path, 'rank1.pt'))
    # parameters and buffers were broadcast from rank 0
    for n in r0['params']:
        assert torch.equal(r0['params'][n], r1['params'][n]), n
    for n in r0['bufs']:
        assert torch.equal(r0['bufs'][n], r1['bufs'][n]), n
    # both ranks hold identical, fully reduced gradients
    for n in r0['grads']:
        assert torch.equal(r0['grads'][n], r1['grads'][n]), n
    # ... equal to the single-process full-batch gradient (loss = mean over the global batch)
    sys.path.insert(0, ROOT)
    from oracle import net as onet
    net = onet.OracleVGG(0.125, 'cifar100')
    net.add_dataset('t1', 5)
    net.set_dataset('t1')
    net.load_state_dict({**r0['params'], **r0['bufs']}, strict=False)
    net.eval()
    loss = F.cross_entropy(net(r0['x']), r0['t'])
    loss.backward()
    for n, p in net.named_parameters():
        if p.grad is None:
            continue
        scale = float(p.grad.abs().max()) + 1e-12
        np.testing.assert_allclose(r0['grads'][n].numpy(), p.grad.numpy(), rtol=1e-4, atol=1e-6 * scale, err_msg=n)


def _worker_piggymasks(rank, world, port, out_dir):
    """The driver's phase pattern: piggymask Parameters are re-created (twice) after wrapping; refresh_hooks() must hook
    every new one even though CPython hands the freed Parameters' ids to the new ones."""
    import gc
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    from cpg_amd import dist as cdist
    from oracle import net as onet
    torch.manual_seed(5)
    net = onet.OracleVGG(0.0625, 'cifar100')
    net.add_dataset('t1', 5)
    net.set_dataset('t1')
    model = cdist.DataParallel(net, large_numel=1 << 10)
    model.eval()
    g = torch.Generator().manual_seed(11)
    out = {}
    for rnd in range(3):
        for _, m in net.masked_layers():
            m.piggymask = torch.nn.Parameter(torch.full(tuple(m.weight.shape), 0.01))       # frees the previous one
        gc.collect()
        model.refresh_hooks()
        x = torch.randn(8, 3, 32, 32, generator=g)
        t = torch.randint(0, 5, (8,), generator=g)
        xs, ts = cdist.shard_batch(x, t)
        net.zero_grad()
        F.cross_entropy(model(xs), ts).backward()
        model.finish_gradient_sync()
        out[rnd] = {n: m.piggymask.grad.clone() for n, m in net.masked_layers()}
    torch.save(out, os.path.join(out_dir, 'pm_rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_recreated_piggymasks_stay_hooked(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker_piggymasks, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, 'pm_rank0.pt'))
    r1 = torch.load(os.path.join(tmp_path, 'pm_rank1.pt'))
    for rnd in r0:
        assert len(r0[rnd]) == 15
        for n in r0[rnd]:
            assert float(r0[rnd][n].abs().max()) > 0, (rnd, n)
            assert torch.equal(r0[rnd][n], r1[rnd][n]), 'round %d: piggymask gradient of %s was not all-reduced' % (rnd, n)


def _worker_metrics(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from cpg_amd.utils import Metric
    m = Metric('train_accuracy')
    # rank 0 sees an easy shard, rank 1 a hard one: the local means straddle the driver's 0.95 bar
    for acc, num in ((0.99, 4), (0.97, 4)) if rank == 0 else ((0.90, 4), (0.92, 2)):
        m.update(torch.tensor(acc), num)
    local = float(m.avg)
    m.all_reduce_()
    torch.save({'local': local, 'global': float(m.avg), 'n': float(m.n)}, os.path.join(out_dir, 'metric_rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_metrics_are_global_and_rank_identical(tmp_path):
    """CPGSession.run_task decides grow / stop-the-sweep on the train accuracy Manager.train returns; under data
    parallelism that must be the accuracy over the GLOBAL batches (nn.DataParallel's gathered output, utils/manager.py:60),
    bit-identical on every rank, or the ranks take different branches and their collectives mismatch."""
    world, port = 2, _free_port()
    mp.spawn(_worker_metrics, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, 'metric_rank0.pt'))
    r1 = torch.load(os.path.join(tmp_path, 'metric_rank1.pt'))
    assert r0['local'] > 0.95 > r1['local']                       # local decisions WOULD differ
    assert r0['global'] == r1['global'] and r0['n'] == r1['n'] == 14.0
    np.testing.assert_allclose(r0['global'], (0.99 * 4 + 0.97 * 4 + 0.90 * 4 + 0.92 * 2) / 14, rtol=1e-6)


def test_seed_per_rank_differs():
    sys.path.insert(0, ROOT)
    from cpg_amd import dist as cdist
    assert cdist.seed_per_rank(1) == 1                       # no process group: rank 0
    a = torch.rand(3)
    cdist.seed_per_rank(1)
    assert torch.equal(a, torch.rand(3))


def test_shard_batch_and_inactive_wrapper():
    sys.path.insert(0, ROOT)
    from cpg_amd import dist as cdist
    x, t = torch.arange(12.).view(6, 2), torch.arange(6)
    xs, ts = cdist.shard_batch(x, t, rank=1, world=3)
    assert xs.tolist() == [[4., 5.], [6., 7.]] and ts.tolist() == [2, 3]
    with pytest.raises(ValueError):
        cdist.shard_batch(x, t, rank=0, world=4)
    m = cdist.DataParallel(torch.nn.Linear(2, 2))          # no process group: transparent wrapper
    assert [n for n, _ in m.named_modules()][1] == 'module'
    m(x).sum().backward()
    m.finish_gradient_sync()
    m.sync_buffers()
