"""RCCL tests that ARM THEMSELVES: skipped on a one-GPU box, run on any box that shows >= 2 GPUs.

The reference data-parallelises with nn.DataParallel over every visible GPU (CPG_cifar100_main_normal.py:112-114,199-200;
experiment2/CPG_imagenet.sh:46; experiment3/FvGeEm_CPG_face.sh:53).  Here that is one process per GPU over RCCL
(torch.distributed backend 'nccl'): N = the largest of 8 / 4 / 2 ranks the box has GPUs for, rank r on cuda:r.

  (i)  the two scenarios of tests/test_dist_gpu.py (task-1 prune with rank-prune events; task-2 piggymasks with the packed
       gradient exchange) over RCCL: every rank ends bit-identical to rank 0, and rank 0 equals ONE process on the full batch;
  (ii) bench.py --gpus N (weak scaling) and --global-batch 256 (the reference's own split): the JSON line is the LAST stdout line,
       multi_gpu.rccl_ranks == N, the replicas are identical after the cycle, the exchange starts with the four ~103 MB row chunks
       of features.45's gradient.

A one-GPU box runs none of this (tests/test_dist_gpu.py covers the composition with two ranks sharing the GPU over gloo, and
test_data_parallel_wrapper_over_rccl_world1 the RCCL calls at world size 1)."""
import json
import os
import subprocess
import sys

import pytest
import torch

from test_dist_gpu import _free_port, _two_ranks, check_task1_against_single_process, check_task2_against_single_process

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NGPU = torch.cuda.device_count() if torch.cuda.is_available() else 0
WORLD = max([n for n in (8, 4, 2) if n <= NGPU], default=0)          # (the 8-image test batches split evenly over 2 / 4 / 8 ranks)
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(NGPU < 2, reason='RCCL across ranks needs >= 2 GPUs (this box shows %d)' % NGPU)]


def test_rccl_ranks_task1_prune_match_single_process(tmp_path):
    check_task1_against_single_process(_two_ranks(tmp_path, 'task1_prune', world=WORLD, backend='nccl'))


def test_rccl_ranks_task2_packed_exchange_match_single_process(tmp_path):
    check_task2_against_single_process(_two_ranks(tmp_path, 'task2_finetune', world=WORLD, backend='nccl'))


def _bench(extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(WORLD), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', str(WORLD), '--steps', '8', '--warmup', '3',
           '--no-cpu-baseline'] + extra
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    out = json.loads(lines[-1])                             # the ONE JSON line is the last thing on stdout (RCCL's banner comes before it)
    return out


@pytest.mark.parametrize('split', ['weak', 'strong'])
def test_bench_over_rccl(split):
    out = _bench([] if split == 'weak' else ['--global-batch', '256'])
    assert out['n_gpus'] == WORLD and out['steps'] == 8 and out['scaling'] == split
    assert out['config']['global_batch'] == (256 * WORLD if split == 'weak' else 256)
    mg = out['multi_gpu']
    assert mg['backend'] == 'rccl' and mg['rccl_ranks'] == WORLD
    assert mg['replicas_identical_after_cycle'] is True
    assert len(mg['per_rank_ms_per_step']) == WORLD
    chunks = mg['buckets'][:4]
    assert [b['kind'] for b in chunks] == ['chunk'] * 4 and all(abs(b['bytes'] - 4096 * 25088 * 4 / 4) < 1e6 for b in chunks), chunks
    assert out['value'] > 0 and out['ms_per_step'] > 0
    assert abs(out['value'] - out['config']['global_batch'] * 1000.0 / out['ms_per_step']) <= 1e-3 * out['value']
