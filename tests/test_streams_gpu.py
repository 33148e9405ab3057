"""What include/cpg_hip.h promises about streams and threads (SURVEY.md section 8b "Ownership" / "Threading"): every entry point
enqueues on the stream it is GIVEN -- the Python mirror hands it torch's current stream -- and the library may be called concurrently
from several host threads on different streams.

  (i)  whole training cycles (conv forward / input gradient / weight gradient incl. the Winograd and tail pieces, the stems, linear layers,
       BatchNorm / PReLU passes, gradient routing fused into SGD and Adam, rank-prune events, histograms, apply_mask) run under
       `with torch.cuda.stream(side)` give results BIT-equal to the default-stream run -- first with the default stream BLOCKED by a
       spinning kernel for the whole duration of a family's launches: work that leaked onto stream 0 (a launch, a memset, a copy) could
       not have finished when the side stream reports done, and the results would differ;
  (ii) two Python threads, two streams, two different networks, their train steps interleaved, 25 + 25 steps each -- bit-equal to the
       serial run (the planners read the process-wide option table and write `thread_local` error text: that is what this guards).
"""
import copy
import threading

import pytest
import torch
import torch.nn as nn

import _sequence as sq

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']


def _make(arch, width, ncls=5):
    """(net, wrapped model, owner masks) at the reference's seeded initialisation; built on the host, serially (global RNG)."""
    import cpg_amd.models as models
    from cpg_amd.driver import _Plain, masked_layers
    torch.manual_seed(1)
    kw = dict(dataset_history=[], dataset2num_classes={}, network_width_multiplier=width, shared_layer_info={})
    net = models.custom_vgg_cifar100(CFG, **kw) if arch == 'vgg' else getattr(models, arch)(**kw)
    dataset = 'face_verification' if arch == 'spherenet20' else 't1'
    net.add_dataset(dataset, ncls)
    net.set_dataset(dataset)
    sq.apply_pretrained(net, arch)
    net.to(DEV)
    model = _Plain(net)
    masks = {n: torch.ones(m.weight.shape, dtype=torch.uint8, device=DEV) for n, m in masked_layers(model)}
    return net, model, masks, dataset


def _batches(shape, ncls, n, seed):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn(*shape, generator=g).to(DEV), torch.randint(0, ncls, (shape[0],), generator=g).to(DEV)) for _ in range(n)]


def _cycle(model, masks, dataset, train, val, lr=1e-3):
    """One prune-mode epoch (an event every 2nd step) + validate on whatever stream is current; returns every tensor it produced."""
    from cpg_amd.driver import default_args
    from cpg_amd.utils import Optimizers
    from cpg_amd.utils.fused_sgd import MaskedSGD
    from cpg_amd.utils.manager import Manager
    args = default_args(mode='prune', dataset=dataset, pruning_frequency=2, target_sparsity=0.3, initial_sparsity=0.0, lr=lr)
    mgr = Manager(args, model, {}, masks, train, val, 0, len(train))
    params = [p for n, p in model.named_parameters()]
    opts = Optimizers()
    opts.add(MaskedSGD(params, pruner=mgr.pruner, lr=lr, momentum=0.9, nesterov=True), lr)
    outs = []
    h = model.register_forward_hook(lambda m, i, o: outs.append((o[0] if isinstance(o, tuple) else o).detach().clone()))
    mgr.train(opts, 0, [lr], 0)
    if dataset == 'face_verification':
        outs.extend(mgr.eval_embeddings(0))
    else:
        mgr.validate(0)
    h.remove()
    res = {'out%d' % i: o for i, o in enumerate(outs)}
    res.update({'w/' + k: v.detach().clone() for k, v in model.state_dict().items()})
    res.update({'m/' + k: v.clone() for k, v in masks.items()})
    res['sparsity'] = torch.tensor(mgr.pruner.calculate_sparsity())
    return res


def _same(a, b, what):
    assert set(a) == set(b)
    for k in a:
        assert torch.equal(a[k].cpu(), b[k].cpu()), '%s: %s differs' % (what, k)


CASES = {'vgg': (0.25, (8, 3, 32, 32)), 'resnet50': (0.25, (8, 3, 64, 64)), 'spherenet20': (0.25, (4, 3, 112, 112))}


def _fresh(arch):
    width, shape = CASES[arch]
    net, model, masks, dataset = _make(arch, width)
    return model, masks, dataset, _batches(shape, 5, 6, 3), _batches(shape, 5, 1, 4)


def _spin(seconds):
    """Occupy the CURRENT stream for about `seconds` with torch's spinning kernel (calibrated: its unit differs between builds)."""
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    torch.cuda._sleep(20_000_000)
    b.record()
    b.synchronize()
    per = a.elapsed_time(b) / 1e3 / 20_000_000
    torch.cuda._sleep(int(seconds / per))


@pytest.mark.parametrize('arch', ['vgg', 'resnet50', 'spherenet20'])
def test_cycle_on_a_side_stream_is_bit_equal_and_never_touches_stream_0(arch):
    model, masks, dataset, train, val = _fresh(arch)
    ref = _cycle(model, masks, dataset, train, val)
    again = _cycle(*_fresh(arch))
    _same(ref, again, '%s: two default-stream runs (the kernels are deterministic)' % arch)
    # ---- the same cycle on a side stream while stream 0 spins
    model, masks, dataset, train, val = _fresh(arch)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    _spin(12.0)                                            # stream 0 is busy for the next ~12 s
    busy = torch.cuda.Event()
    busy.record()                                          # ... this event completes when the spin does
    with torch.cuda.stream(side):
        got = _cycle(model, masks, dataset, train, val)
        side.synchronize()
        got = {k: v.cpu() for k, v in got.items()}         # (copies on the side stream too)
    still_spinning = not busy.query()
    torch.cuda.synchronize()
    assert still_spinning, 'the side-stream cycle waited for stream 0 (or outlasted the 12 s spin: lengthen it)'
    _same(ref, got, '%s on a side stream' % arch)


def test_two_tasks_with_piggymasks_on_a_side_stream():
    """Tasks >= 2 (binarizer in the pack passes, piggymask-gradient epilogues, fused Adam, shared_ratio, the evaluate path) through
    CPGSession.run_task on a side stream == on the default stream."""
    from cpg_amd.driver import CPGSession, default_args

    def run():
        sess = CPGSession('custom_vgg_cifar100', 0.25, device=DEV, seed=1)
        args = default_args(lr=1e-2, lr_mask=2e-3, pruning_frequency=1, pruning_interval=1, prune_lr=1e-3)
        res = {}
        for t in (1, 2):
            tr, va = _batches((8, 3, 32, 32), 5, 4, 10 + t), _batches((8, 3, 32, 32), 5, 2, 20 + t)
            sess.run_task('t%d' % t, 5, tr, va, accuracy_goal=0.0, finetune_epochs=1, prune_epochs=1, sparsities=(0.3,), args=args,
                          min_train_acc=-1.0, retrain_epochs=1)
            for k, o in enumerate(sess.evaluate('t%d' % t, va)[1]):
                res['logits%d_%d' % (t, k)] = o.clone()
        res.update({'w/' + k: v.detach().clone() for k, v in sess.net.state_dict().items()})
        res.update({'m/' + k: v.clone() for k, v in sess.masks.items()})
        return res
    ref = run()
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        got = run()
        side.synchronize()
    _same(ref, got, 'two-task session on a side stream')


def test_two_threads_two_streams_interleaved_steps_equal_the_serial_runs():
    serial = {}
    for arch in ('vgg', 'resnet50'):
        serial[arch] = [_cycle(*_fresh(arch)) for _ in range(1)][0]
    # 25 + 25 steps per thread: the 6-step epoch above 4 times over on fresh optimizers would change the numbers; instead each thread
    # repeats the WHOLE cycle from a fresh state N times and every repetition must equal the serial result
    states = {arch: [_fresh(arch) for _ in range(4)] for arch in ('vgg', 'resnet50')}      # built serially (global RNG)
    torch.cuda.synchronize()
    results, errors = {'vgg': [], 'resnet50': []}, []
    gate = threading.Barrier(2)

    def worker(arch):
        try:
            stream = torch.cuda.Stream()
            gate.wait()
            with torch.cuda.stream(stream):
                for st in states[arch]:
                    results[arch].append({k: v.cpu() for k, v in _cycle(*st).items()})
                stream.synchronize()
        except Exception as e:                             # noqa: BLE001
            errors.append((arch, repr(e)))
    threads = [threading.Thread(target=worker, args=(a,)) for a in ('vgg', 'resnet50')]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for arch in ('vgg', 'resnet50'):
        assert len(results[arch]) == 4
        for i, got in enumerate(results[arch]):
            _same(serial[arch], got, '%s, repetition %d beside the other thread' % (arch, i))
