#!/usr/bin/env python3
"""bench.py -- images/sec of one CPG train -> gradual-prune -> retrain cycle, VGG16-BN task 1.

Contract (see the task statement): `python bench.py --gpus N --steps K --warmup W`.  For N > 1 it runs one rank per GPU
over RCCL: either it is launched by `torch.distributed.run` (RANK / WORLD_SIZE in the environment), or -- started as a
plain `python bench.py --gpus N` -- it re-launches ITSELF through `torch.distributed.run --nproc-per-node N`; in both
cases it refuses to run when the world size is not N.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1]): `custom_vgg` (VGG16-BN, 224x224, Dropout), 5-way head,
fp32, batch 256 PER GPU (weak scaling; configs[2] is 2048 = 8 x 256), synthetic N(0,1) images / randint(0,5) labels
already resident in HBM, weights from the reference's init at seed 1 (Dropout streams re-seeded per rank), wd 4e-5.

The K timed steps are the section-8d cycle with a shape that does NOT depend on K:

    validate (apply_mask + 2 eval batches of 100)   after every 20th train step (20-step epochs: 1 validate per 20 steps)
    phase A "finetune"                              the first A = max(1, round(K / 11)) steps, SGD-nesterov lr 1e-2 (SphereNet-20: 1e-3)
    phase B "prune 0.0 -> 0.1" + recovery           the other K - A steps, lr 1e-3 (SphereNet-20: 5e-4); pruning window = first 4 f steps of the
                                                    phase, rank-prune event every f = max(1, A // 2) steps -> 4 events,
                                                    then fixed-mask recovery
    mask statistics                                 every train step and every validate batch (utils/manager.py:77-88,126-136)

At the default K = 220 this is exactly section 8d: 20-step epochs, 1 finetune + 10 prune-run epochs, events at steps
10/20/30/40 of the prune run, 11 validates.  At the driver's K = 20 it is the same cycle compressed 11 x: 2 finetune
steps, 18 prune-run steps with events at steps 1-4, 1 validate -- the same validate : train ratio and the same number of
prune events, so the images/sec figure is the same quantity.

A "step" is one minibatch through the hot path (zero_grad, forward, loss, backward, gradient routing, SGD step, prune
event when due, statistics); validates are inside the timed region.  value = global_batch * K / wall, wall = max over
ranks of the barrier-bracketed timed region.
"""
import argparse
import json
import os
import sys
import time
import types

import torch
import torch.distributed as dist
import torch.nn as nn

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import cpg_amd.models as models                     # noqa: E402
from cpg_amd import dist as cdist                    # noqa: E402
from cpg_amd.models import layers as nl              # noqa: E402
from cpg_amd.utils import Optimizers, settle_host_gc   # noqa: E402
from cpg_amd.utils.fused_sgd import MaskedAdam, MaskedSGD   # noqa: E402
from cpg_amd.utils.manager import Manager            # noqa: E402
from cpg_amd.utils.prune import SparsePruner         # noqa: E402

VGG_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']
PEAK_FP32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 256 FLOP/clk x 2.4 GHz
PEAK_BF16_MFMA_TFLOPS = 2500.0       # same guide: dense bf16 (v_mfma_f32_32x32x16_bf16), only used by the opt-in --math bf16 run
# Per topology: builder, input size, dataset name (selects the loss in Manager: 'face_verification' -> AngleLoss), classes of the
# synthetic task, and the ALGORITHMIC flops of one train step per image = 2 x (3 x MACs of the masked layers - MACs of the first
# layer, which has no input gradient) -- SURVEY.md section 8 / BASELINE.md section 3: 15.466 / 4.087 / 2.029 G MACs per image.
ARCHS = {
    'vgg16': dict(size=224, dataset='task1', classes=5, dataset2='task2', classes2=5, flop_train=92.62e9, flop_fwd=30.932e9,
                  workload='configs[1]: VGG16-BN custom_vgg 224x224'),
    'resnet50': dict(size=224, dataset='cubs_cropped', classes=200, dataset2='stanford_cars_cropped', classes2=196, flop_train=2 * (3 * 4.087e9 - 0.118e9), flop_fwd=2 * 4.087e9,
                     workload='configs[3] topology on one GPU: ResNet-50 (Bottleneck, masked 7x7 s2 / 1x1 / 3x3 s1 / 3x3 s2 convs) 224x224'),
    # lrs: the reference's own learning rates for this configuration (experiment3/FvGeEm_CPG_face.sh:21-25,130: finetune 1e-3, prune run
    # 5e-4).  The VGG16 cycle's 1e-2 / 1e-3 makes this BatchNorm-free network diverge on random labels within 5 steps -- on torch's own ops
    # exactly as on the HIP kernels (tools/attic/diag_sph_nan.py) -- and every K >= 60 cycle then ran on NaN weights (`cycle_check` said so).
    'spherenet20': dict(size=112, dataset='face_verification', classes=4630, dataset2='gender', classes2=2, flop_train=2 * (3 * 2.029e9 - 0.0054e9), flop_fwd=2 * 2.029e9,
                        workload='configs[4] topology on one GPU: SphereNet-20 112x112, AngleLinear head + AngleLoss', lrs=(1e-3, 5e-4)),
}
LRS = (1e-2, 1e-3)      # (finetune, prune run) of the timed cycle: experiment1's; main() takes ARCHS[arch]['lrs'] when the topology has its own
FLOP_PER_IMG_TRAIN = ARCHS['vgg16']['flop_train']


class KernelClock:
    """HIP-event timing of every masked-layer kernel launch inside the timed region (events are
    recorded on torch's current stream, the stream the C ABI launches on)."""

    def __init__(self):
        self.records = []           # (kind, algorithmic flops, start_event, end_event, executed flops, algorithmic bytes)
        self.enabled = False
        # `--clock-every N`: only every N-th train step of the cycle (and every validate) carries events -- an event pair costs the
        # queue ~ 7 us per launch (same-box A/B: all launches clocked 2345 vs none 2360 images/s on VGG16, 71.2 vs 70.4 ms on
        # ResNet-50's 160 launches per step).  Launches of the other steps are only counted (`unclocked`), so the whole-region
        # figures still cover every launch.
        self.timing = True
        self.step = None                    # index of the train step the launches belong to (None: a validate)
        self.unclocked = [0, 0.0, 0.0]      # launches, algorithmic flops, executed flops of the enabled-but-untimed launches

    def wrap(self, lib):
        clock = self

        def timed(name, kind, flops_fn, wino=None, bytes_fn=None):
            raw = getattr(lib, name)
            bytes_fn = bytes_fn or (conv_bytes if flops_fn is conv_flops else None)

            def call(*args):
                if not clock.enabled:
                    return raw(*args)
                if not clock.timing:
                    fl = flops_fn(args)
                    u = clock.unclocked
                    u[0] += 1
                    u[1] += fl
                    u[2] += fl / 2.25 if wino is not None and wino(args) else fl
                    return raw(*args)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                rc = raw(*args)
                e.record()
                fl = flops_fn(args)
                clock.records.append((kind(args), fl, s, e, fl / 2.25 if wino is not None and wino(args) else fl,
                                      bytes_fn(args) if bytes_fn else 0.0, clock.step))
                return rc
            return call

        def conv_flops(args):
            d = args[0]._obj
            oh = (d.H + 2 * d.pad_h - d.dil_h * (d.R - 1) - 1) // d.stride_h + 1
            ow = (d.W + 2 * d.pad_w - d.dil_w * (d.S - 1) - 1) // d.stride_w + 1
            return 2.0 * d.N * d.K * oh * ow * d.C * d.R * d.S

        def conv_bytes(args):           # each pass reads two of {x, y or gy, W} once and writes the third once
            d = args[0]._obj
            oh = (d.H + 2 * d.pad_h - d.dil_h * (d.R - 1) - 1) // d.stride_h + 1
            ow = (d.W + 2 * d.pad_w - d.dil_w * (d.S - 1) - 1) // d.stride_w + 1
            return 4.0 * (d.N * d.C * d.H * d.W + d.N * d.K * oh * ow + d.K * d.C * d.R * d.S)

        # Launches that run Winograd F(2x2, 3x3) (cpg_conv2d_winograd: the library's own dispatch rule) execute 16 / 36 of the
        # algorithmic multiply-adds: reported beside the algorithmic rate, never instead of it
        def wino_fwd(args):
            return bool(lib.cpg_conv2d_winograd(args[0], 0))

        def wino_dgrad(args):
            return bool(lib.cpg_conv2d_winograd(args[0], 1))

        def wino_wgrad(args):
            return bool(lib.cpg_conv2d_winograd(args[0], 2))

        def wino_eval(args):            # cpg_conv2d_fwd_bn_eval runs k_wg1 / k_wg3 <.., BNE> where the library says so (pass 3)
            return bool(lib.cpg_conv2d_winograd(args[0], 3))

        def conv_kind(prefix):
            def k(args):
                d = args[0]._obj
                return '%s %dx%d c%d->%d @%d' % (prefix, d.R, d.S, d.C, d.K, d.H)
            return k

        def lin_flops(bi):
            return lambda a: 2.0 * a[bi] * a[bi + 1] * a[bi + 2]

        def lin_bytes(bi):
            return lambda a: 4.0 * (a[bi] * a[bi + 1] + a[bi + 1] * a[bi + 2] + a[bi] * a[bi + 2])

        class Proxy(object):
            pass
        p = Proxy()
        for n in dir(lib):
            if n.startswith('cpg_'):
                setattr(p, n, getattr(lib, n))
        p.cpg_conv2d_fwd = timed('cpg_conv2d_fwd', conv_kind('conv_fwd'), conv_flops, wino_fwd)
        # same contraction as cpg_conv2d_fwd; its epilogue also emits the BatchNorm partial sums
        p.cpg_conv2d_fwd_bnstats = timed('cpg_conv2d_fwd_bnstats', conv_kind('conv_fwd'), conv_flops, wino_fwd)
        # the fused stem (conv -> BatchNorm2d -> ReLU, conv output never written): the layer's algorithmic flops are counted on its
        # BatchNorm/ReLU pass; the statistics pass before it recomputes the same conv (time counted, no extra algorithmic flops);
        # its backward passes (cpg_stem_bn_relu_bwd_*) are BatchNorm-backward + weight-gradient work and are not clocked
        # (a family of its own: these launches also do the BatchNorm's statistics / apply work and are HBM-bound -- in conv_fwd they
        # would charge BatchNorm time to the Winograd kernels' roofline)
        p.cpg_stem_bn_stats = timed('cpg_stem_bn_stats', conv_kind('stem_bn_fwd'), lambda a: 0.0, None, lambda a: 0.0)
        p.cpg_stem_bn_relu_fwd = timed('cpg_stem_bn_relu_fwd', conv_kind('stem_bn_fwd'), conv_flops)
        # ... and the inference variant with the eval-mode BatchNorm + ReLU folded into the epilogue (validate)
        p.cpg_conv2d_fwd_bn_eval = timed('cpg_conv2d_fwd_bn_eval', conv_kind('conv_fwd'), conv_flops, wino_eval)
        # the opt-in bf16 MFMA kernels (--math bf16) get their own families: they are measured against the bf16 peak
        p.cpg_conv2d_fwd_bf16 = timed('cpg_conv2d_fwd_bf16', conv_kind('conv_fwd_bf16'), conv_flops)
        p.cpg_conv2d_dgrad_bf16 = timed('cpg_conv2d_dgrad_bf16', conv_kind('conv_dgrad_bf16'), conv_flops)
        p.cpg_conv2d_fwd_bf16x3 = timed('cpg_conv2d_fwd_bf16x3', conv_kind('conv_fwd_bf16'), conv_flops)
        p.cpg_conv2d_dgrad_bf16x3 = timed('cpg_conv2d_dgrad_bf16x3', conv_kind('conv_dgrad_bf16'), conv_flops)
        p.cpg_conv2d_wgrad_bf16x3 = timed('cpg_conv2d_wgrad_bf16x3', conv_kind('conv_wgrad_bf16'), conv_flops)
        p.cpg_conv2d_wgrad_bf16 = timed('cpg_conv2d_wgrad_bf16', conv_kind('conv_wgrad_bf16'), conv_flops)
        # (same contraction; its epilogue also does the BatchNorm-backward reduction of the layer below)
        p.cpg_conv2d_dgrad_bnbwd = timed('cpg_conv2d_dgrad_bnbwd', conv_kind('conv_dgrad'), conv_flops)
        p.cpg_conv2d_dgrad = timed('cpg_conv2d_dgrad', conv_kind('conv_dgrad'), conv_flops, wino_dgrad)
        # ... and with the other consumer's gradient added in the epilogue (ResNet's bottlenecks, SphereNet's residual units): the same
        # contraction + one more tensor of the input's size read.  (Until round 5 these launches were neither clocked nor counted: ResNet-50's
        # input-gradient family and whole_step left out 13 launches per step.)
        p.cpg_conv2d_dgrad_add = timed('cpg_conv2d_dgrad_add', conv_kind('conv_dgrad'), conv_flops, wino_dgrad,
                                       lambda a: conv_bytes(a) + 4.0 * a[0]._obj.N * a[0]._obj.C * a[0]._obj.H * a[0]._obj.W)
        p.cpg_conv2d_wgrad = timed('cpg_conv2d_wgrad', conv_kind('conv_wgrad'), conv_flops, wino_wgrad)
        p.cpg_linear_fwd = timed('cpg_linear_fwd', lambda a: 'linear_fwd', lin_flops(6), None, lin_bytes(6))
        p.cpg_linear_dgrad = timed('cpg_linear_dgrad', lambda a: 'linear_dgrad', lin_flops(5), None, lin_bytes(5))
        p.cpg_linear_wgrad = timed('cpg_linear_wgrad', lambda a: 'linear_wgrad', lin_flops(8), None, lin_bytes(8))
        return p

    def summary(self, train_steps=None):
        """{kind: [launches, ms, algorithmic flops, executed flops, algorithmic bytes]} over the timed region.  With `--clock-every N`
        only some train steps carry events while every validate does: each train-step record stands for train_steps / clocked steps
        launches (a stratified sample -- without the weights the validate launches, batch 100 with the BatchNorm epilogue, would
        count N times too often in a family's average)."""
        agg = {}
        steps = set(r[6] for r in self.records if r[6] is not None)
        self.train_steps_clocked, self.train_ms, self.launches_clocked = len(steps), 0.0, len(self.records)
        w_train = float(train_steps) / len(steps) if train_steps and steps else 1.0
        for kind, flops, s, e, executed, nbytes, step in self.records:
            ms = s.elapsed_time(e)
            w = 1.0
            if step is not None:
                self.train_ms += ms
                w = w_train
            a = agg.setdefault(kind, [0, 0.0, 0.0, 0.0, 0.0])
            a[0] += w
            a[1] += w * ms
            a[2] += w * flops
            a[3] += w * executed
            a[4] += w * nbytes
        for a in agg.values():
            a[0] = int(round(a[0]))
        return agg


def _flush_c_stdio():
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:       # (no libc handle: nothing buffered that we could reach)
        pass


def csrc_digest():
    """sha256 over the kernel sources (cpg_amd/csrc, sorted by name): committed counter files carry the digest of the sources they
    were measured on, so that a kernel change makes `roofline.traffic` say it is stale instead of silently quoting old bytes."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'cpg_amd', 'csrc')
    for name in sorted(os.listdir(d)):
        if name.endswith(('.hip', '.h', '.cpp')):
            h.update(name.encode())
            with open(os.path.join(d, name), 'rb') as f:
                h.update(f.read())
    return h.hexdigest()


TRAFFIC_FILES = ('r06_traffic_bench.json', 'r05_traffic_bench.json', 'r04_traffic_bench.json', 'r03_traffic_bench.json')


def pmc_traffic(arch, family, batch):
    """HBM bytes per launch of a kernel family from committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in separate
    passes, gfx950 correction 2 x FETCH_SIZE + WRITE_SIZE).  First choice: profiles/r03_traffic_bench.json -- the passes ran over
    THIS command (bench.py --arch A --steps 20), so the launch mix is the bench's own (tools/bench_traffic.py); otherwise, for
    VGG16, the round-2 passes over one launch of each conv of a pass (profiles/r02_traffic.json).  None when neither applies."""
    if batch != 256:                 # every committed pass ran at the bench's default batch
        return None
    for fname in TRAFFIC_FILES:
        try:
            with open(os.path.join(ROOT, 'profiles', fname)) as f:
                doc = json.load(f)
            fam = doc['archs'][arch]['families'][family]
            stamp = doc.get('csrc_sha256')
            return {'hbm_bytes_per_launch': round(fam['hbm_bytes_per_launch_corrected']),
                    'fetch_size_bytes_per_launch': round(fam['fetch_size_bytes_per_launch']),
                    'write_size_bytes_per_launch': round(fam['write_size_bytes_per_launch']),
                    'launches_counted': fam['launches'], 'source': 'profiles/' + fname,
                    'measured_at_commit': doc.get('commit'), 'measured_on_csrc_sha256': stamp,
                    # the kernels changed since the counters were collected (or the file carries no digest): the bytes are history
                    'stale': stamp != csrc_digest()}
        except (OSError, KeyError, ValueError):
            continue
    if arch != 'vgg16':
        return None
    for name in ('r02_traffic.json', 'r01_traffic.json'):
        try:
            with open(os.path.join(ROOT, 'profiles', name)) as f:
                fam = json.load(f)['families'].get(family)
            if fam is None:
                return None
            return {'hbm_bytes_per_launch': round(fam['hbm_bytes_per_launch_corrected']),
                    'algorithmic_bytes_per_launch_of_that_pass': round(fam['algorithmic_bytes_per_launch']), 'source': 'profiles/' + name}
        except (OSError, KeyError, ValueError):
            continue
    return None


DATASET = 'task1'                    # set by main() from --arch
WIDTH = 1.0                          # the ROOTED width multiplier the models and SparsePruner's statistics see; set by main() from --width-multiplier


def build_model(device, arch='vgg16'):
    torch.manual_seed(1)                       # reference default seed (CPG_cifar100_main_normal.py:79,135)
    kw = dict(dataset_history=[], dataset2num_classes={}, network_width_multiplier=WIDTH, shared_layer_info={})
    if arch == 'vgg16':
        net = models.custom_vgg(VGG_CFG, **kw)             # CPG_imagenet_main.py:193-196
    elif arch == 'resnet50':
        net = models.resnet50(**kw)                        # CPG_imagenet_main.py:189-192
    else:
        net = models.spherenet20(**kw)                     # CPG_face_main.py:173-178
    net.add_dataset(ARCHS[arch]['dataset'], ARCHS[arch]['classes'])
    net.set_dataset(ARCHS[arch]['dataset'])
    return net.to(device)


def make_args(mode, freq, width=None, finetune_again=False):
    width = WIDTH if width is None else width
    return types.SimpleNamespace(mode=mode, dataset=DATASET, finetune_again=finetune_again, target_sparsity=0.1,
                                 initial_sparsity=0.0, pruning_frequency=freq, weight_decay=4e-5,
                                 network_width_multiplier=width, cuda=True, log_path=None, progress=False)


def validate(mgr, epoch):
    """Manager.validate (utils/manager.py:103-152: apply_mask + eval forward + statistics).  The face task of config 5 has no
    classification validate in the reference -- CPG_face_main.py:417 calls evalLFW (utils/manager.py:156-195): apply_mask, then
    eval-mode EMBEDDINGS of image pairs, scored on the host with sklearn (out of scope: real LFW pairs).  Its device part is
    reproduced here on the synthetic validation batches."""
    if DATASET != 'face_verification':
        return mgr.validate(epoch)
    mgr.pruner.apply_mask()
    mgr.model.eval()
    root = mgr.model.module if hasattr(mgr.model, 'module') else mgr.model
    with torch.no_grad():
        for data, _ in mgr.val_loader:
            root.forward_to_embeddings(data)
            mgr.last_stats = {'sparsity': mgr.pruner.calculate_sparsity(), 'zero ratio': mgr.pruner.calculate_zero_ratio()}


EPOCH_STEPS = 20                     # 20-step epochs, validate after every epoch


def cycle_plan(steps):
    """(A, f): finetune steps and rank-prune frequency of the K-step cycle (module docstring)."""
    A = min(steps, max(1, int(round(steps / 11.0))))
    f = max(1, A // 2)
    return A, f


def make_optimizers(model, pruner, lr, lr_mask=None):
    """SGD-nesterov over everything but the piggymasks and the other tasks' heads, the masked weights through the fused routing +
    step pass (CPG_cifar100_main_normal.py:320-341); from task 2 on also Adam(lr_mask) over the piggymasks (:342-346), through
    MaskedAdam's fused routing + step pass.  lr_mask None = task 1 (no piggymask exists)."""
    root = model.module if hasattr(model, 'module') else model
    idx = root.datasets.index(DATASET)
    sgd_params, adam_params = [], []
    for name, p in model.named_parameters():
        if 'classifiers' in name:
            if '.%d.' % idx in name:
                sgd_params.append(p)
        elif 'piggymask' in name:
            adam_params.append(p)
        else:
            sgd_params.append(p)
    o = Optimizers()
    o.add(MaskedSGD(sgd_params, pruner=pruner, lr=lr, momentum=0.9, nesterov=True), lr)
    if lr_mask is not None and adam_params:
        o.add(MaskedAdam(adam_params, pruner=pruner, lr=lr_mask), lr_mask)
    return o


def run_cycle(model, masks, pool, val_pool, steps, clock=None, marks=None, counts=None, task=1):
    """The K-step CPG cycle of one task.  Returns number of train steps executed.  `marks` collects (label, steps, event) at the
    phase boundaries (events only -- no synchronisation inside the timed region); `counts` receives the number of
    validates and rank-prune events that actually ran.  task >= 2: the same cycle with a piggymask on every masked layer --
    finetune with Adam(lr_mask 5e-4) on the piggymasks, the prune run with lr_mask 0 (experiment1/CPG_cifar100_scratch_mul_1.5.sh:
    39,57,116), `shared_ratio` in every validate batch (utils/manager.py:130-135, utils/prune.py:180-193)."""
    A, f = cycle_plan(steps)
    window = 4 * f
    done = 0
    n_val = 0

    def mark(label, n=0):
        if marks is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append((label, n, ev))
    mark('start')

    every = max(1, int(getattr(clock, 'every', 1))) if clock is not None else 1

    class SampledSteps(object):
        """A phase's batches (Manager.train iterates over it once); switches the kernel clock's events on for every `every`-th step
        of the cycle and back on behind the last one (validates are always clocked)."""

        def __init__(self, n, offset):
            self.batches, self.first = [pool[(offset + i) % len(pool)] for i in range(n)], offset

        def __len__(self):
            return len(self.batches)

        def __iter__(self):
            for i, b in enumerate(self.batches):
                clock.timing, clock.step = (self.first + i) % every == every // 2, self.first + i
                yield b
            clock.timing, clock.step = True, None

    def loader(n, offset):
        if clock is not None:
            return SampledSteps(n, offset)
        return [pool[(offset + i) % len(pool)] for i in range(n)]

    def sgd(lr, pruner, lr_mask):
        return make_optimizers(model, pruner, lr, lr_mask if task > 1 else None)

    def chunks(n_phase):
        """split a phase's steps at the global every-20th-step validate points: yields (n_steps, validate_after)"""
        left = n_phase
        pos = done
        while left > 0:
            n = min(left, EPOCH_STEPS - pos % EPOCH_STEPS)
            yield n, (pos + n) % EPOCH_STEPS == 0
            pos += n
            left -= n

    # phase A: finetune (free slots claimed by task 1)
    mgr = Manager(make_args('finetune', f), model, {}, masks, None, val_pool, 0, 0)
    mgr.pruner.make_finetuning_mask()
    opt = sgd(LRS[0], mgr.pruner, 5e-4)
    epoch = 0
    for n, val in list(chunks(A)):
        mgr.train_loader = loader(n, done)
        mgr.train(opt, epoch, [LRS[0]], 0)
        mark('finetune_train', n)
        done += n
        if val:
            validate(mgr, epoch)
            mark('validate', 1)
            n_val += 1
            epoch += 1
    # phase B: prune 0.0 -> 0.1 (4 rank-prune events inside the window), then recovery at the fixed mask
    events = 0
    if steps - done > 0:
        mgrB = Manager(make_args('prune', f), model, {}, masks, None, val_pool, 0, window)
        opt = sgd(LRS[1], mgrB.pruner, 0.0)
        step = 0
        for n, val in list(chunks(steps - done)):
            # keep window and recovery steps in separate marks
            parts = [n] if step >= window or step + n <= window else [window - step, n - (window - step)]
            for m in parts:
                mgrB.train_loader = loader(m, done)
                in_window = step < window
                _, step = mgrB.train(opt, epoch, [LRS[1]], step)
                mark('prune_window_train' if in_window else 'recovery_train', m)
                done += m
            if val:
                validate(mgrB, epoch)
                mark('validate', 1)
                n_val += 1
                epoch += 1
        events = mgrB.pruner.prune_events
    if counts is not None:
        counts.update(validates=n_val, prune_events=events, finetune_steps=A, prune_frequency=f, prune_window_steps=window)
    return done


def begin_task2(model, masks, arch, device):
    """Turn the resident network into the starting state of task 2 (what the reference reads back from task 1's final checkpoint,
    CPG_cifar100_main_normal.py:196-290): every slot owned by task 1 except the 30 % smallest-magnitude weights of each layer, which
    are free and zero (a finished gradual-prune sweep to 0.3: utils/prune.py:30-53 + make_pruned_zero), a new head, and a
    piggymask `full(0.01)` on every masked layer (:263-270).  Phase A's make_finetuning_mask then hands the free slots to task 2."""
    global DATASET
    root = model.module if hasattr(model, 'module') else model
    for m in masks.values():
        m.fill_(1)
    pr = SparsePruner(model, masks, make_args('prune', 1), 0, 1, 1)
    pr._rank_prune_layers(0.3)
    pr.make_pruned_zero()
    free = sum(int((m == 0).sum()) for m in masks.values()) / float(sum(m.numel() for m in masks.values()))
    root.add_dataset(arch['dataset2'], arch['classes2'])
    root.set_dataset(arch['dataset2'])
    root.classifiers.to(device)
    DATASET = arch['dataset2']
    fresh_piggymasks(model, masks)
    return free


def fresh_piggymasks(model, masks):
    root = model.module if hasattr(model, 'module') else model
    prefix = 'module.' if hasattr(model, 'module') else ''
    for name, m in root.named_modules():
        if isinstance(m, (nl.SharableConv2d, nl.SharableLinear)):
            m.piggymask = nn.Parameter(torch.full_like(masks[prefix + name], 0.01, dtype=torch.float32))
    if hasattr(model, 'refresh_hooks'):
        model.refresh_hooks()


def finetune_again_leg(model, masks, pool, val_pool, steps):
    """The piggymask retrain that ends every task >= 2 (`--mode finetune --finetune_again`, lr 1e-3, lr_mask 1e-4, fresh
    piggymasks: experiment1/CPG_cifar100_scratch_mul_1.5.sh:189-206, CPG_cifar100_main_normal.py:272-279,386-388): a validate
    first (:391-393), then `steps` train steps -- no slot changes owner, weights of the task and the picker over older tasks' learn."""
    fresh_piggymasks(model, masks)
    mgr = Manager(make_args('finetune', 1, finetune_again=True), model, {}, masks, [pool[i % len(pool)] for i in range(2)], val_pool, 0, 0)
    opt = make_optimizers(model, mgr.pruner, 1e-3, 1e-4)
    validate(mgr, -1)
    mgr.train(opt, 0, [1e-3, 1e-4], 0)                     # warm (Adam state allocation)
    mgr.train_loader = [pool[i % len(pool)] for i in range(steps)]
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    mgr.train(opt, 0, [1e-3, 1e-4], 0)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / steps


def phase_report(marks, model, masks, batch):
    """finetune-only / prune-window / recovery step times, validate time and the latency of one rank-prune
    event over all 15 masked layers (measured after the timed region, on a copy of the owner masks)."""
    acc = {}
    for (_, _, e0), (label, n, e1) in zip(marks[:-1], marks[1:]):
        a = acc.setdefault(label, [0.0, 0])
        a[0] += e0.elapsed_time(e1)
        a[1] += n
    rep = {}
    for label, (ms, n) in acc.items():
        if n:
            rep[label + ('_ms_per_call' if label == 'validate' else '_ms_per_step')] = round(ms / n, 3)
    for label in ('finetune_train', 'prune_window_train', 'recovery_train'):
        if label + '_ms_per_step' in rep:
            rep[label + '_images_per_sec'] = round(batch / rep[label + '_ms_per_step'] * 1e3, 1)
    pr = SparsePruner(model, {k: v.clone() for k, v in masks.items()}, make_args('prune', 1), 0, 100, None)
    pr._rank_prune_layers(0.05)                        # warm
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    pr._rank_prune_layers(0.10)
    e.record()
    torch.cuda.synchronize()
    rep['prune_event_ms'] = round(s.elapsed_time(e), 3)
    return rep


def optin_modes(model, masks, pool, steps, batch):
    """Train-step time of the same model in the two OPT-IN conv arithmetics (cpg_amd.models.layers.set_conv_math), measured
    after the timed cycle and reported beside it -- information for the reader, not the metric: 'bf16x3' (two-term bf16 split,
    3 MFMAs per product) holds north_star's 1e-4 logit bar (tests/test_hip_parity.py::test_first_forward_logits_golden_bf16x3),
    'bf16' does not (2e-2 of the output scale) -- both COMPUTED here by parity_check under the mode, not asserted."""
    res = {}
    for mode in ('bf16x3', 'bf16'):
        nl.set_conv_math(mode)
        try:
            mgr = Manager(make_args('finetune', 1), model, {}, {k: v.clone() for k, v in masks.items()},
                          [pool[i % len(pool)] for i in range(steps)], None, 0, 0)
            opt = Optimizers()
            opt.add(torch.optim.SGD(model.parameters(), lr=0.0, momentum=0.9, nesterov=True), 0.0)
            mgr.train_loader = [pool[0], pool[1]]
            mgr.train(opt, 0, [0.0], 0)                        # warm the kernels of this mode
            mgr.train_loader = [pool[i % len(pool)] for i in range(steps)]
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            s.record()
            mgr.train(opt, 0, [0.0], 0)
            e.record()
            torch.cuda.synchronize()
            ms = s.elapsed_time(e) / steps
            pc = parity_check(model.module if hasattr(model, 'module') else model, pool[0][0], WIDTH)      # (under THIS conv arithmetic)
            res[mode] = {'train_ms_per_step': round(ms, 3), 'train_images_per_sec': round(batch / ms * 1e3, 1),
                         'meets_1e-4_logit_bar': None if pc is None else pc['ok'], 'parity_check': pc}
        finally:
            nl.set_conv_math('fp32')
    return res


def parity_check(net, x, width, images=4):
    """IN-RUN correctness evidence, AFTER the timed region and outside it: the model the cycle just trained (its weights, BatchNorm
    statistics, piggymasks, head -- whatever state the cycle left) is copied into the CPU oracle (oracle.net.OracleVGG: checker only, never
    timed, never on the product path) and both run `images` images of the bench's own input pool forward, (a) in TRAIN mode -- the kernels
    the timed train steps ran: Winograd forward with the BatchNorm-statistics epilogue, fused BN -> ReLU (-> pool), the FC GEMMs; Dropout's p
    is 0 for the check, its masks come from different generators -- and (b) in EVAL mode (the inference epilogues the validates ran).
    north_star's bar: logits within 1e-4 (of the logit scale).  The conv arithmetic is whatever nl.set_conv_math says at the call.
    All three topologies (oracle/net.py: OracleVGG, OracleResNet, OracleSphereNet -- the A-Softmax head's (cos, phi) pair is compared
    element by element)."""
    import numpy as np
    from oracle import net as onet                     # checker only
    root = net
    kind = type(root).__name__
    if not hasattr(root, 'classifiers') or kind not in ('VGG', 'ResNet', 'SphereNet'):
        return None
    t0 = time.perf_counter()
    ref = {'VGG': lambda: onet.OracleVGG(width, 'imagenet'), 'ResNet': lambda: onet.OracleResNet(width),
           'SphereNet': lambda: onet.OracleSphereNet(width)}[kind]()
    for ds in root.datasets:
        ref.add_dataset(ds, root.dataset2num_classes[ds])
    cur = [i for i, c in enumerate(root.classifiers) if c is root.classifier][0]
    ref.set_dataset(root.datasets[cur])
    sd = {k: v.detach().cpu() for k, v in root.state_dict().items()}
    hip_layers = dict(root.named_modules())
    with torch.no_grad():
        for name, mod in ref.named_modules():
            if name in ('', 'head'):
                continue
            for pn, p in list(mod.named_parameters(recurse=False)) + list(mod.named_buffers(recurse=False)):
                key = name + '.' + pn
                if pn == 'piggymask':
                    continue
                p.copy_(sd[key])
            if isinstance(mod, onet._Masked):
                pm = getattr(hip_layers[name], 'piggymask', None)
                mod.piggymask = None if pm is None else nn.Parameter(pm.detach().cpu().clone())
                mod.threshold = float(hip_layers[name].info['threshold']) if hasattr(hip_layers[name], 'info') else mod.threshold
    xs = x[:images]
    xc = xs.detach().cpu()
    drops = [m for m in root.modules() if isinstance(m, nn.Dropout)]
    old_p = [m.p for m in drops]
    bns = [m for m in root.modules() if isinstance(m, nn.BatchNorm2d)]
    saved = [(m.running_mean.clone(), m.running_var.clone(), m.num_batches_tracked.clone()) for m in bns]
    was_training = root.training
    res = {}
    try:
        for m in drops:
            m.p = 0.0
        for m in ref.modules():
            if isinstance(m, nn.Dropout):
                m.p = 0.0
        for mode in ('train', 'eval'):
            root.train(mode == 'train')
            ref.train(mode == 'train')
            with torch.no_grad():
                got, want = root(xs), ref(xc)
            got = got if isinstance(got, tuple) else (got,)                     # (the A-Softmax head returns (cos, phi): both are compared)
            want = want if isinstance(want, tuple) else (want,)
            res[mode], res[mode + '_finite'] = 0.0, True
            for a_, b_ in zip(got, want):
                a_, b_ = a_.detach().float().cpu(), b_.detach()
                res[mode] = max(res[mode], float((a_ - b_).abs().max()) / max(float(b_.abs().max()), 1e-30))
                res[mode + '_finite'] = res[mode + '_finite'] and bool(torch.isfinite(a_).all())
    finally:
        for m, p_ in zip(drops, old_p):
            m.p = p_
        for m, (a_, b_, c_) in zip(bns, saved):          # (the train-mode pass moved the running statistics: put them back)
            m.running_mean.copy_(a_), m.running_var.copy_(b_), m.num_batches_tracked.copy_(c_)
        root.train(was_training)
    worst = max(res['train'], res['eval'])
    return {'max_rel_logit_err': float('%.3g' % worst), 'train_mode_forward': float('%.3g' % res['train']),
            'eval_mode_forward': float('%.3g' % res['eval']), 'bar': 1e-4, 'ok': bool(worst < 1e-4 and res['train_finite'] and res['eval_finite']),
            'images': int(xs.shape[0]), 'conv_math': nl.CONV_MATH,
            'oracle': 'oracle.net.Oracle%s on the host, state copied from the timed model after the cycle; outside the timed region' % kind,
            'host_seconds': round(time.perf_counter() - t0, 1)}


# --task-sequence: per topology the session builder, the task list (dataset, classes, finetune lr, prune-run lr, lr_mask) of the reference's
# script for that configuration, whether task 1 is the pretrained pass-through and whether tasks >= 2 end with the piggymask
# retrain (`--finetune_again`: experiment1 only)
SEQUENCES = {
    'vgg16': dict(builder='custom_vgg', pass_through_first=False, piggymask_retrain=True,
                  tasks=[('task%d' % i, 5, 1e-2, 1e-3, 5e-4) for i in range(1, 21)],                   # experiment1/CPG_cifar100_scratch_mul_1.5.sh:46-211
                  flow='finetune -> prune -> piggymask retrain per task, growth forced once', epochs='1 finetune + 10 prune + 1 retrain from task 2'),
    'resnet50': dict(builder='resnet50', pass_through_first=True, piggymask_retrain=False,                # experiment2/CPG_imagenet.sh:6-44
                     tasks=[('imagenet', 1000, 1e-3, 3e-4, 1e-4), ('cubs_cropped', 200, 1e-3, 1e-3, 1e-4), ('stanford_cars_cropped', 196, 1e-2, 1e-3, 1e-4),
                            ('flowers', 102, 1e-3, 1e-3, 1e-4), ('wikiart', 195, 1e-3, 1e-3, 1e-4), ('sketches', 250, 1e-3, 1e-3, 1e-4)],
                     flow="configs[3]'s flow: task 1 = pretrained pass-through (claim, validate, no training) + prune run; tasks >= 2 = finetune with "
                          'piggymasks + prune run', epochs='task 1: 10 prune; tasks >= 2: 1 finetune + 10 prune'),
    'spherenet20': dict(builder='spherenet20', pass_through_first=True, piggymask_retrain=False,          # experiment3/FvGeEm_CPG_face.sh:7-26,130
                        tasks=[('face_verification', 4630, 1e-3, 5e-4, 5e-4), ('gender', 3, 5e-4, 5e-4, 5e-4), ('emotion', 7, 5e-4, 5e-4, 5e-4),
                               ('age0', 8, 5e-4, 5e-4, 5e-4)],                                            # (FvGeEmAg0_CPG_face.sh:7-29: the fourth task)
                        flow="configs[4]'s flow: face_verification (AngleLinear + AngleLoss, pass-through + prune run, evaluated as embeddings) -> gender "
                             '(nn.Linear + CE) -> emotion (class-weighted CE) -> age (CE), per-task bias / PReLU stash', epochs='task 1: 10 prune; tasks >= 2: 1 finetune + 10 prune'),
}


def run_task_sequence(a, device):
    """--task-sequence T: T tasks back to back through cpg_amd.driver.CPGSession at the bench's full size (configs[1]: custom_vgg 224 x 224,
    batch 256), each with the section-8d cycle the headline times for task 1 -- finetune (1 epoch) -> gradual prune 0 -> 0.1 (10 epochs, rank-
    prune events in the first two) with a validate after every epoch -> choose the ratio -> (task >= 2) the piggymask retrain (1 epoch)
    -- and the reference's GROWTH forced on the last task (an accuracy goal no model reaches: exit code 2 -> raw multiplier + 0.5 ->
    sqrt -> 78 / 156 / 313 / 627 channels, the previous tasks' weights in the top-left corner, experiment1/CPG_cifar100_scratch_mul_1.5.sh:
    46-211).  Owner ids >= 3, piggymasks picking from several older tasks, growth mid-sequence and CPGSession itself at full size.
    Its own JSON line (never the headline): per task the wall time per train step, the owner-id histogram, shared_ratio, and the
    gradient payload a data-parallel exchange sends per step (owned slots, + the older tasks' slots for the piggymask gradients in finetune
    mode) against the dense one."""
    from cpg_amd.driver import CPGSession, default_args
    T = a.task_sequence
    E = max(2, a.steps // 11)                      # steps per epoch: 220 -> 20 (section 8d)
    plan = SEQUENCES[a.arch]
    if T > len(plan['tasks']):
        sys.exit('bench.py: --task-sequence %d: the %s sequence has %d tasks' % (T, a.arch, len(plan['tasks'])))
    sz = ARCHS[a.arch]['size']
    sess = CPGSession(plan['builder'], width_multiplier=a.width_multiplier, device=device, seed=1, freeze_gc=True)
    g = torch.Generator(device=device).manual_seed(1)
    xs = [torch.randn(a.batch, 3, sz, sz, generator=g, device=device) for _ in range(3)]
    xv = [torch.randn(100, 3, sz, sz, generator=g, device=device) for _ in range(2)]
    grow_at = a.grow_at_task if a.grow_at_task else (T if (a.arch == 'vgg16' and T >= 2) else 0)

    class Counting(object):
        def __init__(self, batches):
            self.batches, self.served = batches, 0

        def __len__(self):
            return len(self.batches)

        def __iter__(self):
            for b in self.batches:
                self.served += 1
                yield b
    tasks = []
    t_all = time.perf_counter()
    first_logits = {}
    invariant = {'checked': 0, 'bit_identical': True}
    # what each task's FIRST rank-prune event saw and did (device-side counts only): when more candidates are exact zeros than the rank k
    # asks for, the k-th smallest |w| is 0 and `abs(w) <= cutoff` (utils/prune.py:45) releases ALL of them -- the sparsity of that task then
    # reads above the schedule's target (round 5: 0.396 after a run to 0.1); tests/test_sequence_gpu.py holds the event bit-equal to the oracle
    first_events = {}
    orig_rank_prune = SparsePruner._rank_prune_layers

    def spy(self, ratio):
        first = self.prune_events == 0 and self.current_dataset_idx not in first_events
        zeros = sum(int(((m.weight.data == 0) & (self.masks[n] == self.current_dataset_idx)).sum()) for n, m in self._layers()) if first else 0
        recs = orig_rank_prune(self, ratio)
        if first:
            first_events[self.current_dataset_idx] = {
                'ratio': ratio, 'k_total': sum(r['k'] for r in recs), 'released_total': sum(r['n_released'] for r in recs),
                'owned_slots_exactly_zero_before': zeros, 'layers_with_cutoff_zero': sum(1 for r in recs if r['cutoff'] == 0.0),
                'released_beyond_k': sum(max(0, r['n_released'] - r['k']) for r in recs)}
        return recs
    SparsePruner._rank_prune_layers = spy
    for t in range(1, T + 1):
        name, ncls, lr_ft, lr_pr, lr_mask = plan['tasks'][t - 1]
        labels = [torch.randint(0, ncls, (a.batch,), generator=g, device=device) for _ in range(3)]
        vlabels = [torch.randint(0, ncls, (100,), generator=g, device=device) for _ in range(2)]
        train = Counting([(xs[i % 3], labels[i % 3]) for i in range(E)])
        val = [(xv[i], vlabels[i]) for i in range(2)]
        args = default_args(dataset=name, lr=lr_ft, lr_mask=lr_mask, prune_lr=lr_pr, pruning_frequency=max(1, E // 2), pruning_interval=2,
                            network_width_multiplier=sess.width)
        grow = t == grow_at
        passthrough = plan['pass_through_first'] and t == 1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = sess.run_task(name, ncls, train, val, accuracy_goal=2.0 if grow else 0.0, finetune_epochs=1, prune_epochs=10, sparsities=(a.sequence_sparsity,),
                            args=args, min_train_acc=-1.0, max_width_multiplier=(sess.width_multiplier + 0.5) if grow else None,
                            width_step=0.5, retrain_epochs=1, total_num_tasks=T, pretrained_pass_through=passthrough,
                            piggymask_retrain=plan['piggymask_retrain'])
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        # CPG's invariant, asserted in the run: every task learnt so far answers BIT-identically to when it was finished
        first_logits[name] = (val, sess.evaluate(name, val)[1])
        for older, (oval, outs0) in first_logits.items():
            outs = sess.evaluate(older, oval)[1]
            same = all(torch.equal(x_, y_) for x_, y_ in zip(outs0, outs))
            invariant['checked'] += 1
            invariant['bit_identical'] = invariant['bit_identical'] and same
            if not same:
                invariant.setdefault('broken', []).append('%s after %s' % (older, name))
        hist = torch.zeros(256, dtype=torch.int64, device=device)
        for m in sess.masks.values():
            hist += torch.bincount(m.reshape(-1).long(), minlength=256)
        hist = hist.cpu().tolist()
        n_all = sum(hist)
        pr = SparsePruner(sess.model, sess.masks, default_args(mode='inference', dataset=name, network_width_multiplier=sess.width), 0, 0, t)
        owned, older = hist[t], sum(hist[1:t])
        tasks.append({'task': t, 'dataset': name, 'num_classes': ncls, 'pass_through': passthrough, 'train_steps': train.served, 'wall_s': round(wall, 2),
                      'wall_ms_per_train_step': round(1000.0 * wall / max(1, train.served), 2),
                      'width_multiplier_raw': sess.width_multiplier, 'width_multiplier_rooted': round(sess.width, 6), 'grown_to': res.grown_to,
                      'masked_weights': n_all, 'owner_histogram': {str(i): c for i, c in enumerate(hist) if c},
                      'ratio_to_acc': {str(k): v for k, v in res.ratio_to_acc.items()}, 'chosen_ratio': res.chosen_ratio,
                      'prune_run_exit2_at': res.prune_exit2, 'retrain_kept': res.retrain_kept, 'shared_ratio': round(pr.calculate_shared_part_ratio(), 6) if t > 1 else None,
                      'sparsity': pr.calculate_sparsity(),
                      'dp_payload_bytes_per_step': {'dense_weights': 4 * n_all, 'prune_mode': 4 * owned,
                                                    'finetune_mode': 4 * (owned + (older if t > 1 else 0)),
                                                    'dense_with_piggymasks': (8 if t > 1 else 4) * n_all},
                      'channels': [int(m.weight.shape[0]) for m in sess.net.modules() if isinstance(m, (nl.SharableConv2d, nl.SharableLinear))][:14:3]})
    torch.cuda.synchronize()
    SparsePruner._rank_prune_layers = orig_rank_prune
    for t_ in tasks:
        t_['first_rank_prune_event'] = first_events.get(t_['task'])
    total = time.perf_counter() - t_all
    steps = sum(t['train_steps'] for t in tasks)
    # every earlier task still answers exactly as it did: evaluate task 1 on its own (cropped) network before / after is the driver test's
    # job (tests/test_driver_gpu.py); here: the logits of task 1 are finite and its head reads the narrow share of the features
    acc1, logits1 = sess.evaluate(plan['tasks'][0][0], [(xv[0], torch.zeros(100, dtype=torch.long, device=device))])
    finite = bool(all(torch.isfinite(p).all() for p in sess.net.parameters()))
    return {'metric': 'images/sec over a %d-task CPG sequence through CPGSession, %s %dx%d batch %d (NOT the headline metric: %s)'
                      % (T, a.arch, sz, sz, a.batch, plan['flow']),
            'valid': finite and invariant['bit_identical'], 'weights_finite': finite, 'earlier_tasks_bit_identical': invariant,
            'value': round(a.batch * steps / total, 2), 'unit': 'images/sec', 'n_gpus': 1, 'steps': steps, 'warmup': 0,
            'ms_per_step': round(1000.0 * total / steps, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'data': 'synthetic',
            'dtype': 'f32', 'config': {'workload': '%s, %d-task sequence %s (epochs of %d steps: %s, validate after every epoch), batch %d'
                                                   % (ARCHS[a.arch]['workload'], T, [p_[0] for p_ in plan['tasks'][:T]], E, plan['epochs'], a.batch),
                                       'arch': a.arch, 'tasks': T, 'epoch_steps': E, 'per_gpu_batch': a.batch, 'grow_at_task': grow_at,
                                       'prune_run_target_sparsity': a.sequence_sparsity,
                                       'start_width_multiplier_raw': a.width_multiplier},
            'tasks': tasks, 'task1_after_sequence': {'logits_finite': bool(all(torch.isfinite(o).all() for o in logits1)), 'accuracy': acc1},
            'note': 'wall time includes validates, snapshots, the ratio choice, the growth rebuild (last task: a finetune at the old width, then '
                    'the wider model) -- everything CPGSession.run_task does; the cycle-only step time of a task is the headline bench / --task 2'}


def cpu_plumbing_cycle(steps=220, batch=32):
    """BASELINE.json configs[0] / BASELINE.md section 4: the reference's own CPU-runnable case -- `custom_vgg_cifar100` (VGG16-BN at
    32 x 32, full width), batch 32, the WHOLE section-8d mini-cycle on the host through the oracle: 20 finetune steps (lr 1e-2), a
    200-step prune run 0 -> 0.1 (rank-prune events at steps 10 / 20 / 30 / 40, lr 1e-3), apply_mask + 2 eval batches of 100 + the
    sparsity statistic after every 20 steps.  Returns (images/sec, seconds, prune events, validates)."""
    from oracle import net as onet
    from oracle import ops as oops
    model, pruner, opt = onet.make_task1(1.0, 'cifar100', 'finetune', lr=1e-2, wd=4e-5)
    g = torch.Generator().manual_seed(1)
    pool = [(torch.randn(batch, 3, 32, 32, generator=g), torch.randint(0, 5, (batch,), generator=g)) for _ in range(3)]
    val = [torch.randn(100, 3, 32, 32, generator=g) for _ in range(2)]
    A, f = cycle_plan(steps)
    events = validates = 0
    t0 = time.time()
    for step in range(steps):
        if step == A:                                      # the prune run: same owner index, new schedule, lr 1e-3
            pruner.mode, pruner.begin, pruner.end, pruner.frequency = 'prune', 0, 4 * f, f
            pruner.initial, pruner.target, pruner.last_prune_step = 0.0, 0.1, 0
            opt = torch.optim.SGD(list(model.parameters()), lr=1e-3, momentum=0.9, nesterov=True)
        model.train()
        x, t = pool[step % len(pool)]
        before = pruner.last_prune_step
        onet.train_step(model, pruner, opt, x, t, prune_step=step - A if step >= A else None, torch_routing=True)
        events += int(step >= A and pruner.last_prune_step != before)
        pruner.sparsity()
        if (step + 1) % EPOCH_STEPS == 0:
            pruner.apply_mask()
            model.eval()
            with torch.no_grad():
                for v in val:
                    model(v)
                    pruner.sparsity()
                    oops.zero_ratio([pruner.owners[n] for n, _ in model.masked_layers()], 1.0)
            validates += 1
    dt = time.time() - t0
    return steps * batch / dt, dt, events, validates


def cpu_baseline(steps=220, batch=256, validates=11, prune_events=4, probe_batch=None, level='full', arch='vgg16'):
    """Oracle ("port") of the same cycle on the host cores, reported beside the GPU number (never the target).
    oracle/ is only ever used here as the measured CPU baseline.

    1. Thread setting: 3 timed train steps at `probe_batch` (32 under 'full', 64 under 'quick') under torch's default for the host (one
       thread per physical core); SURVEY 8d's os.cpu_count() (every SMT thread) gets one 8-image step first and the full 3-step probe only
       when that is within 2 x of the default's rate (it has been 4 - 6 x slower on every box); the faster is used below, both rates are fields.
    2. level 'full' (default): 2 timed train steps (after one warm-up step) at the GPU's own batch (256) under that setting -- section
       8d's configuration, no batch extrapolation; `value` is built on this rate.  level 'quick': step 2 is skipped and the probe
       rate is used (`extrapolated_from_probe_batch`: true).
    3. One rank-prune event over all 15 layers, one validate batch of 100 (apply_mask + eval forward).
    4. level 'full': BASELINE.md section 4's plumbing leg -- configs[0] (32 x 32, batch 32) through the whole mini-cycle on the host
       (`plumbing`).
    value = the cycle the GPU ACTUALLY ran (K train steps of `batch` images, its counted prune events and validates of 2 x 100
    images) priced with those CPU times.  Images come from the same seeded N(0,1) / randint generator as the GPU run's."""
    from oracle import net as onet
    from oracle import ops as oops
    if probe_batch is None:
        # 'full': the probe only picks the thread setting (value is built on the batch-256 sample), and every SMT thread is ~4 x slower --
        # 32 images keep that leg at ~1.5 min; 'quick' builds value on the probe itself: 64
        probe_batch = 32 if level == 'full' else 64
    default_threads = torch.get_num_threads()
    all_threads = os.cpu_count() or default_threads
    A = ARCHS[arch]
    model, pruner, opt, crit = onet.make_task1_net(arch, A['dataset'], A['classes'], 'finetune', lr=A.get('lrs', LRS)[0], wd=4e-5)
    model.train()
    g = torch.Generator().manual_seed(1)
    xfull = torch.randn(batch, 3, A['size'], A['size'], generator=g)
    tfull = torch.randint(0, A['classes'], (batch,), generator=g)
    x, t = xfull[:probe_batch].contiguous(), tfull[:probe_batch].contiguous()

    def timed_steps(xb, tb, n, warm=True):
        if warm:
            onet.train_step(model, pruner, opt, xb, tb, torch_routing=True, criterion=crit)      # allocations, primitive cache
        t0 = time.time()
        for _ in range(n):
            onet.train_step(model, pruner, opt, xb, tb, torch_routing=True, criterion=crit)
        return time.time() - t0

    settings = {}
    for i, (name, threads) in enumerate((('default', default_threads), ('cpu_count', all_threads))):
        if name == 'cpu_count' and threads == default_threads:
            settings[name] = dict(settings['default'])
            continue
        torch.set_num_threads(threads)
        if name == 'cpu_count':
            # every SMT thread has been 4 - 6 x SLOWER than one thread per core on every box so far (r05: 0.57 vs 3.38 images/s, 168 s of
            # host time for the 3-step probe): one step at 8 images first; only a setting within 2 x of the default gets the full probe
            x8, t8 = x[:8].contiguous(), t[:8].contiguous()
            dt = timed_steps(x8, t8, 1, warm=False)
            if 8 / dt < 0.5 * settings['default']['train_images_per_sec']:
                settings[name] = {'threads': threads, 'train_steps_timed': 1, 'probe_batch': 8, 'seconds': round(dt, 2),
                                  'train_images_per_sec': round(8 / dt, 3)}
                continue
        dt = timed_steps(x, t, 3, warm=(i == 0))
        settings[name] = {'threads': threads, 'train_steps_timed': 3, 'seconds': round(dt, 2), 'train_images_per_sec': round(probe_batch * 3 / dt, 3)}
    best = max(settings, key=lambda k: settings[k]['train_images_per_sec'])
    threads = settings[best]['threads']
    torch.set_num_threads(threads)
    probe_ips = settings[best]['train_images_per_sec']
    full = None
    if level == 'full':
        dt = timed_steps(xfull, tfull, 2, warm=True)          # (1 warm-up + 2 timed steps of ~70 s each: the step time repeats within 1 %)
        full = {'batch': batch, 'threads': threads, 'train_steps_timed': 2, 'seconds': round(dt, 2), 'train_images_per_sec': round(batch * 2 / dt, 3)}
    train_ips = full['train_images_per_sec'] if full else probe_ips
    # one rank-prune event (utils/prune.py:30-53 on every masked layer: boolean gather + k-th value + masked assign)
    t0 = time.time()
    for name, m in model.masked_layers():
        pruner.owners[name], _, _ = oops.rank_prune(m.weight.data.numpy(), pruner.owners[name], pruner.cur, 0.05)
    prune_s = time.time() - t0
    # one validate batch of 100 (apply_mask + eval forward)
    model.eval()
    xv = xfull[:100].contiguous()
    t0 = time.time()
    pruner.apply_mask()
    with torch.no_grad():
        # (config 5's validate is evalLFW: apply_mask + eval-mode embeddings, utils/manager.py:156-195)
        model.forward_to_embeddings(xv) if 'face_verification' in A['dataset'] else model(xv)
    val_s = time.time() - t0
    n_layers = len(model.masked_layers())
    del model, pruner, opt
    plumbing = None
    if level == 'full' and arch == 'vgg16':
        ips, secs, ev, nv = cpu_plumbing_cycle()
        plumbing = {'workload': 'configs[0]: custom_vgg_cifar100 (VGG16-BN 32x32, full width), batch 32, the full 220-step mini-cycle '
                                '(20 finetune + 200 prune-run steps, %d rank-prune events, %d validates of 2 x 100 images) on the host' % (ev, nv),
                    'value': round(ips, 2), 'unit': 'images/sec', 'seconds': round(secs, 1), 'cores': threads, 'kind': 'port'}
    torch.set_num_threads(default_threads)
    cycle_s = steps * batch / train_ips + prune_events * prune_s + validates * 2 * val_s
    return {'value': round(steps * batch / cycle_s, 3), 'unit': 'images/sec', 'cores': threads, 'kind': 'port',
            'threads_default': default_threads, 'threads_cpu_count': all_threads, 'threads_used_for_value': threads,
            'train_images_per_sec_default_threads': settings['default']['train_images_per_sec'],
            'train_images_per_sec_cpu_count_threads': settings['cpu_count']['train_images_per_sec'],
            'train_steps_timed': {k: v['train_steps_timed'] for k, v in settings.items()},
            'probe_batch': probe_batch, 'gpu_batch': batch, 'full_batch_sample': full, 'extrapolated_from_probe_batch': full is None,
            'train_images_per_sec': round(train_ips, 3), 'prune_event_s': round(prune_s, 2), 'validate_images_per_sec': round(100 / val_s, 2),
            'plumbing': plumbing,
            'sample': '3 train steps (fwd + bwd + gradient routing + SGD-nesterov) at batch %d under %d threads, a 1-step look at %d threads (3 steps when it '
                      'is within 2 x)%s + 1 rank-prune event '
                      'over the %d masked layers (%.1f s) + 1 validate batch of 100 (apply_mask + eval forward, %.1f s) of the oracle %s %dx%d, '
                      'torch-CPU fp32; value = the %d-step cycle the GPU ran (%d prune events, %d validates of 2 x 100 images) priced with the '
                      '%s train rate'
                      % (probe_batch, default_threads, all_threads,
                         (', then 1 warm-up + 2 timed train steps at batch %d under %d threads' % (batch, threads)) if full else '',
                         n_layers, prune_s, val_s, {'vgg16': 'VGG16-BN', 'resnet50': 'ResNet-50', 'spherenet20': 'SphereNet-20 (AngleLinear head + AngleLoss)'}[arch],
                         A['size'], A['size'], steps, prune_events, validates, 'batch-%d' % batch if full else 'probe-batch')}


# ---- 8-GPU prediction: what a SCALE run should show, so that a first hardware run can be judged in one read
XGMI_LINK_GBS = 153.0                # MI355X_MICROARCH / task statement: 7 links x ~153 GB/s per GPU, point to point
RCCL_ALLREDUCE_BUSBW_GBS = 310.0     # assumed large-message bus bandwidth of an 8-GPU RCCL all-reduce over xGMI (two links' worth)
OVERLAP_SLOWDOWN = 1.17              # conv kernels beside 16 busy CUs' worth of other streams' kernels (tools/attic/diag_interference.py)
# cycle ms per step on ONE GPU by per-GPU batch (profiles/r04c_bench*.json; the 128 / 64 / 32 rows are the reference's own split: 256 images over 2 / 4 / 8 GPUs)
SINGLE_GPU_MS_PER_STEP = {'vgg16': {256: 109.3, 128: 57.5, 64: 31.7, 32: 18.7}, 'resnet50': {256: 69.8}, 'spherenet20': {256: 20.6}}
# What the exchange machinery costs a step BEFORE any link time, measured on one GPU with a world-1 RCCL group (CPG_DP_FORCE=1: every hook,
# row-block hand-over, packed / coalesced message, finish_gradient_sync, RCCL's copy kernels on their own stream):
# VGG16 111.6 vs 108.7 ms, ResNet-50 71.7 vs 69.8.  SphereNet-20 was not measured: scaled from ResNet-50 by gradient bytes (89 MB / 94 MB).
WORLD1_EXCHANGE_FLOOR_MS = {'vgg16': 2.9, 'resnet50': 1.9, 'spherenet20': 1.8}


def single_gpu_ms(arch, batch):
    """Measured single-GPU cycle time per step at this per-GPU batch; batches without a measurement are scaled from the nearest larger one
    (an optimistic guess: small batches lose efficiency)."""
    tab = SINGLE_GPU_MS_PER_STEP[arch]
    if batch in tab:
        return tab[batch]
    near = min((b for b in tab if b >= batch), default=max(tab))
    return tab[near] * batch / near


def predict_step_ms(arch, world, buckets, measured_single_gpu_ms=None, batch=256):
    """Weak scaling (256 images per GPU): step = single-GPU step + what the gradient exchange adds.  All but the LAST messages
    run under the remaining backward kernels (which slow down by OVERLAP_SLOWDOWN while RCCL holds CUs); the last message -- the
    coalesced small tensors, issued after backward -- is exposed."""
    t1 = measured_single_gpu_ms or single_gpu_ms(arch, batch)
    if world <= 1:
        return {'predicted_ms_per_step': t1, 'allreduce_ms_total': 0.0, 'exposed_ms': 0.0}
    algbw = RCCL_ALLREDUCE_BUSBW_GBS * world / (2.0 * (world - 1))                  # GB/s of payload
    per_msg_latency_ms = 0.03
    total_ms = sum(b / 1e9 / algbw * 1e3 + per_msg_latency_ms for _, b in buckets)
    last_ms = (buckets[-1][1] / 1e9 / algbw * 1e3 + per_msg_latency_ms) if buckets else 0.0
    hidden = total_ms - last_ms
    # the interference term can never be below the measured world-1 floor (the same kernels beside the backward with zero wire time)
    floor = WORLD1_EXCHANGE_FLOOR_MS.get(arch, 0.0)
    interference = max(hidden * (OVERLAP_SLOWDOWN - 1.0), floor)
    return {'predicted_ms_per_step': round(t1 + interference + last_ms, 3), 'allreduce_ms_total': round(total_ms, 3),
            'exposed_ms': round(last_ms + interference, 3), 'assumed_algbw_GBs': round(algbw, 1),
            'assumed_busbw_GBs': RCCL_ALLREDUCE_BUSBW_GBS, 'single_gpu_ms_per_step': t1, 'world1_exchange_floor_ms': floor,
            'model': 'step(N) = step(1) + max((overlapped all-reduce time) x (%.2f - 1), measured world-1 exchange floor) + last message; '
                     'messages priced at payload / algbw + 30 us' % OVERLAP_SLOWDOWN}


OTHER_WORKLOADS = [('resnet50', ['--arch', 'resnet50']), ('spherenet20', ['--arch', 'spherenet20']),
                   ('vgg16_grown_1.5', ['--width-multiplier', '1.5']), ('vgg16_task2', ['--task', '2'])]


def other_workloads(steps, warmup, timeout_s=240):
    """The other single-GPU workloads of BASELINE.json's configs through the SAME cycle, each in its own `python bench.py` process started
    after the headline's timed region (this process idles meanwhile; nothing of it is inside any timed region): configs[3]'s topology
    (ResNet-50), configs[4]'s (SphereNet-20, at the reference's lrs), the GROWN VGG16 (raw multiplier 1.5: where most of configs[1]'s 20
    tasks run) and the task-2 cycle (piggymasks + Adam).  A summary of each child's own JSON line; the full line is what
    `python bench.py <flags>` prints."""
    import subprocess
    res = {}
    for name, flags in OTHER_WORKLOADS:
        cmd = [sys.executable, os.path.abspath(__file__)] + flags + ['--steps', str(steps), '--warmup', str(warmup), '--no-cpu-baseline',
                                                                     '--optin-steps', '0', '--no-other-workloads']
        t0 = time.perf_counter()
        try:
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s)
            line = [ln for ln in p.stdout.decode(errors='replace').splitlines() if ln.startswith('{')]
            if p.returncode != 0 or not line:
                res[name] = {'error': 'rc %d: %s' % (p.returncode, p.stderr.decode(errors='replace')[-300:]), 'flags': flags}
                continue
            d = json.loads(line[-1])
            ws, rf, pc, cc = d.get('whole_step') or {}, d.get('roofline') or {}, d.get('parity_check') or {}, d.get('cycle_check') or {}
            res[name] = {'flags': ' '.join(flags), 'metric': d['metric'], 'value': d['value'], 'unit': d['unit'], 'valid': d.get('valid'),
                         'ms_per_step': d['ms_per_step'], 'steps': d['steps'],
                         'whole_step': {'frac_of_dense_fp32_mfma_peak': ws.get('frac_of_dense_fp32_mfma_peak')},
                         'roofline': {'kernel': rf.get('kernel'), 'frac': rf.get('frac'), 'achieved': rf.get('achieved'), 'peak': rf.get('peak')},
                         'parity_check': {'ok': pc.get('ok'), 'max_rel_logit_err': pc.get('max_rel_logit_err')},
                         'cycle_check': {'weights_finite': cc.get('weights_finite'), 'prune_events': cc.get('prune_events')},
                         'per_gpu_batch': (d.get('config') or {}).get('per_gpu_batch'), 'process_seconds': round(time.perf_counter() - t0, 1)}
            if d.get('task2'):
                res[name]['task2_over_task1'] = d['task2'].get('task2_over_task1')
        except subprocess.TimeoutExpired:
            res[name] = {'error': 'timed out after %d s' % timeout_s, 'flags': flags}
        except Exception as e:                                # a broken child must not take the headline line with it
            res[name] = {'error': repr(e)[:300], 'flags': flags}
    return res


def _free_port():
    import socket
    sk = socket.socket()
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=220, help='timed train steps (220 = the full section-8d cycle)')
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=256, help='per-GPU batch (config 2: 256)')
    ap.add_argument('--global-batch', type=int, default=0,
                    help="STRONG scaling, the reference's own data-parallel split (CPG_cifar100_main_normal.py:112-114,199: the batch "
                         'is NOT scaled with the GPUs): this many images per step over all ranks, per-GPU batch = global / N; overrides '
                         '--batch and reports "scaling": "strong"')
    ap.add_argument('--task', type=int, default=1, choices=[1, 2],
                    help='1 = the headline (task 1: no piggymask).  2 = the cycle 19 of the 20 tasks of configs[1] run: owner masks of a '
                         'finished task 1 (30 %% of every layer free), a piggymask on every masked layer, SGD + Adam(lr_mask); its own line, '
                         'never the headline; also times a task-1 cycle of the same length in the same process (task1_ms_per_step)')
    ap.add_argument('--width-multiplier', type=float, default=1.0,
                    help="the reference's RAW --network_width_multiplier (main() takes its square root, CPG_cifar100_main_normal.py:115): 1.5 = "
                         'the GROWN network most of the 20 tasks of configs[1] run in (experiment1/CPG_cifar100_scratch_mul_1.5.sh:90-94: '
                         'int(v * 1.2247) = 78 / 156 / 313 / 627 channels, 30723 -> 5016 -> 5016 FC); its own line, never the headline')
    ap.add_argument('--task-sequence', type=int, default=0,
                    help='T > 0: T tasks back to back through cpg_amd.driver.CPGSession at full size, growth forced on the last one; prints its '
                         'own JSON line (per-task step time, owner-id histogram, shared_ratio, data-parallel payload) instead of the cycle bench')
    ap.add_argument('--sequence-sparsity', type=float, default=0.1,
                    help="--task-sequence: target of every task's one gradual-prune run (default 0.1, the first run of the reference's sweep). "
                         'The reference sweeps on to 0.9 / 0.95 and keeps the sparsest stage that holds the accuracy goal, so a task normally '
                         'hands MOST of its slots on; at 0.1 the free capacity shrinks 10 x per task and the smallest layer (1 728 weights) runs out '
                         'of candidates at task 4 (exit code 2, utils/prune.py:38-42): use 0.5 for sequences longer than 3 tasks')
    ap.add_argument('--grow-at-task', type=int, default=0,
                    help='--task-sequence: force the growth step (exit code 2 -> raw multiplier + 0.5) on this task (default: the last task of a '
                         'vgg16 sequence, never on the other topologies)')
    ap.add_argument('--arch', default='vgg16', choices=sorted(ARCHS),
                    help="topology of the cycle: 'vgg16' = the headline (BASELINE.json configs[1]); 'resnet50' / 'spherenet20' = the "
                         'topologies of configs[3] / configs[4] through the same cycle (their own lines, never the headline)')
    ap.add_argument('--math', default='fp32', choices=['fp32', 'bf16', 'bf16x3'],
                    help="arithmetic of the 3x3 conv forward / input gradient: 'fp32' (default, the reference's precision) or the "
                         "OPT-IN 'bf16' MFMA path (never the headline: it does not meet north_star's 1e-4 parity bar)")
    ap.add_argument('--optin-steps', type=int, default=8,
                    help='after the timed cycle, also time this many train steps in each opt-in conv arithmetic (reported beside the '
                         'headline as opt_in_conv_math, never as value); 0 = skip')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-other-workloads', action='store_true',
                    help='the default headline run (vgg16, task 1, width 1.0, one GPU) also runs 20-step cycles of the other single-GPU '
                         "workloads (resnet50, spherenet20, the grown VGG16, task 2) in child processes after its own timed region and reports "
                         'them under "other_workloads"; this flag skips that')
    ap.add_argument('--cpu-baseline', default='full', choices=['full', 'quick'],
                    help="'full' (default, ~6 min of host time): SURVEY 8d's CPU baseline -- 2 timed train steps at batch 256 on the faster thread "
                         "setting + BASELINE.md section 4's configs[0] plumbing cycle; 'quick': the batch-64 probe only (~1.5 min)")
    ap.add_argument('--no-kernel-clock', action='store_true')
    ap.add_argument('--clock-every', type=int, default=4,
                    help='HIP events around the masked-layer launches of every N-th train step of the timed cycle and of every validate '
                         '(1 = every launch; the launches of the other steps are counted, not timed)')
    a = ap.parse_args()

    if a.global_batch:
        if a.global_batch % a.gpus:
            sys.exit('bench.py: --global-batch %d is not divisible by --gpus %d' % (a.global_batch, a.gpus))
        a.batch = a.global_batch // a.gpus
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ and 'RANK' not in os.environ:
        # started as a plain `python bench.py --gpus N`: become N ranks, one per GPU
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(a.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != a.gpus:
        sys.exit('bench.py: --gpus %d but WORLD_SIZE is %d; launch with torch.distributed.run --nproc-per-node %d '
                 '(or plain `python bench.py --gpus %d`, which re-launches itself)' % (a.gpus, world, a.gpus, a.gpus))
    backend = None
    if world > 1 or os.environ.get('CPG_DP_FORCE') == '1':
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        ndev = torch.cuda.device_count()
        # 'nccl' is RCCL on ROCm.  CPG_BENCH_BACKEND=gloo lets several ranks share one GPU for a functional test
        # of the multi-process path on a single-GPU box (not a performance configuration).
        backend = os.environ.get('CPG_BENCH_BACKEND', 'nccl')
        if backend == 'nccl' and world > ndev:
            sys.exit('bench.py: %d ranks but only %d GPUs visible (RCCL needs one GPU per rank)' % (world, ndev))
        torch.cuda.set_device(local % ndev)
        dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    device = torch.device('cuda', torch.cuda.current_device())

    if a.task_sequence > 0:
        if world != 1:
            sys.exit('bench.py: --task-sequence runs on one GPU')
        print(json.dumps(run_task_sequence(a, device)), flush=True)
        return
    global DATASET, WIDTH, LRS
    arch = ARCHS[a.arch]
    DATASET = arch['dataset']
    LRS = arch.get('lrs', LRS)
    WIDTH = 1.0 if a.width_multiplier == 1.0 else a.width_multiplier ** 0.5
    nl.set_conv_math(a.math)
    from cpg_amd import _lib
    clock = KernelClock()
    clock.every = max(1, a.clock_every) if a.steps >= 4 * max(1, a.clock_every) else 1     # (short runs: every launch)
    if os.environ.get('ROCPROF_COUNTER_COLLECTION', '').lower() not in ('', '0', 'false', 'off'):
        # under `rocprofv3 --pmc` a queue that carries events on some steps only was aborted with HSA_STATUS_ERROR_INVALID_PACKET_FORMAT
        # (ResNet-50, ROCm 7.2; every launch clocked, or none, runs): counter passes clock every launch
        clock.every = 1
    if not a.no_kernel_clock and rank == 0:
        proxy = clock.wrap(_lib.lib())
        _lib._lib = proxy                                  # route the Python mirror's calls through the timers

    net = build_model(device, a.arch)                      # every rank: the reference's seed-1 initial weights
    model = cdist.DataParallel(net)
    dropout_seed = cdist.seed_per_rank(1)                  # Dropout masks differ per rank, as nn.DataParallel's replicas' do
    model.sync_events = [] if world > 1 else None
    masks = {n: torch.zeros(m.weight.shape, dtype=torch.uint8, device=device) for n, m in model.named_modules()
             if isinstance(m, (nl.SharableConv2d, nl.SharableLinear))}

    g = torch.Generator(device=device).manual_seed(1 + rank)          # each rank its own shard of the global batch
    sz, ncls = arch['size'], arch['classes']
    pool = [(torch.randn(a.batch, 3, sz, sz, generator=g, device=device),
             torch.randint(0, ncls, (a.batch,), generator=g, device=device)) for _ in range(3)]
    val_pool = [(torch.randn(100, 3, sz, sz, generator=g, device=device),
                 torch.randint(0, ncls, (100,), generator=g, device=device)) for _ in range(2)]

    def warm_up(n, lr_mask=None):
        """n untimed train steps on a COPY of the owner masks with every learning rate 0 (weights, masks and -- restored below -- BatchNorm
        statistics unchanged): allocator, first-launch code loading.  The last one (n >= 2) is a prune-mode step with a rank-prune
        event, so that the kernels of an event and of the optimizers' prune-mode paths are not loaded inside the timed prune window;
        before it one validate (eval-mode kernels)."""
        wmasks = {k: v.clone() for k, v in masks.items()}
        wm = Manager(make_args('finetune', 1), model, {}, wmasks, [pool[i % len(pool)] for i in range(n - 1 if n >= 2 else n)], val_pool, 0, 0)
        wm.pruner.make_finetuning_mask()
        wm.train(make_optimizers(model, wm.pruner, 0.0, lr_mask), 0, [0.0], 0)
        validate(wm, 0)                                   # (every slot is owned by a task <= the current one: apply_mask changes nothing)
        if n >= 2:                                        # AFTER the validate: the event releases slots of the mask copy, which an
            wp = Manager(make_args('prune', 1), model, {}, wmasks, [pool[(n - 1) % len(pool)]], val_pool, 0, 4)     # apply_mask would zero
            wp.train(make_optimizers(model, wp.pruner, 0.0, lr_mask), 0, [0.0], 1)       # (prune step 1 of a window of 4: an event)

    # warm-up: W untimed train steps + one validate
    if a.warmup > 0:
        warm_up(a.warmup)
        for bn in model.modules():                        # lr = 0 keeps weights; also restore BN statistics
            if isinstance(bn, nn.BatchNorm2d):
                bn.reset_running_stats()
        if model.sync_events is not None:
            model.sync_events = []
    # what cpg_amd.driver.CPGSession does once a task's model stands: one full collection, then everything alive goes to the
    # collector's permanent generation (a generation-2 pass over torch + the model is 80 ms of host time with no kernel enqueued)
    settle_host_gc()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    task1_ms = free_share = None
    if a.task == 2:
        # the comparison leg: a task-1 cycle of the same length in this process (its own barrier-bracketed clock, kernels not clocked)
        barrier()
        t1 = time.perf_counter()
        run_cycle(model, masks, pool, val_pool, a.steps)
        barrier()
        task1_ms = 1000.0 * (time.perf_counter() - t1) / a.steps
        free_share = begin_task2(model, masks, arch, device)
        # task 2 has its own label space (a new head with classes2 outputs): same images, labels of that range
        pool = [(x, torch.randint(0, arch['classes2'], (x.shape[0],), generator=g, device=device)) for x, _ in pool]
        val_pool = [(x, torch.randint(0, arch['classes2'], (x.shape[0],), generator=g, device=device)) for x, _ in val_pool]
        # warm the task-2 kernels (pack passes with the binarizer, piggymask-gradient epilogues, fused Adam in both modes, shared_ratio)
        warm_up(max(2, min(3, a.warmup)), lr_mask=0.0)
        if model.sync_events is not None:
            model.sync_events = []
        settle_host_gc()

    barrier()
    clock.enabled = True
    t0 = time.perf_counter()
    marks, counts = [], {}
    done = run_cycle(model, masks, pool, val_pool, a.steps, clock, marks, counts, task=a.task)
    barrier()
    dt_local = dt = time.perf_counter() - t0
    clock.enabled = False
    assert done == a.steps
    per_rank_ms = [1000.0 * dt / a.steps]
    if world > 1:
        tt = torch.zeros(world, device=device, dtype=torch.float64)
        tt[rank] = dt_local
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        per_rank_ms = [round(1000.0 * float(v) / a.steps, 3) for v in tt.tolist()]
        dt = float(tt.max().item())
        # replicated state must still be identical: compare a checksum of the weights and owner masks across ranks
        chk = torch.stack([sum(p.detach().double().sum() for p in net.parameters()),
                           sum(m.double().sum() for m in masks.values())])
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        replicas_identical = bool(torch.equal(lo, hi))
    again_ms = phases = None
    if a.task == 2:
        # (every rank: the leg's train steps exchange gradients; after the timed region, the replica checksum and the phase report,
        # which prices a rank-prune event on the cycle's final masks)
        phases = phase_report(marks, model, masks, a.batch) if rank == 0 else None
        again_ms = finetune_again_leg(model, masks, pool, val_pool, max(2, min(a.steps, 10)))
    if rank == 0:
        global_batch = a.batch * world
        value = global_batch * a.steps / dt
        A, f = cycle_plan(a.steps)
        metric = ('images/sec per CPG train-prune-retrain cycle, VGG16 task-1' if a.arch == 'vgg16' else
                  'images/sec per CPG train-prune-retrain cycle, %s (NOT the headline metric: the same cycle on another topology)' % a.arch)
        if a.task != 1:
            metric = ('images/sec per CPG train-prune-retrain cycle, %s task-%d (NOT the headline metric: the cycle of tasks >= 2 -- a piggymask '
                      'on every masked layer, SGD + Adam)' % (a.arch, a.task))
        if a.batch != 256 and a.task == 1 and a.arch == 'vgg16':
            metric += ' (NOT the headline configuration: %d images per GPU instead of 256)' % a.batch
        if a.width_multiplier != 1.0:
            metric += (' (NOT the headline configuration: the network grown to raw width multiplier %g = x %.4f channels per layer)'
                       % (a.width_multiplier, WIDTH))
        out = {'metric': metric, 'value': round(value, 2),
               'unit': 'images/sec', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
               'ms_per_step': round(1000.0 * dt / a.steps, 3), 'higher_is_better': True, 'scaling': 'strong' if a.global_batch else 'weak',
               'vs_baseline': None, 'data': 'synthetic',
               'dtype': 'f32' if a.math == 'fp32' else
               ('bf16 operands' if a.math == 'bf16' else 'bf16x3 (two-term bf16 split of every operand, 3 MFMAs per product)')
               + ' / f32 accumulate in the 3x3 convolutions (OPT-IN, not the headline); stem, linear layers, BatchNorm, optimizer f32',
               'config': {'workload': '%s, task-%d CPG cycle (finetune -> prune 0.0->0.1 -> recovery, validate after every 20th train '
                                      'step), batch %d per GPU' % (arch['workload'], a.task, a.batch), 'task': a.task,
                          'arch': a.arch, 'width_multiplier_raw': a.width_multiplier, 'width_multiplier_rooted': WIDTH,
                          'masked_layer_channels': [int(m.weight.shape[0]) for m in net.modules()
                                                    if isinstance(m, (nl.SharableConv2d, nl.SharableLinear))] if a.width_multiplier != 1.0 else None,
                          'global_batch': global_batch, 'per_gpu_batch': a.batch, 'parallelism': 'dp%d' % world,
                          'epoch_steps': EPOCH_STEPS, 'cycle': counts, 'lr_finetune': LRS[0], 'lr_prune_run': LRS[1],
                          'host_gc': 'collected + frozen after the warm-up (cpg_amd.utils.settle_host_gc, as CPGSession.start_task)'},
               # train steps only, in the ALGORITHMIC flops of SURVEY 8d (Winograd launches execute 16/36 of them, so this can
               # exceed the dense peak; whole_step below prices the step against what the MFMA pipe really had to do)
               # (the per-image constant is the width-1.0 network's; other widths: see whole_step.algorithmic_tflops, from the launches)
               'algorithmic_tflops_train_steps': round(value * arch['flop_train'] / world / 1e12, 2) if a.width_multiplier == 1.0 else None}
        if world > 1:
            ev = model.sync_events or []
            sync_ms = sum(s.elapsed_time(e) for s, e in ev)
            out['multi_gpu'] = {'backend': 'rccl' if backend == 'nccl' else backend, 'rccl_ranks': world if backend == 'nccl' else 0,
                                'per_rank_ms_per_step': per_rank_ms,
                                'exposed_allreduce_ms_per_step': round(sync_ms / max(1, len(ev)), 3),
                                'allreduced_gradient_bytes_per_step': int(sum(p.numel() for p in net.parameters() if p.requires_grad) * 4),
                                'dropout_seed_rank0': dropout_seed, 'replicas_identical_after_cycle': replicas_identical,
                                'shared_chip_hint': getattr(model, 'shared_chip_hint', 0)}
            # the messages of one train step in launch order (kind, bytes): 'chunk' = a row block of a very large linear weight
            # handed over while its producer still runs, 'tensor' = one large gradient, 'packed' = the surviving slots of a
            # layer, 'coalesced' = all small tensors (BatchNorm, biases, head) in one message after backward
            buckets = list(model.last_bucket_log)
            out['multi_gpu']['buckets'] = [{'kind': k, 'bytes': b} for k, b in buckets]
            out['multi_gpu'].update(predict_step_ms(a.arch, world, buckets, batch=a.batch))
        agg = clock.summary(a.steps)
        if agg:
            tot_ms = sum(v[1] for v in agg.values())
            fam = {}
            for kind, (cnt, ms, fl, ex, nb) in agg.items():
                fk = fam.setdefault(kind.split(' ')[0], [0, 0.0, 0.0, 0.0, 0.0])
                fk[0] += cnt
                fk[1] += ms
                fk[2] += fl
                fk[3] += ex
                fk[4] += nb
            dom = max(fam, key=lambda k: fam[k][1])
            cnt, ms, fl, ex, nb = fam[dom]
            ach = fl / (ms * 1e-3) / 1e12                      # algorithmic flops (SURVEY 8d units) per second
            exe = ex / (ms * 1e-3) / 1e12                      # multiply-adds the MFMA pipe really executed, as flops per second
            traffic = pmc_traffic(a.arch, dom, a.batch) if a.width_multiplier == 1.0 else None     # (the committed passes ran at width 1.0)
            dense = PEAK_BF16_MFMA_TFLOPS if dom.endswith('_bf16') else PEAK_FP32_MFMA_TFLOPS
            # The ceiling of THIS launch mix: a launch that runs Winograd F(2x2,3x3) needs 16/36 of its algorithmic multiply-adds,
            # so its algorithmic ceiling is 2.25 x the dense MFMA peak; a direct launch's is the dense peak.  Weighted by MFMA
            # time that is dense_peak x (algorithmic flops / executed flops) -- and achieved / peak == executed rate / dense peak.
            peak = dense * fl / ex
            out['roofline'] = {'bound': 'mfma', 'kernel': dom, 'achieved': round(exe, 2), 'peak': dense,
                               'unit': 'TFLOP/s', 'frac': round(exe / dense, 4),
                               'achieved_algorithmic': round(ach, 2), 'launch_mix_ceiling': round(peak, 2),
                               'winograd_share_of_algorithmic_flops': round((fl - ex) / (fl * (1 - 16.0 / 36.0)), 4),
                               'note': 'achieved = multiply-adds the MFMA pipe EXECUTED (x 2) / HIP-event time of the launches; peak = the dense fp32 '
                                       'MFMA peak of the guide.  achieved_algorithmic counts SURVEY 8d\'s flops instead (a Winograd F(2x2,3x3) launch '
                                       'executes 16/36 of them, so it can exceed the dense peak); launch_mix_ceiling = dense peak x algorithmic / '
                                       'executed.  frac = achieved / peak = achieved_algorithmic / launch_mix_ceiling',
                               'traffic': (traffic or {}).get('hbm_bytes_per_launch'),
                               'traffic_source': 'static: %s (rocprofv3 --pmc passes of this workload, committed; not collected in this run)'
                                                 % traffic['source'] if traffic else None,
                               'traffic_stale': traffic['stale'] if traffic and 'stale' in traffic else None,
                               'traffic_detail': traffic,
                               # SURVEY 8d's bytes: each launch reads two of {x, y or gy, W} once and writes the third once
                               'algorithmic_bytes_per_launch': round(nb / cnt),
                               'traffic_over_algorithmic': round(traffic['hbm_bytes_per_launch'] * cnt / nb, 3) if traffic else None,
                               'launches': cnt, 'avg_launch_ms': round(ms / cnt, 4),
                               'share_of_masked_kernel_time': round(ms / tot_ms, 3)}
            # the whole timed region against the dense peak, in executed multiply-adds (every masked launch, train + validate)
            # (every launch of the region: the clocked ones un-weighted + the ones that were only counted)
            ex_all = sum(r[4] for r in clock.records) + clock.unclocked[2]
            fl_all = sum(r[1] for r in clock.records) + clock.unclocked[1]
            out['whole_step'] = {'mfma_tflops_executed': round(ex_all / dt / 1e12, 2),
                                 'frac_of_dense_fp32_mfma_peak': round(ex_all / dt / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                                 'algorithmic_tflops': round(fl_all / dt / 1e12, 2),
                                 'frac_of_launch_mix_ceiling': round(ex_all / dt / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                                 'launch_mix_ceiling_tflops': round(PEAK_FP32_MFMA_TFLOPS * fl_all / ex_all, 2),
                                 'note': 'all masked conv / linear launches of the timed region (train steps and validates; clocked or only '
                                         'counted) over the wall time'}
            out['kernel_clock'] = {'every': clock.every, 'train_steps_clocked': clock.train_steps_clocked,
                                   'launches_clocked': clock.launches_clocked, 'launches_counted_only': clock.unclocked[0],
                                   'note': 'HIP events around the masked-layer launches of every N-th train step of the cycle and of every '
                                           'validate (an event pair costs the queue ~ 7 us); roofline / kernel_families are averages over the '
                                           'clocked launches, a train-step launch weighted by train steps / clocked train steps (a stratified '
                                           'sample of the region: launch counts and ms are the estimates for all of it)'}
            out['kernel_families'] = {k: dict({'launches': v[0], 'ms': round(v[1], 2), 'tflops': round(v[2] / (v[1] * 1e-3) / 1e12, 2),
                                               'mfma_tflops_executed': round(v[3] / (v[1] * 1e-3) / 1e12, 2),
                                               'algorithmic_bytes_per_launch': round(v[4] / v[0]),
                                               # what those bytes alone take at the 6.3 TB/s a float4 copy reaches on this part, beside the
                                               # launch's measured average: the families that stream a weight (linear_* at task >= 2 / small
                                               # batches) are to be read against THIS floor, not against the MFMA peak
                                               'avg_launch_ms': round(v[1] / v[0], 4), 'hbm_floor_ms_per_launch': round(v[4] / v[0] / 6.3e12 * 1e3, 4),
                                               'frac_of_dense_peak_executed': round(v[3] / (v[1] * 1e-3) / 1e12 /
                                                                                    (PEAK_BF16_MFMA_TFLOPS if k.endswith('_bf16') else PEAK_FP32_MFMA_TFLOPS), 4)})
                                      for k, v in sorted(fam.items())}
            if any(v[3] < v[2] for v in fam.values()):
                out['winograd_note'] = ('launches for which cpg_conv2d_winograd() answers 1 (3x3 s1 p1 convs on even maps with >= 16 channels: forward, '
                                        'input gradient, weight gradient, inference epilogue) run Winograd F(2x2,3x3): "tflops" counts the ALGORITHMIC '
                                        'flops of SURVEY section 8d, "mfma_tflops_executed" the multiply-adds the MFMA pipe really performed (16/36 of them)')
            out['masked_kernel_ms_per_step'] = round(clock.train_ms / max(1, clock.train_steps_clocked), 2)     # (train steps; validates apart)
            if os.environ.get('CPG_BENCH_DETAIL'):
                out['kernel_detail'] = {k: {'n': v[0], 'ms': round(v[1], 2), 'tflops': round(v[2] / (v[1] * 1e-3) / 1e12, 2), 'winograd': v[3] < v[2]}
                                        for k, v in sorted(agg.items())}
        out['phases'] = phases if phases is not None else phase_report(marks, model, masks, a.batch)
        # in-run correctness evidence (outside the timed region): finite loss, the sparsity the cycle's rank-prune events must have reached,
        # and the timed model's logits against the CPU oracle
        spars = SparsePruner(model, masks, make_args('prune', 1), 0, 1, 1).calculate_sparsity() if a.task == 1 else None
        out['cycle_check'] = {'prune_events': counts.get('prune_events'), 'sparsity_after_cycle': spars,
                              'expected_sparsity': 0.1 if (a.task == 1 and counts.get('prune_events', 0) >= 4) else None,
                              'weights_finite': bool(all(torch.isfinite(p).all() for p in net.parameters()))}
        out['valid'] = True
        if not out['cycle_check']['weights_finite']:
            # a cycle that left NaN / inf weights times kernels on garbage: no throughput is reported for it (machine-readable: valid false,
            # value null, the measured number under invalid_value) and the metric string says so too
            out['metric'] += ' (INVALID RUN: non-finite weights after the cycle)'
            out['valid'], out['invalid_value'], out['value'] = False, out['value'], None
        out['parity_check'] = parity_check(net, pool[0][0], WIDTH) if world == 1 else None
        if a.task == 2:
            out['task2'] = {'task1_ms_per_step': round(task1_ms, 3), 'task2_over_task1': round(1000.0 * dt / a.steps / task1_ms, 4),
                            'free_share_handed_to_task2': round(free_share, 4), 'lr_mask_finetune': 5e-4, 'lr_mask_prune': 0.0,
                            'finetune_again_ms_per_step': round(again_ms, 3),
                            'note': 'task1_ms_per_step = the task-1 cycle of the same K in the same process, before the switch; the timed '
                                    'region is the task-2 cycle alone; finetune_again = the piggymask retrain leg (lr 1e-3, lr_mask 1e-4, '
                                    'fresh piggymasks), train steps only, timed after the cycle'}
        if a.math == 'fp32' and world == 1 and a.optin_steps > 0 and a.arch == 'vgg16' and a.task == 1 and a.batch == 256 and a.width_multiplier == 1.0:
            out['opt_in_conv_math'] = optin_modes(model, masks, pool, a.optin_steps, a.batch)
        if (not a.no_other_workloads and not a.no_cpu_baseline and world == 1 and a.arch == 'vgg16' and a.task == 1 and a.width_multiplier == 1.0
                and a.batch == 256 and a.math == 'fp32'):
            # (the full default line only: tooling runs -- profilers, A/B scripts -- pass --no-cpu-baseline and get the headline cycle alone)
            # free this process's cached blocks first: the children allocate their own pools on the same GPU
            torch.cuda.empty_cache()
            out['other_workloads'] = other_workloads(20, a.warmup)
            # ... and flat in `config`, for readers that keep only the scalar fields of the contract's objects
            for name, w_ in out['other_workloads'].items():
                ok = ('error' not in w_ and bool(w_.get('valid', True)) and bool((w_.get('parity_check') or {}).get('ok'))
                      and bool((w_.get('cycle_check') or {}).get('weights_finite')))
                out['config']['other_%s_images_per_sec' % name] = w_.get('value')
                out['config']['other_%s_whole_step_frac_of_dense_peak' % name] = (w_.get('whole_step') or {}).get('frac_of_dense_fp32_mfma_peak')
                out['config']['other_%s_parity_and_finite' % name] = ok
        if not a.no_cpu_baseline and world == 1:
            if a.task == 1 and a.width_multiplier == 1.0:
                out['cpu_baseline'] = cpu_baseline(steps=a.steps, batch=a.batch, validates=counts['validates'],
                                                   prune_events=counts['prune_events'], level=a.cpu_baseline, arch=a.arch)
            else:
                out['cpu_baseline'] = None      # (the task-2 / grown lines: the task-1 width-1.0 line of the same topology carries it)
    # The ONE JSON line must be the last thing on stdout.  RCCL writes its version banner through C stdio, which is block-buffered on
    # a pipe and would otherwise come out at process exit, behind the line: every rank empties its C buffers, then a barrier, then
    # rank 0 prints.
    _flush_c_stdio()
    if dist.is_initialized():
        dist.barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()
        _flush_c_stdio()


if __name__ == '__main__':
    main()
