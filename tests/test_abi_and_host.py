"""CPU-side checks: the C-ABI library loads and exports every symbol include/cpg_hip.h declares,
host-side logic (schedule, gate, topology, init parity) matches the golden fixtures, and the
product path refuses to run without a HIP device (no CPU fallback)."""
import json
import os
import re
import types

import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import GOLDEN, ROOT, load_golden

import cpg_amd._lib as L
import cpg_amd.models as M
from cpg_amd.models import layers as nl
from cpg_amd.utils.prune import SparsePruner

VGG_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512, 'M']


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, 'include', 'cpg_hip.h')).read()
    declared = set(re.findall(r'\b(cpg_[a-z0-9_]+)\s*\(', header))
    declared -= {'cpg_conv_desc', 'cpg_prune_result'}
    assert declared == set(L.EXPORTS), declared ^ set(L.EXPORTS)
    lib = L.lib()                      # raises if the .so is missing or a symbol is absent
    assert lib.cpg_version() == L.ABI_VERSION == 3
    assert lib.cpg_rank_prune_workspace_bytes() > 0


def test_library_exports_nothing_but_the_header():
    """The export list of the .so IS include/cpg_hip.h: cross-file helpers (cpg_conv3x3_wino_run, ...) are linked hidden
    (cpg_amd/build.py::export_map; CPG_EXPORT_ALL=1 is the tools' escape hatch, not the shipped build)."""
    import shutil
    import subprocess
    nm = shutil.which('nm') or '/opt/rocm/lib/llvm/bin/llvm-nm'
    if not os.path.exists(nm) or os.environ.get('CPG_EXPORT_ALL') == '1' or os.environ.get('CPG_HIP_LIB'):
        pytest.skip('no nm / a tools build of the library')
    out = subprocess.check_output([nm, '-D', '--defined-only', L.LIB_PATH]).decode()
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert exported == set(L.EXPORTS), exported ^ set(L.EXPORTS)


def test_struct_layouts_match_header():
    import ctypes
    assert ctypes.sizeof(L.ConvDesc) == 14 * 4
    assert ctypes.sizeof(L.PruneResult) == 32
    assert L.PruneResult.cutoff.offset == 24 and L.PruneResult.status.offset == 28


def test_no_cpu_fallback():
    conv = nl.SharableConv2d(3, 4, 3, padding=1, bias=False)
    nn.init.normal_(conv.weight)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        conv(torch.zeros(1, 3, 8, 8))
    lin = nl.SharableLinear(4, 3)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        lin(torch.zeros(2, 4))
    with pytest.raises(NotImplementedError):
        nl.SharableLinear(4, 3, threshold_fn='ternarizer')


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'cpg_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, re.M), os.path.join(dirpath, f)


def build(arch, width, ncls=5):
    torch.manual_seed(1)
    kw = dict(dataset_history=[], dataset2num_classes={}, network_width_multiplier=width, shared_layer_info={})
    m = {'vgg_cifar100': lambda: M.custom_vgg_cifar100(VGG_CFG, **kw), 'vgg': lambda: M.custom_vgg(VGG_CFG, **kw),
         'resnet50': lambda: M.resnet50(**kw), 'spherenet20': lambda: M.spherenet20(**kw)}[arch]()
    m.add_dataset('t1', ncls)
    m.set_dataset('t1')
    return m


class _Wrap(nn.Module):
    def __init__(self, m):
        super().__init__()
        self.module = m


@pytest.mark.parametrize('arch', ['vgg_cifar100', 'vgg', 'resnet50', 'spherenet20'])
def test_topology_names_and_shapes_match_reference(arch):
    topo = json.load(open(os.path.join(GOLDEN, 'topology.json')))[arch]
    m = _Wrap(build(arch, 1.0))
    layers = []
    for name, mod in m.named_modules():
        if isinstance(mod, nl.SharableConv2d):
            layers.append([name, 'conv', list(mod.weight.shape), list(mod.stride), list(mod.padding), mod.bias is not None])
        elif isinstance(mod, nl.SharableLinear):
            layers.append([name, 'linear', list(mod.weight.shape), None, None, mod.bias is not None])
    assert layers == topo['masked_layers']
    assert [[n, list(p.shape)] for n, p in m.named_parameters()] == topo['param_names']
    assert sum(p.numel() for p in m.parameters()) == topo['n_params']


@pytest.mark.parametrize('arch,width,fx', [('vgg_cifar100', 0.125, 'first_forward_vgg_cifar100'), ('vgg', 0.125, 'first_forward_vgg'),
                                           ('resnet50', 0.25, 'first_forward_resnet50'), ('spherenet20', 0.25, 'first_forward_spherenet20')])
def test_seeded_init_matches_reference(arch, width, fx):
    g = load_golden(fx)
    m = build(arch, width, int(g['num_classes']))
    dig = np.array([[float(p.detach().double().sum()), float(p.detach().double().abs().sum())] for p in m.parameters()])
    np.testing.assert_allclose(dig, g['param_digest'], rtol=1e-12, atol=1e-12)
    np.testing.assert_array_equal(next(iter(m.parameters())).detach().reshape(-1)[:64].numpy(), g['first_param_head'])


def test_piggymask_assignment_registers_parameter():
    conv = nl.SharableConv2d(3, 4, 3, padding=1, bias=False)
    assert conv.piggymask is None and 'piggymask' not in dict(conv.named_parameters())
    conv.piggymask = nn.Parameter(torch.full(conv.weight.shape, 0.01))
    assert 'piggymask' in dict(conv.named_parameters()) and 'piggymask' in conv.state_dict()
    assert conv.info['threshold'] == 5e-3


def _pruner(mode, begin, end, freq, init, target):
    net = build('vgg_cifar100', 0.125)
    model = _Wrap(net)
    args = types.SimpleNamespace(mode=mode, dataset='t1', finetune_again=False, target_sparsity=target, initial_sparsity=init,
                                 pruning_frequency=freq, weight_decay=4e-5, network_width_multiplier=0.125)
    return SparsePruner(model, {}, args, begin, end, 1)


def test_schedule_and_gate_match_reference():
    tab = load_golden('schedule')['table']
    pruners = {}
    for begin, end, freq, init, target, step, upd, ratio in tab:
        key = (begin, end, freq, init, target)
        if key not in pruners:
            pruners[key] = _pruner('prune', int(begin), int(end), int(freq), init, target)
        p = pruners[key]
        got = p._time_to_update_masks(int(step))
        assert int(got) == int(upd)
        if got:
            p.last_prune_step = int(step)
        assert p._adjust_sparsity(int(step)) == ratio


def test_current_dataset_idx_rules():
    assert _pruner('prune', 0, 8, 3, 0.0, 0.1).current_dataset_idx == 1
    assert _pruner('finetune', 0, 8, 3, 0.0, 0.1).current_dataset_idx == 0      # +1 in make_finetuning_mask
    with pytest.raises(SystemExit):
        _pruner('bogus', 0, 8, 3, 0.0, 0.1)


def test_angle_head_matches_reference():
    """AngleLinear / AngleLoss (config 5 head, stock torch ops on any device) against the reference's outputs."""
    from cpg_amd.models.spherenet import AngleLinear, AngleLoss
    g = load_golden('angle_head')
    lin = AngleLinear(16, 10)
    lin.weight.data.copy_(torch.from_numpy(g['w']))
    x = torch.from_numpy(g['x']).requires_grad_(True)
    t = torch.from_numpy(g['t'])
    cos, phi = lin(x)
    np.testing.assert_allclose(cos.detach().numpy(), g['cos'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(phi.detach().numpy(), g['phi'], rtol=1e-5, atol=1e-5)
    crit = AngleLoss()
    for k in range(3):
        lin.zero_grad()
        loss = crit(lin(x), t)
        loss.backward()
        assert abs(float(loss) - g['losses'][k]) < 1e-5
        np.testing.assert_allclose(lin.weight.grad.numpy(), g['gw'][k], rtol=1e-4, atol=1e-6)
    assert crit.it == 3


def _two_task_vgg(width):
    torch.manual_seed(5)
    kw = dict(dataset_history=[], dataset2num_classes={}, network_width_multiplier=width, shared_layer_info={})
    net = M.custom_vgg_cifar100(VGG_CFG, **kw)
    net.add_dataset('t1', 5)
    net.add_dataset('t2', 5)
    net.set_dataset('t2')
    return net


def test_checkpoint_written_by_the_reference_loads(tmp_path):
    """tests/golden/reference_checkpoint-7.pth.tar was written by the reference's Manager.save_checkpoint; our loader
    restores every tensor, and our writer produces the same dictionary layout."""
    from cpg_amd.utils import checkpoint as ckpt
    ref = torch.load(os.path.join(GOLDEN, 'reference_checkpoint-7.pth.tar'), map_location='cpu', weights_only=False)
    assert set(ref) == {'model_state_dict', 'dataset_history', 'dataset2num_classes', 'masks', 'shared_layer_info'}
    net = _two_task_vgg(0.0625)
    model = _Wrap(net)
    ckpt.load_state(model, ref['model_state_dict'], for_evaluate=False)
    sd = net.state_dict()
    for k, v in ref['model_state_dict'].items():
        if 'piggymask' in k or k.startswith('classifier.'):
            continue
        assert torch.equal(sd[k], v), k
    # the per-task layers re-attach (inference path) and become the module's own tensors
    info = ref['shared_layer_info']
    ckpt.load_state(model, ref['model_state_dict'], for_evaluate=True)
    ckpt.attach_task_layers(model, info, 't2')
    assert torch.equal(net.features[1].running_mean, info['t2']['bn_layer_running_mean']['features.1'])
    # our writer: same keys at every level, same tensors
    for name, mod in net.named_modules():
        if isinstance(mod, (nl.SharableConv2d, nl.SharableLinear)):
            mod.piggymask = info['t2']['piggymask'][name]
    path = str(tmp_path / 'ours.pth.tar')
    ckpt.save_checkpoint(model, ref['masks'], {'t2': {'network_width_multiplier': 0.0625}}, 't2', path)
    ours = torch.load(path, map_location='cpu', weights_only=False)
    assert set(ours) == set(ref) and ours['dataset_history'] == ref['dataset_history']
    assert set(ours['model_state_dict']) == set(ref['model_state_dict'])
    for k in ref['model_state_dict']:
        if not k.startswith('classifier'):
            assert torch.equal(ours['model_state_dict'][k], ref['model_state_dict'][k]), k
    for key, table in ref['shared_layer_info']['t2'].items():
        if isinstance(table, dict):
            assert set(ours['shared_layer_info']['t2'][key]) == set(table), key
    assert set(ours['masks']) == set(ref['masks'])


def test_checkpoint_growth_and_crop():
    """resume into a WIDER network copies into the top-left corner (utils/manager.py:247-257); inference into a
    NARROWER one crops (utils/manager.py:286-297)."""
    from cpg_amd.utils import checkpoint as ckpt
    small, big = _two_task_vgg(0.0625), _two_task_vgg(0.125)
    ssd = {k: v.clone() for k, v in small.state_dict().items()}
    before = {k: v.clone() for k, v in big.state_dict().items()}
    ckpt.load_state(_Wrap(big), ssd, for_evaluate=False)
    w_s, w_b = ssd['features.3.weight'], big.state_dict()['features.3.weight']
    assert torch.equal(w_b[:w_s.shape[0], :w_s.shape[1]], w_s)
    assert torch.equal(w_b[w_s.shape[0]:], before['features.3.weight'][w_s.shape[0]:])
    # (heads are rebuilt at each task's own width by _reconstruct_classifiers, so they never need cropping)
    bsd = {k: v.clone() for k, v in big.state_dict().items() if not k.startswith('classifiers.')}
    ckpt.load_state(_Wrap(small), bsd, for_evaluate=True)
    assert torch.equal(small.state_dict()['features.45.weight'], bsd['features.45.weight'][:256, :32])


def test_task_layer_snapshots_are_copies_and_attach_copies_back():
    """shared_layer_info[task] must hold COPIES (one process trains the same BatchNorm tensors for the next task), with
    Parameter-ness kept so a file we write still re-attaches in the reference; attach_task_layers copies values into the
    live tensors (identity of the live Parameters unchanged) and restores / clears piggymasks."""
    from cpg_amd.utils import checkpoint as ckpt
    from torch.nn.parameter import Parameter
    net = _two_task_vgg(0.0625)
    model = _Wrap(net)
    bn = net.features[1]
    bn.running_mean.fill_(0.25)
    bn.weight.data.fill_(1.5)
    info = {}
    ckpt.collect_task_layers(model, info, 't1')
    assert info['t1']['piggymask'] == {}
    snap_w = info['t1']['bn_layer_weight']['features.1']
    assert isinstance(snap_w, Parameter) and snap_w is not bn.weight and snap_w.data_ptr() != bn.weight.data_ptr()
    # "task 2" trains on: new BN values and piggymasks
    bn.running_mean.fill_(-3.0)
    bn.weight.data.fill_(7.0)
    for _, m in net.named_modules():
        if isinstance(m, (nl.SharableConv2d, nl.SharableLinear)):
            m.piggymask = Parameter(torch.full_like(m.weight.detach(), 0.25))
    ckpt.collect_task_layers(model, info, 't2')
    assert float(info['t1']['bn_layer_running_mean']['features.1'][0]) == 0.25          # task 1's snapshot did not move
    assert float(info['t2']['bn_layer_running_mean']['features.1'][0]) == -3.0
    live_w = bn.weight
    ckpt.attach_task_layers(model, info, 't1', piggymasks=True)
    assert bn.weight is live_w and float(bn.weight.detach()[0]) == 1.5 and float(bn.running_mean[0]) == 0.25
    assert net.features[0].piggymask is None
    ckpt.attach_task_layers(model, info, 't2', piggymasks=True)
    assert float(bn.weight.detach()[0]) == 7.0 and float(net.features[0].piggymask.detach().view(-1)[0]) == 0.25
    net.features[0].piggymask.data.fill_(9.0)                                            # training the live copy ...
    assert float(info["t2"]["piggymask"]["features.0"].detach().view(-1)[0]) == 0.25       # ... does not reach the snapshot


def test_resize_masks_pad_on_growth_and_crop_for_inference():
    """CPG_cifar100_main_normal.py:208-249: widen -> zero (free) masks with the old mask in the top-left corner; evaluate an
    older, narrower task -> crop; anything else is refused."""
    from cpg_amd.utils import checkpoint as ckpt
    small, big = _Wrap(_two_task_vgg(0.0625)), _Wrap(_two_task_vgg(0.125))
    g = torch.Generator().manual_seed(3)
    masks = {n: torch.randint(1, 3, m.weight.shape, generator=g, dtype=torch.uint8) for n, m in small.named_modules()
             if isinstance(m, (nl.SharableConv2d, nl.SharableLinear))}
    old = {k: v.clone() for k, v in masks.items()}
    assert ckpt.resize_masks(small, masks, 'finetune') is False                           # same width: untouched
    with pytest.raises(AssertionError):
        ckpt.resize_masks(big, dict(masks), 'prune')
    with pytest.raises(AssertionError):
        ckpt.resize_masks(big, dict(masks), 'inference')
    assert ckpt.resize_masks(big, masks, 'finetune') is True
    for n, m in big.named_modules():
        if isinstance(m, (nl.SharableConv2d, nl.SharableLinear)):
            o = old[n]
            assert masks[n].shape == m.weight.shape and masks[n].dtype == torch.uint8
            corner = masks[n][tuple(slice(0, s) for s in o.shape)]
            assert torch.equal(corner, o)
            assert int(masks[n].sum()) == int(o.sum())                                    # everything outside the corner is free (0)
    assert ckpt.resize_masks(small, masks, 'inference') is True
    for n in old:
        assert torch.equal(masks[n], old[n])


def _growth_session_vs_reference(device):
    """CPGSession.grow against the REFERENCE's growth path (tests/golden/make_golden.py::gen_growth ran bash's `raw + step`,
    main()'s square root, the seeded wider model, the inline mask padding and Manager.load_checkpoint's top-left copy:
    experiment1/CPG_cifar100_scratch_mul_1.5.sh:90-94, CPG_cifar100_main_normal.py:115,135,155-232, utils/manager.py:233-264).
    The session rebuilds the same task-1 state from the recorded seeds (its seeded initialisation is bit-identical to the
    reference's: test_seeded_init_matches_reference), grows by the same RAW step and must land on the same channel counts,
    the same bits in every tensor of the grown model and the same padded owner masks."""
    import math
    import zlib
    from cpg_amd.driver import CPGSession

    def crc(t):
        return zlib.crc32(t.detach().cpu().contiguous().numpy().tobytes()) & 0xFFFFFFFF
    g = load_golden('growth')
    raw0, step = float(g['raw0']), float(g['step'])
    sess = CPGSession('custom_vgg_cifar100', width_multiplier=raw0, device=device, seed=int(g['seed']))
    assert sess.width == float(g['width0']) == math.sqrt(raw0)
    sess.start_task('t1', 5)
    gen = torch.Generator().manual_seed(int(g['mask_seed']))
    for name, mod in sess.model.named_modules():
        if isinstance(mod, (nl.SharableConv2d, nl.SharableLinear)):
            sess.masks[name].copy_(torch.randint(0, 2, mod.weight.shape, generator=gen, dtype=torch.uint8))
        elif isinstance(mod, nn.BatchNorm2d):
            mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=gen) * 0.1)
            mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=gen) + 0.5)
    sess.commit_task('t1')
    snap = sess.snapshot()
    sess.grow(sess.width_multiplier + step, snap)              # bash: network_width_multiplier + step; main(): sqrt
    sess.start_task('t2', 5)
    assert sess.width_multiplier == float(g['raw1']) and sess.width == float(g['width1']) == math.sqrt(raw0 + step)
    assert sess.shared_layer_info['t1']['network_width_multiplier'] == float(g['width0'])
    assert sess.shared_layer_info['t2']['network_width_multiplier'] == float(g['width1'])        # the ROOTED value (:291)
    sd = sess.net.state_dict()
    names = [str(n) for n in g['names']]
    ours = [k for k in sd if not k.endswith('piggymask')]       # (task 2's piggymasks are created by start_task, :251-270)
    assert ours == names
    # the ragged channel counts of the rooted multiplier: int(v * sqrt(2 / 64)) = 11 / 22 / 45 / 90
    assert [sd['features.%d.weight' % i].shape[0] for i in (0, 7, 14, 24)] == [11, 22, 45, 90]
    for n, shape, c in zip(names, g['shapes'], g['crc']):
        assert list(sd[n].shape) == [int(v) for v in shape[:sd[n].dim()]], n
        if 't/' + n in g.files:
            assert np.array_equal(sd[n].detach().cpu().numpy(), g['t/' + n]), n
        assert crc(sd[n]) == int(c), 'grown tensor %s differs from the reference' % n
    assert sorted(sess.masks) == [str(n) for n in g['mask_names']]
    for k in sess.masks:
        assert sess.masks[k].dtype == torch.uint8 and np.array_equal(sess.masks[k].cpu().numpy(), g['m/' + k]), k
    for _, m in sess.net.named_modules():
        if isinstance(m, (nl.SharableConv2d, nl.SharableLinear)):
            assert m.piggymask is not None and m.piggymask.shape == m.weight.shape and bool((m.piggymask == 0.01).all())


def test_growth_matches_reference_host():
    _growth_session_vs_reference('cpu')                         # (construction, copies and mask padding need no kernel)


@pytest.mark.gpu
def test_growth_matches_reference_gpu():
    _growth_session_vs_reference('cuda:0')


@pytest.mark.parametrize('arch', ['resnet50', 'spherenet20'])
def test_growth_of_the_other_topologies_matches_reference_host(arch):
    """CPGSession.grow on ResNet-50 (the int() placement of models/resnet.py:68-75,115,173: Bottleneck widths, expansion x 4, downsample
    convs) and SphereNet-20 (biased convs, PReLU slopes) against the reference's own growth path (make_golden.py::gen_growth_other_nets):
    same names in the same order, same shapes, every tensor of the grown model crc-equal, owner masks zero-padded bit for bit."""
    import math
    import zlib
    from cpg_amd.driver import CPGSession

    def crc(t):
        return zlib.crc32(t.detach().cpu().contiguous().numpy().tobytes()) & 0xFFFFFFFF
    g = load_golden('growth_' + arch)
    raw0, step = float(g['raw0']), float(g['step'])
    first, second = str(g['first']), str(g['second'])
    sess = CPGSession(arch, width_multiplier=raw0, device='cpu', seed=int(g['seed']))
    sess.start_task(first, 6)
    gen = torch.Generator().manual_seed(int(g['mask_seed']))
    for name, mod in sess.model.named_modules():
        if isinstance(mod, (nl.SharableConv2d, nl.SharableLinear)):
            sess.masks[name].copy_(torch.randint(0, 2, mod.weight.shape, generator=gen, dtype=torch.uint8))
        elif isinstance(mod, nn.BatchNorm2d):
            mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=gen) * 0.1)
            mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=gen) + 0.5)
        elif isinstance(mod, nn.PReLU):
            mod.weight.data.copy_(torch.rand(mod.weight.shape, generator=gen) * 0.5)
    sess.commit_task(first)
    snap = sess.snapshot()
    sess.grow(sess.width_multiplier + step, snap)
    sess.start_task(second, 5)
    assert sess.width == float(g['width1']) == math.sqrt(raw0 + step)
    sd = sess.net.state_dict()
    ours = [k for k in sd if not k.endswith('piggymask')]
    assert ours == [str(n) for n in g['names']]
    for n, shape, c in zip(ours, g['shapes'], g['crc']):
        assert list(sd[n].shape) == [int(v) for v in shape[:sd[n].dim()]], n
        if 't/' + n in g.files:
            assert np.array_equal(sd[n].detach().cpu().numpy(), g['t/' + n]), n
        assert crc(sd[n]) == int(c), 'grown tensor %s differs from the reference' % n
    assert sorted(sess.masks) == [str(n) for n in g['mask_names']]
    for n, c, shape in zip(sorted(sess.masks), g['mask_crc'], g['mask_shapes']):
        assert list(sess.masks[n].shape) == [int(v) for v in shape] and crc(sess.masks[n]) == int(c), n


def test_growth_from_raw_1_to_1p5_gives_the_reference_channel_counts():
    """experiment1's one growth step: raw 1.0 + 0.5 -> sqrt(1.5) = 1.2247 -> int(v * 1.2247) = 78 / 156 / 313 / 627 channels and
    5016-wide FC layers (models/vgg.py:104-121) -- not 1.5 x the channels (96 / 192 / 384 / 768), which is what an additive step in
    model space would build."""
    from cpg_amd.driver import CPGSession
    sess = CPGSession('custom_vgg_cifar100', width_multiplier=1.0, device='cpu', seed=1)
    assert sess.width == 1.0
    sess.grow(sess.width_multiplier + 0.5)
    assert sess.width_multiplier == 1.5 and sess.width == 1.5 ** 0.5
    sd = sess.net.state_dict()
    assert [sd['features.%d.weight' % i].shape[0] for i in (0, 3, 7, 14, 24, 40)] == [78, 78, 156, 313, 627, 627]
    assert tuple(sd['features.45.weight'].shape) == (5016, 627) and tuple(sd['features.47.weight'].shape) == (5016, 5016)


def test_choose_ratio_table():
    """The selection rule itself (host logic): walk the record from the sparsest ratio down."""
    from cpg_amd.driver import choose_ratio
    rec = {0.0: 0.80, 0.1: 0.81, 0.2: 0.79, 0.3: 0.70}
    assert choose_ratio(rec, 0.78, 0.0, False) == 0.2
    assert choose_ratio(rec, 0.78, 0.09, False) == 0.3          # --allow_acc_loss
    assert choose_ratio(rec, 0.95, 0.0, False) == 0.0           # nothing holds the goal: the pre-prune checkpoint
    assert choose_ratio(rec, 0.95, 0.0, True) == 0.3            # at the width cap with a missed goal: the sparsest anyway
    assert choose_ratio({0.0: 0.9}, 0.5, 0.0, False) == 0.0     # the sweep recorded nothing


def test_settle_host_gc_freezes_what_is_alive():
    """cpg_amd.utils.settle_host_gc (called by CPGSession once a task's model stands, by bench.py after its warm-up): everything alive
    moves to the collector's permanent generation, and a later call first gives cyclic garbage among the frozen objects back."""
    import gc
    import weakref
    from cpg_amd.utils import settle_host_gc

    class Node(object):
        pass
    try:
        a, b = Node(), Node()
        a.other, b.other = b, a                                  # a cycle that only the collector can free
        wa = weakref.ref(a)
        settle_host_gc()
        assert gc.get_freeze_count() > 1000                      # the interpreter, torch, the test modules ...
        del a, b
        gc.collect()
        assert wa() is not None                                  # frozen: a plain collection does not see the dead cycle
        settle_host_gc()                                         # unfreeze + collect + freeze again
        assert wa() is None
    finally:
        gc.unfreeze()


def test_grouped_conv_constructs_and_has_no_cpu_fallback():
    """groups > 1 (models/layers.py:108-109 forwards it to F.conv2d): the layer and the resnext factories construct with the reference's
    parameter shapes; the forward runs one groups == 1 launch per group on the HIP path -- and, like everything else, refuses CPU tensors."""
    conv = nl.SharableConv2d(8, 12, 3, padding=1, groups=2)
    assert tuple(conv.weight.shape) == (12, 4, 3, 3) and conv.groups == 2
    nn.init.normal_(conv.weight)
    nn.init.zeros_(conv.bias)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        conv(torch.zeros(1, 8, 6, 6))
    with pytest.raises(RuntimeError, match='does not match weight'):
        conv(torch.zeros(1, 6, 6, 6))
    net = M.resnext50_32x4d(dataset_history=[], dataset2num_classes={}, network_width_multiplier=1.0, shared_layer_info={})
    assert net.layer1[0].conv2.groups == 4 and tuple(net.layer1[0].conv2.weight.shape) == (128, 32, 3, 3)


@pytest.mark.parametrize('src,names,agprs', [('conv3x3_wino.hip', r'k_wg[123]I', (128, 256)), ('conv3x3_wino_wgrad.hip', r'k_wgwI', (256,))])
def test_winograd_kernels_keep_their_accumulators_to_themselves(src, names, agprs, tmp_path):
    """The Winograd kernels hold their accumulators in accumulation registers that only the inline asm names (clobber lists), so the
    register allocator believes those registers are free between the asm statements: if a kernel runs out of vector registers it
    parks values there and silently overwrites an accumulator (it did, once, in an epilogue).  Checked on the compiled code: every
    v_accvgpr_write is one of the zero-fills, there are as many zero-fills as accumulators, and nothing spills to scratch."""
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc')
    out = tmp_path / 'k.s'
    subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fno-gpu-rdc', '-x', 'hip', '-S', '--cuda-device-only',
                    os.path.join(ROOT, 'cpg_amd', 'csrc', src), '-o', str(out)], check=True, capture_output=True, timeout=600)
    txt = out.read_text()
    found = 0
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end', txt, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if not re.search(names, name):
            continue
        found += 1
        zero = len(re.findall(r'v_accvgpr_write_b32 a\d+, 0\b', body))
        other = len(re.findall(r'v_accvgpr_write', body)) - zero
        assert zero in agprs and other == 0, (name, zero, other)
        assert 'scratch_' not in body, name
        assert len(re.findall(r'v_accvgpr_read', body)) % zero == 0, name
    assert found >= 2


def test_library_options_table_on_the_host(monkeypatch):
    """The option table of the C ABI (cpg_set_option / cpg_get_option) needs no GPU: values, unset, the Winograd-kernel spellings, unknown
    names, and that a switch changes the library's own dispatch answer (cpg_conv2d_winograd) while an environment variable set AFTER
    loading does not."""
    import ctypes
    lib = L.lib()
    d = L.ConvDesc()
    d.N, d.C, d.H, d.W, d.K, d.R, d.S = 4, 64, 28, 28, 64, 3, 3
    d.stride_h = d.stride_w = d.pad_h = d.pad_w = d.dil_h = d.dil_w = d.groups = 1
    assert L.get_option('CPG_NO_WINO') in (None, 0)
    assert lib.cpg_conv2d_winograd(ctypes.byref(d), 0) == 1 and lib.cpg_conv2d_winograd(ctypes.byref(d), 2) == 1
    monkeypatch.setenv('CPG_NO_WINO', '1')
    assert lib.cpg_conv2d_winograd(ctypes.byref(d), 0) == 1               # the environment was read once, at load time
    with L.option('CPG_NO_WINO', 1):
        assert L.get_option('CPG_NO_WINO') == 1 and lib.cpg_conv2d_winograd(ctypes.byref(d), 0) == 0
        with L.option('CPG_NO_WINO', None):
            assert L.get_option('CPG_NO_WINO') is None and lib.cpg_conv2d_winograd(ctypes.byref(d), 0) == 1
        assert L.get_option('CPG_NO_WINO') == 1
    assert lib.cpg_conv2d_winograd(ctypes.byref(d), 0) == 1
    with L.option('CPG_NO_WINO_WGRAD', 1):
        assert lib.cpg_conv2d_winograd(ctypes.byref(d), 0) == 1 and lib.cpg_conv2d_winograd(ctypes.byref(d), 2) == 0
    for spelling, value in (('block', 0), ('wave', 1), ('pair', 2), ('64', 3)):
        with L.option('CPG_WINO_KERNEL', spelling):
            assert L.get_option('CPG_WINO_KERNEL') == value
    assert L.get_option('CPG_WINO_KERNEL') is None
    assert lib.cpg_set_option(b'CPG_NO_SUCH_SWITCH', 1) != 0 and b'unknown option' in lib.cpg_last_error()
    v = ctypes.c_int32(0)
    assert lib.cpg_get_option(b'CPG_NO_SUCH_SWITCH', ctypes.byref(v)) != 0
    # the shared-chip hint is process-wide
    import threading
    assert lib.cpg_get_shared_chip_hint() == 0
    lib.cpg_set_shared_chip_hint(1)
    seen = []
    t = threading.Thread(target=lambda: seen.append(lib.cpg_get_shared_chip_hint()))
    t.start(); t.join()
    lib.cpg_set_shared_chip_hint(0)
    assert seen == [1]


def test_bench_cycle_plan_and_scaling_table():
    """bench.py's host logic: the K-step cycle keeps its shape (A finetune steps, 4 prune events in a window of 4 f steps), the
    single-GPU time table behind the scaling prediction, --global-batch arithmetic."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.cycle_plan(220) == (20, 10) and b.cycle_plan(20) == (2, 1) and b.cycle_plan(1) == (1, 1)
    assert b.single_gpu_ms('vgg16', 32) < b.single_gpu_ms('vgg16', 64) < b.single_gpu_ms('vgg16', 256)
    assert abs(b.single_gpu_ms('resnet50', 128) - b.single_gpu_ms('resnet50', 256) / 2) < 1e-9      # (no measurement: scaled)
    pr = b.predict_step_ms('vgg16', 8, [('chunk', 1 << 26), ('tensor', 1 << 24), ('coalesced', 1 << 16)], batch=32)
    assert pr['single_gpu_ms_per_step'] == b.single_gpu_ms('vgg16', 32) and pr['predicted_ms_per_step'] > pr['single_gpu_ms_per_step']
    assert len(b.csrc_digest()) == 64



def test_bench_kernel_clock_weights_the_sampled_train_steps():
    """bench.py --clock-every N: events on some train steps only, on every validate; the summary weights a train-step record by
    train steps / clocked train steps, so family averages estimate the all-launch figures (launch counts included)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod2', os.path.join(ROOT, 'bench.py'))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)

    class Ev(object):
        def __init__(self, ms):
            self.ms = ms

        def elapsed_time(self, other):
            return other.ms

    c = b.KernelClock()
    for step in (2, 6):                                     # two of eight train steps clocked, 3 launches each, 2 ms each
        for _ in range(3):
            c.records.append(('conv_fwd 3x3', 10.0, Ev(0), Ev(2.0), 10.0 / 2.25, 100.0, step))
    c.records.append(('conv_fwd 3x3', 4.0, Ev(0), Ev(1.0), 4.0, 40.0, None))       # a validate launch: always clocked, weight 1
    agg = c.summary(train_steps=8)
    n, ms, fl, ex, nb = agg['conv_fwd 3x3']
    assert n == 25 and abs(ms - (24 * 2.0 + 1.0)) < 1e-9 and abs(fl - (24 * 10.0 + 4.0)) < 1e-9 and abs(nb - 2440.0) < 1e-9
    assert c.train_steps_clocked == 2 and abs(c.train_ms - 12.0) < 1e-9 and c.launches_clocked == 7
    assert b.KernelClock().summary(train_steps=8) == {}
