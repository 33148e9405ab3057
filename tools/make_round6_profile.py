#!/usr/bin/env python3
"""gpurun_out/ of tools/run_round6_profile.sh (TAG) -> profiles/<out>_final.md, <out>_other_nets.md, <out>_task2.md, <out>_batch_split.md, <out>_grown.md,
the bench JSON lines, profiles/<out>_task_sequence.json and profiles/r06_traffic_bench.json.

    python tools/make_round6_profile.py r6a r06a
"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, out = sys.argv[1], sys.argv[2]
G = os.path.join(ROOT, 'gpurun_out')
P = os.path.join(ROOT, 'profiles')


def last_json(path):
    with open(path) as f:
        lines = [ln for ln in f.read().splitlines() if ln.startswith('{')]
    d = json.loads(lines[-1])
    if d.get('valid') is False or d.get('value') is None:
        sys.exit('%s: an INVALID line (non-finite weights / a broken invariant) does not go into a profile bundle' % path)
    return d


def read(path):
    with open(path) as f:
        return f.read()


def save(name, obj):
    with open(os.path.join(P, name), 'w') as f:
        json.dump(obj, f, indent=1)
        f.write('\n')


def g(name):
    return os.path.join(G, name % tag)


b220, b20 = last_json(g('bench_%s.log')), last_json(g('bench_%s_k20.log'))
t2, t2_220 = last_json(g('bench_%s_task2.log')), last_json(g('bench_%s_task2_k220.log'))
batches = {b: last_json(os.path.join(G, 'bench_%s_b%d.log' % (tag, b))) for b in (128, 64, 32)}
others = {a: last_json(os.path.join(G, 'bench_%s_%s.log' % (tag, a))) for a in ('resnet50', 'spherenet20')}
under = {a: last_json(os.path.join(G, 'prof_%s_%s.log' % (tag, a))) for a in ('vgg16', 'resnet50', 'spherenet20', 'task2', 'b32')}
save('%s_bench.json' % out, b220)
save('%s_bench_k20.json' % out, b20)
save('%s_bench_task2.json' % out, t2)
save('%s_bench_task2_k220.json' % out, t2_220)
for b, d in batches.items():
    save('%s_bench_batch%d.json' % (out, b), d)
for a, d in others.items():
    save('%s_bench_%s.json' % (out, a), d)
for a, d in under.items():
    save('%s_bench_under_rocprof_%s.json' % (out, a), d)
traffic_src = g('traffic_%s.json')
if os.path.exists(traffic_src) and os.path.getsize(traffic_src) > 100:
    shutil.copy(traffic_src, os.path.join(P, 'r06_traffic_bench.json'))
pytest_line = [ln for ln in read(g('pytest_%s.log')).splitlines() if ' passed' in ln or ' failed' in ln][-1].strip()
smoke_line = read(g('smoke_%s.txt')).strip()
net = read(g('net_%s.txt')).strip().splitlines()


def fam_table(d):
    rows = ['| family | launches | avg launch ms | algorithmic TFLOP/s | executed on the MFMA pipe | of the 157.3 peak |', '|---|---:|---:|---:|---:|---:|']
    for k, v in d['kernel_families'].items():
        rows.append('| %s | %d | %.3f | %.1f | %.1f | %.2f |' % (k, v['launches'], v['ms'] / v['launches'], v['tflops'], v['mfma_tflops_executed'],
                                                               v['frac_of_dense_peak_executed']))
    return '\n'.join(rows)


def traffic_table():
    try:
        doc = json.load(open(os.path.join(P, 'r06_traffic_bench.json')))
    except (OSError, ValueError):
        return ''
    benches = {'vgg16': b20, 'resnet50': others['resnet50'], 'spherenet20': others['spherenet20']}
    rows = ['## HBM traffic of the bench\'s own launch mix (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --arch A --steps 20`; '
            '2 x FETCH_SIZE + WRITE_SIZE, per library call; profiles/r06_traffic_bench.json, kernel sources %s at commit %s)\n'
            % (doc.get('csrc_sha256', '?')[:12], doc.get('commit')),
            '| topology | family | launches counted | HBM MB / launch | algorithmic MB / launch | ratio | split-reduce / pack share of the bytes |',
            '|---|---|---:|---:|---:|---:|---:|']
    for a, fams in doc['archs'].items():
        kd = benches[a].get('kernel_families', {})
        for fam, v in fams['families'].items():
            alg = kd.get(fam, {}).get('algorithmic_bytes_per_launch')
            rows.append('| %s | %s | %d | %.0f | %s | %s | %.3f |' % (a, fam, v['launches'], v['hbm_bytes_per_launch_corrected'] / 1e6,
                                                                 '%.0f' % (alg / 1e6) if alg else '-',
                                                                 '%.2f' % (v['hbm_bytes_per_launch_corrected'] / alg) if alg else '', v['helper_kernels_share_of_bytes']))
    return '\n'.join(rows) + '\n\n'


def roof(d):
    r = d['roofline']
    return ('`roofline`: %s, achieved %.1f TFLOP/s executed on the MFMA pipe of the %.1f dense fp32 peak = **frac %.4f** (algorithmic, SURVEY 8d\'s '
            'flops: %.1f of a launch-mix ceiling of %.1f); whole timed region: %.1f TFLOP/s executed = %.3f of the dense peak'
            % (r['kernel'], r['achieved'], r['peak'], r['frac'], r['achieved_algorithmic'], r['launch_mix_ceiling'], d['whole_step']['mfma_tflops_executed'],
               d['whole_step']['frac_of_dense_fp32_mfma_peak']))


with open(os.path.join(P, '%s_final.md' % out), 'w') as f:
    w = f.write
    w('# Round 6, state "%s": the headline (VGG16, configs[1], task 1) on 1x MI355X\n\n' % tag)
    w('Commands (`tools/run_round6_profile.sh`, one gpurun call, TAG=%s; composed by `tools/make_round6_profile.py`):\n\n' % tag)
    w('    python -m pytest tests -m gpu -q                  -> %s\n' % pytest_line)
    w('    __graft_entry__.smoke()                           -> %s\n' % smoke_line)
    w('    python bench.py                                   -> profiles/%s_bench.json      (%.2f img/s, %.3f ms/step, K = 220: the full section-8d cycle, full CPU baseline)\n'
      % (out, b220['value'], b220['ms_per_step']))
    w('    python bench.py --gpus 1 --steps 20 --warmup 5    -> profiles/%s_bench_k20.json  (%.2f img/s, %.3f ms/step: the driver\'s command, %s)\n'
      % (out, b20['value'], b20['ms_per_step'], read(g('bench_%s_k20.wall')).strip()))
    w('    rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --optin-steps 0\n')
    w('                                                      -> tables below; its bench line: profiles/%s_bench_under_rocprof_vgg16.json (%.1f img/s)\n\n'
      % (out, under['vgg16']['value']))
    w('* %s\n' % roof(b20))
    w('* `parity_check` (after the timed region: the timed model vs the CPU oracle, 4 images, train- and eval-mode forward): `%s`; `cycle_check`: `%s`\n' % (json.dumps({k: v for k, v in (b20.get('parity_check') or {}).items() if k != 'oracle'}), json.dumps(b20.get('cycle_check'))))
    cb = b220.get('cpu_baseline') or {}
    w('* `cpu_baseline` (K = 220 run, section-8d configuration): %s\n\n' % json.dumps({k: v for k, v in cb.items() if k != 'sample'}))
    ow = b20.get('other_workloads') or {}
    if ow:
        w('## `other_workloads` of the driver\'s command: the other single-GPU workloads of BASELINE.json\'s configs, 20-step cycles in child processes after the headline\'s timed region\n\n')
        w('| workload | flags | img/s | ms/step | whole step / dense fp32 MFMA peak | dominant family: frac | parity_check | weights finite |\n|---|---|---:|---:|---:|---|---|---|\n')
        for k, v in ow.items():
            if 'error' in v and v.get('value') is None:
                w('| %s | %s | ERROR: %s | | | | | |\n' % (k, v.get('flags'), v['error']))
                continue
            w('| %s | `%s` | %.1f | %.3f | %.3f | %s: %.3f | %s (%.2g) | %s |\n' % (k, v['flags'], v['value'], v['ms_per_step'], v['whole_step']['frac_of_dense_fp32_mfma_peak'],
                                                                              v['roofline']['kernel'], v['roofline']['frac'], v['parity_check']['ok'], v['parity_check']['max_rel_logit_err'],
                                                                              v['cycle_check']['weights_finite']))
        w('\n')
    w('## bench.py, K = 20 (HIP events around the C-ABI launches of every 4th train step and of every validate; launch counts and ms are the estimates for the whole timed region)\n\n%s\n\n' % fam_table(b20))
    w('phases: `%s`\n\n' % json.dumps(b20['phases']))
    w(traffic_table())
    w('## rocprofv3 --kernel-trace --stats of the same command (25 train + 4 eval passes incl. warm-up)\n\n')
    w(read(g('summary_%s_vgg16.md')))
    w('\n')

with open(os.path.join(P, '%s_task2.md' % out), 'w') as f:
    w = f.write
    x = t2['task2']
    w('# Round 6, state "%s": the cycle of tasks >= 2 (19 of the 20 tasks of configs[1]) -- `bench.py --task 2`, VGG16-BN 224x224, batch 256, fp32\n\n' % tag)
    w('Owner masks of a finished task 1 (the 30 % smallest weights of every layer free and zero, handed to task 2 by make_finetuning_mask), a piggymask '
      '`full(0.01)` on all 15 masked layers, MaskedSGD on the weights + MaskedAdam(lr_mask 5e-4) on the piggymasks in the finetune phase, lr_mask 0 in the '
      'prune run, `shared_ratio` in every validate batch; the task-1 cycle of the same length runs first in the same process.\n\n')
    w('| run | task-2 ms/step | task-1 ms/step (same process) | ratio | finetune_again leg ms/step | images/s |\n|---|---:|---:|---:|---:|---:|\n')
    for name, d in (('K = 20', t2), ('K = 220', t2_220)):
        xx = d['task2']
        w('| %s | %.3f | %.3f | **%.4f** | %.3f | %.1f |\n' % (name, d['ms_per_step'], xx['task1_ms_per_step'], xx['task2_over_task1'], xx['finetune_again_ms_per_step'], d['value']))
    w('\nphases (K = 20): `%s`\n\n* %s\n\n%s\n\n' % (json.dumps(t2['phases']), roof(t2), fam_table(t2)))
    w('What the piggymask costs per train step (families above against the task-1 line of profiles/%s_final.md; kernels in the table below): the masked linear '
      'layers run `k_gemm` with the binariser in its operand loader and the gW / gPM epilogue instead of the plain pointwise GEMMs (features.45 reads 411 MB of '
      'piggymask beside 411 MB of weights in every pass); the weight-pack kernels read the piggymask; `k_split_reduce` writes gPM beside gW; `k_adam_route` '
      '(37 B per piggymask element) runs in the finetune phase only -- in the prune run every piggymask gradient is routed to zero and the Adam state is still '
      'exactly zero, so MaskedAdam leaves the zero gradient and skips the pass (bit-identical: m = v = 0, pm unchanged).\n\n' % out)
    w('## rocprofv3 --kernel-trace --stats of `bench.py --task 2 --steps 20 --warmup 5 --no-cpu-baseline --optin-steps 0` (the task-1 leg + the task-2 cycle + the finetune_again leg)\n\n')
    w(read(g('summary_%s_task2.md')))
    w('\n')

with open(os.path.join(P, '%s_batch_split.md' % out), 'w') as f:
    w = f.write
    w('# Round 6, state "%s": the per-GPU compute of the reference\'s own data-parallel split (batch NOT scaled with the GPUs: 256 / N images per GPU)\n\n' % tag)
    w('`bench.py --batch B --steps 20 --warmup 5` on 1x MI355X (CPG_cifar100_main_normal.py:112-114,199: batch 256 over 8 GPUs = 32 per GPU).  The validate of '
      'the cycle stays 2 batches of 100 per rank.\n\n')
    w('| per-GPU batch | = 256 over | cycle ms/step | images/s (1 GPU) | finetune / prune-window / recovery train ms/step | masked-kernel ms/step | whole step, executed MFMA flops / dense peak |\n|---:|---:|---:|---:|---|---:|---:|\n')
    allb = dict(batches)
    allb[256] = b20
    for b in (256, 128, 64, 32):
        d = allb[b]
        ph = d['phases']
        w('| %d | %d GPUs | %.3f | %.1f | %.2f / %.2f / %.2f | %.2f | %.3f |\n' % (b, 256 // b, d['ms_per_step'], d['value'], ph.get('finetune_train_ms_per_step', 0),
                                                                            ph.get('prune_window_train_ms_per_step', 0), ph.get('recovery_train_ms_per_step', 0),
                                                                            d['masked_kernel_ms_per_step'], d['whole_step']['frac_of_dense_fp32_mfma_peak']))
    w('\n## per layer and pass: batch 256 vs batch 32 (HIP events, ms per launch / algorithmic TFLOP/s)\n\n| launch | ms @256 | TF @256 | ms @32 | TF @32 | efficiency kept |\n|---|---:|---:|---:|---:|---:|\n')
    ka = last_json(g('bench_%s_b128.log'))  # (detail is on for the batch runs only)
    kd32 = batches[32].get('kernel_detail', {})
    kd128 = batches[128].get('kernel_detail', {})
    for k in sorted(kd32):
        if k in kd128:
            a, b = kd128[k], kd32[k]
            w('| %s (128 -> 32) | %.3f | %.1f | %.3f | %.1f | %.2f |\n' % (k, a['ms'] / a['n'], a['tflops'], b['ms'] / b['n'], b['tflops'], b['tflops'] / max(a['tflops'], 1e-9)))
    w('\n(the batch-256 line carries no per-layer detail in this bundle; the 128 column is within 2-5 %% of it, see %s_final.md)\n\n' % out)
    w('## rocprofv3 --kernel-trace --stats of `bench.py --batch 32 --steps 20 --warmup 5 --no-cpu-baseline --optin-steps 0`\n\n')
    w(read(g('summary_%s_b32.md')))
    w('\n')

with open(os.path.join(P, '%s_other_nets.md' % out), 'w') as f:
    w = f.write
    w('# Round 6, state "%s": ResNet-50 and SphereNet-20 (BASELINE configs[3] / configs[4] topologies) on 1x MI355X, batch 256, fp32\n\n' % tag)
    w('    python tools/net_bench.py --arch resnet50 | spherenet20 --steps 10   (forward + backward + SGD, no cycle around it)\n')
    for ln in net:
        w('        ' + ln + '\n')
    w('    python bench.py --arch resnet50 | spherenet20 --steps 20 --warmup 5  (the task-1 CPG cycle: finetune -> prune -> recovery, validate, statistics)\n')
    for a, d in others.items():
        w('        %-12s %.1f img/s, %.3f ms/step -> profiles/%s_bench_%s.json\n' % (a, d['value'], d['ms_per_step'], out, a))
    w('    python tools/generic_bench.py --iters 5  (per shape class through the C ABI, TFLOP/s algorithmic; ms)\n\n')
    w('```\n' + read(g('generic_%s.txt')).strip() + '\n```\n\n')
    for a, d in others.items():
        w('## %s\n\n* %s\n* train steps alone: %.1f TFLOP/s algorithmic (`algorithmic_tflops_train_steps`)\n\n' % (a, roof(d), d['algorithmic_tflops_train_steps']))
        w(fam_table(d) + '\n\nphases: `%s`\n\n' % json.dumps(d['phases']))
        w('rocprofv3 --kernel-trace --stats of `bench.py --arch %s --steps 20 --warmup 5 --no-cpu-baseline` (bench line under the profiler: %.1f img/s):\n\n' % (a, under[a]['value']))
        w(read(os.path.join(G, 'summary_%s_%s.md' % (tag, a))))
        w('\n')
grown = last_json(g('bench_%s_grown.log'))
grown_under = last_json(os.path.join(G, 'prof_%s_grown.log' % tag))
seq = last_json(g('bench_%s_seq3.log'))
save('%s_bench_grown.json' % out, grown)
save('%s_bench_under_rocprof_grown.json' % out, grown_under)
save('%s_task_sequence.json' % out, seq)
with open(os.path.join(P, '%s_grown.md' % out), 'w') as f:
    w = f.write
    w('# Round 6, state "%s": the GROWN network of configs[1] -- raw width multiplier 1.5 -> sqrt -> int(v x 1.2247) = 78 / 156 / 313 / 627 channels, 30723 -> 5016 -> 5016 FC\n\n' % tag)
    w('(experiment1/CPG_cifar100_scratch_mul_1.5.sh:90-94 adds 0.5 to the raw multiplier on exit code 2, CPG_cifar100_main_normal.py:115-116 takes the square root, '
      'models/vgg.py:124-154 builds int(v * m) channels: the state most of the 20 tasks run in.)  `bench.py --width-multiplier 1.5 --steps 20 --warmup 5`, batch 256, fp32, task-1 cycle; '
      'its own line, never the headline.\n\n')
    w('* %.2f img/s, %.3f ms per cycle step (width 1.0 in the same bundle: %.2f img/s, %.3f ms); masked kernels %.2f ms per train step\n' % (grown['value'], grown['ms_per_step'], b20['value'], b20['ms_per_step'], grown['masked_kernel_ms_per_step']))
    w('* %s\n' % roof(grown))
    w('* `parity_check` (the timed model against the CPU oracle built at the same width, after the cycle): `%s`; `cycle_check`: `%s`\n\n' % (json.dumps({k: v for k, v in (grown.get('parity_check') or {}).items() if k != 'oracle'}), json.dumps(grown.get('cycle_check'))))
    w(fam_table(grown) + '\n\nphases: `%s`\n\n' % json.dumps(grown['phases']))
    w('Every 3 x 3 layer runs the Winograd kernels (forward / input gradient: the last chunk of 4 input channels starts at C - 4 and overlaps its neighbour, `wg_chunk_base`; '
      'weight gradient: the last block of 32 channels starts at C - 32 / K - 32, `k0` / `c0` in `k_wgw`).  "Executed" counts the multiply-adds of the TRUE channel counts, so the padding of the last '
      '32- / 64-channel block is part of the loss: 78 outputs are 96 computed by the one-wave kernel (128 by `k_wg3`, which therefore hands 78- and 156-output layers to `k_wg1`: '
      '`wino_variant`), 78 x 78 weight-gradient blocks are 96 x 96.\n\n')
    w('## per layer and pass (`tools/conv_bench.py --width-multiplier 1.5 --iters 5`; TFLOP/s algorithmic = 2.25 x executed for the Winograd launches)\n\n```\n' + read(g('conv_bench_%s_grown.txt')).strip() + '\n```\n\n')
    w('## rocprofv3 --kernel-trace --stats of `bench.py --width-multiplier 1.5 --steps 20 --warmup 5 --no-cpu-baseline --optin-steps 0` (bench line under the profiler: %.1f img/s)\n\n' % grown_under['value'])
    w(read(os.path.join(G, 'summary_%s_grown.md' % tag)))
    w('\n## The 3-task sequence at full size through `CPGSession` (`bench.py --task-sequence 3`, profiles/%s_task_sequence.json): growth forced on task 3\n\n' % out)
    w('| task | train steps | wall ms per train step | raw / rooted width | masked weights | owner histogram | chosen ratio | shared_ratio | DP payload per step: prune / finetune mode vs dense |\n|---:|---:|---:|---|---:|---|---:|---:|---|\n')
    for t in seq['tasks']:
        pay = t['dp_payload_bytes_per_step']
        w('| %d | %d | %.2f | %g / %.4f | %d | `%s` | %s | %s | %.0f / %.0f MB vs %.0f (+ piggymasks: %.0f) |\n' % (
            t['task'], t['train_steps'], t['wall_ms_per_train_step'], t['width_multiplier_raw'], t['width_multiplier_rooted'], t['masked_weights'],
            json.dumps(t['owner_histogram']), t['chosen_ratio'], t['shared_ratio'], pay['prune_mode'] / 1e6, pay['finetune_mode'] / 1e6, pay['dense_weights'] / 1e6, pay['dense_with_piggymasks'] / 1e6))
    w('\n%.2f img/s over all %d train steps (wall time includes validates, snapshots, the ratio choice, and for task 3 a finetune at the old width followed by the rebuild at the new one).\n' % (seq['value'], seq['steps']))
shutil.copy(g('conv_bench_%s.txt'), os.path.join(P, '%s_conv_bench.txt' % out))

# ---- round 6: the sequences, the 220-step cycles of the other topologies, the launch-diet A/B
seqs = {'vgg16_6tasks': last_json(g('bench_%s_seq6.log')), 'resnet50': last_json(g('bench_%s_seq3_resnet50.log')), 'spherenet20': last_json(g('bench_%s_seq3_spherenet20.log'))}
save('%s_task_sequence6.json' % out, seqs['vgg16_6tasks'])
save('%s_task_sequence_resnet50.json' % out, seqs['resnet50'])
save('%s_task_sequence_spherenet20.json' % out, seqs['spherenet20'])
k220 = {a: last_json(os.path.join(G, 'bench_%s_%s_k220.log' % (tag, a))) for a in ('resnet50', 'spherenet20')}
for a, d in k220.items():
    save('%s_bench_%s_k220.json' % (out, a), d)
shutil.copy(g('ab_launch_diet_%s.txt'), os.path.join(P, '%s_ab_launch_diet.txt' % out))
with open(os.path.join(P, '%s_sequences.md' % out), 'w') as f:
    w = f.write
    w('# Round 6, state "%s": BASELINE configs[3] / configs[4] as multi-task SEQUENCES, and a 6-task sequence of configs[1], at full size through `CPGSession`\n\n' % tag)
    w('`bench.py --task-sequence T --arch A` (batch 256, fp32, 20-step epochs: task 1 of the ResNet-50 / SphereNet-20 sequences is the pretrained pass-through '
      '+ a 10-epoch prune run, every later task 1 finetune epoch with piggymasks + a 10-epoch prune run; VGG16: finetune + prune run + piggymask retrain, '
      'growth forced at task 4, prune runs to 0.5).  Every line asserts in the run that EVERY earlier task answers bit-identically after each later task '
      '(`earlier_tasks_bit_identical`) and that the weights are finite (`valid`).\n\n')
    for name, d in seqs.items():
        w('## %s: %.1f img/s over %d train steps (%.3f ms per step, everything run_task does included); valid: %s, bit-identity checks: %s\n\n'
          % (name, d['value'], d['steps'], d['ms_per_step'], d['valid'], json.dumps(d['earlier_tasks_bit_identical'])))
        w('| task | dataset | classes | pass-through | train steps | wall ms / train step | raw width | owner histogram | sparsity | shared_ratio | first rank-prune event: k / released / exact zeros owned before / released beyond k |\n|---:|---|---:|---|---:|---:|---:|---|---:|---:|---|\n')
        for t in d['tasks']:
            ev = t.get('first_rank_prune_event') or {}
            w('| %d | %s | %d | %s | %d | %.2f | %g | `%s` | %.4f | %s | %s / %s / %s / %s |\n' % (
                t['task'], t['dataset'], t['num_classes'], t['pass_through'], t['train_steps'], t['wall_ms_per_train_step'], t['width_multiplier_raw'],
                json.dumps(t['owner_histogram']), t['sparsity'], t['shared_ratio'], ev.get('k_total'), ev.get('released_total'),
                ev.get('owned_slots_exactly_zero_before'), ev.get('released_beyond_k')))
        w('\n')
    w('A sparsity above the prune run\'s target is the reference\'s own rule at work: when more of a task\'s slots are EXACTLY zero than the rank k asks for '
      '(slots the task claimed whose gradient never left zero: dead input features of the synthetic images in the 25088-wide linear layer), the k-th smallest '
      '|w| is 0 and `abs(w) <= cutoff` (utils/prune.py:45) releases all of them; tests/test_sequence_gpu.py holds such an event bit-equal to the oracle at full size.\n\n')
    w('## 220-step cycles of the other two topologies (`bench.py --arch A`, K = 220)\n\n')
    for a, d in k220.items():
        w('* %s: %.1f img/s, %.3f ms/step; %s; cycle_check `%s`\n' % (a, d['value'], d['ms_per_step'], roof(d), json.dumps(d['cycle_check'])))
    w('\n## The launch diet of ABI 3, interleaved on one box (`%s_ab_launch_diet.txt`): default | CPG_PACK_CACHE=0 | CPG_MULTI_TENSOR=0 | both off, K = 40, three rounds\n\n```\n%s\n```\n'
      % (out, read(g('ab_launch_diet_%s.txt')).strip()))
print('wrote profiles/%s_{final,task2,batch_split,other_nets,grown,sequences}.md' % out)
