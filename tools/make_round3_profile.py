#!/usr/bin/env python3
"""gpurun_out/ of tools/run_round3_profile.sh (TAG) -> profiles/<tag>_final.md, profiles/<tag>_other_nets.md and the bench JSON lines.

    python tools/make_round3_profile.py r3a r03a
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, out = sys.argv[1], sys.argv[2]
G = os.path.join(ROOT, 'gpurun_out')
P = os.path.join(ROOT, 'profiles')


def last_json(path):
    with open(path) as f:
        lines = [ln for ln in f.read().splitlines() if ln.startswith('{')]
    return json.loads(lines[-1])


def read(path):
    with open(path) as f:
        return f.read()


def save(name, obj):
    with open(os.path.join(P, name), 'w') as f:
        json.dump(obj, f, indent=1)
        f.write('\n')


b220 = last_json(os.path.join(G, 'bench_%s.log' % tag))
b20 = last_json(os.path.join(G, 'bench_%s_k20.log' % tag))
others = {a: last_json(os.path.join(G, 'bench_%s_%s.log' % (tag, a))) for a in ('resnet50', 'spherenet20')}
under = {a: last_json(os.path.join(G, 'prof_%s_%s.log' % (tag, a))) for a in ('vgg16', 'resnet50', 'spherenet20')}
save('%s_bench.json' % out, b220)
save('%s_bench_k20.json' % out, b20)
for a, d in others.items():
    save('%s_bench_%s.json' % (out, a), d)
for a, d in under.items():
    save('%s_bench_under_rocprof_%s.json' % (out, a), d)
pytest_line = [ln for ln in read(os.path.join(G, 'pytest_%s.log' % tag)).splitlines() if ' passed' in ln or ' failed' in ln][-1].strip()
net = read(os.path.join(G, 'net_%s.txt' % tag)).strip().splitlines()


def fam_table(d):
    rows = ['| family | launches | avg launch ms | algorithmic TFLOP/s | executed on the MFMA pipe | of the 157.3 peak |', '|---|---:|---:|---:|---:|---:|']
    for k, v in d['kernel_families'].items():
        rows.append('| %s | %d | %.3f | %.1f | %.1f | %.2f |' % (k, v['launches'], v['ms'] / v['launches'], v['tflops'], v['mfma_tflops_executed'],
                                                               v['frac_of_dense_peak_executed']))
    return '\n'.join(rows)


def traffic_table():
    """HBM bytes per launch (PMC passes over the bench itself, profiles/r03_traffic_bench.json) beside the algorithmic bytes the bench counts."""
    try:
        t = json.load(open(os.path.join(P, 'r03_traffic_bench.json')))['archs']
    except (OSError, KeyError, ValueError):
        return ''
    benches = {'vgg16': b20, 'resnet50': others['resnet50'], 'spherenet20': others['spherenet20']}
    rows = ['## HBM traffic of the bench\'s own launch mix (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --arch A --steps 20`; '
            '2 x FETCH_SIZE + WRITE_SIZE, per library call; profiles/r03_traffic_bench.json)\n',
            '| topology | family | launches counted | HBM MB / launch | algorithmic MB / launch (two operands read once, the third written once) | ratio | split-reduce / pack share of the bytes |',
            '|---|---|---:|---:|---:|---:|---:|']
    for a, fams in t.items():
        kd = benches[a].get('kernel_families', {})
        for fam, v in fams['families'].items():
            alg = None
            r = benches[a]['roofline']
            if kd.get(fam, {}).get('algorithmic_bytes_per_launch'):
                alg = kd[fam]['algorithmic_bytes_per_launch']
            elif r['kernel'] == fam and r.get('algorithmic_bytes_per_launch'):
                alg = r['algorithmic_bytes_per_launch']
            rows.append('| %s | %s | %d | %.0f | %s | %s | %.3f |' % (a, fam, v['launches'], v['hbm_bytes_per_launch_corrected'] / 1e6,
                                                                 '%.0f' % (alg / 1e6) if alg else '(not in this bench line)',
                                                                 '%.2f' % (v['hbm_bytes_per_launch_corrected'] / alg) if alg else '', v['helper_kernels_share_of_bytes']))
    return '\n'.join(rows) + '\n\n'


def roof(d):
    r = d['roofline']
    return ('`roofline`: %s, achieved %.1f TFLOP/s algorithmic (%.1f executed), peak %.1f (launch-mix ceiling; dense %.1f), **frac %.4f**; '
            'whole timed region: %.1f TFLOP/s executed = %.3f of the dense peak'
            % (r['kernel'], r['achieved'], r['achieved_executed'], r['peak'], r['peak_dense'], r['frac'], d['whole_step']['mfma_tflops_executed'],
               d['whole_step']['frac_of_dense_fp32_mfma_peak']))


with open(os.path.join(P, '%s_final.md' % out), 'w') as f:
    w = f.write
    w('# Round 3, state "%s": the headline (VGG16, configs[1]) on 1x MI355X\n\n' % tag)
    w('Commands (`tools/run_round3_profile.sh`, one gpurun call, TAG=%s; composed by `tools/make_round3_profile.py`):\n\n' % tag)
    w('    python -m pytest tests -m gpu -q                  -> %s\n' % pytest_line)
    w('    python bench.py                                   -> profiles/%s_bench.json      (%.2f img/s, %.3f ms/step, K = 220: the full section-8d cycle)\n'
      % (out, b220['value'], b220['ms_per_step']))
    w('    python bench.py --steps 20 --warmup 5             -> profiles/%s_bench_k20.json  (%.2f img/s, %.3f ms/step: the driver\'s command)\n'
      % (out, b20['value'], b20['ms_per_step']))
    w('    rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --optin-steps 0\n')
    w('                                                      -> tables below; its bench line: profiles/%s_bench_under_rocprof_vgg16.json (%.1f img/s)\n\n'
      % (out, under['vgg16']['value']))
    w('Counters of this state\'s kernels over whole train steps: profiles/r03_pmc_step_{vgg16,resnet50,spherenet20}.md (MFMA-busy, other vector '
      'instructions per MFMA); per-unit phase times: docs/LAB_NOTEBOOK.md section 8 (tools/attic/diag_wg_timing.py); HBM bytes: the traffic table below.\n\n* %s\n' % roof(b20))
    w('* `cpu_baseline`: %s\n\n' % json.dumps({k: v for k, v in b20['cpu_baseline'].items() if k != 'sample'}))
    w('## bench.py, K = 20 (HIP events around every C-ABI launch of the timed region)\n\n%s\n\n' % fam_table(b20))
    w('phases: `%s`\n\n' % json.dumps(b20['phases']))
    w(traffic_table())
    w('## rocprofv3 --kernel-trace --stats of the same command (25 train + 4 eval passes incl. warm-up)\n\n')
    w(read(os.path.join(G, 'summary_%s_vgg16.md' % tag)))
    w('\n')

with open(os.path.join(P, '%s_other_nets.md' % out), 'w') as f:
    w = f.write
    w('# Round 3, state "%s": ResNet-50 and SphereNet-20 (BASELINE configs[3] / configs[4] topologies) on 1x MI355X, batch 256, fp32\n\n' % tag)
    w('    python tools/net_bench.py --arch resnet50 | spherenet20 --steps 10   (forward + backward + SGD, no cycle around it)\n')
    for ln in net:
        w('        ' + ln + '\n')
    w('    python bench.py --arch resnet50 | spherenet20 --steps 20 --warmup 5  (the task-1 CPG cycle: finetune -> prune -> recovery, validate, statistics)\n')
    for a, d in others.items():
        w('        %-12s %.1f img/s, %.3f ms/step -> profiles/%s_bench_%s.json\n' % (a, d['value'], d['ms_per_step'], out, a))
    w('    python tools/generic_bench.py --iters 5  (per shape class through the C ABI, TFLOP/s algorithmic; ms)\n\n')
    w('```\n' + read(os.path.join(G, 'generic_%s.txt' % tag)).strip() + '\n```\n\n')
    for a, d in others.items():
        w('## %s\n\n' % a)
        w('Winograd F(2x2,3x3) runs the 3x3 s1 layers (forward / input gradient also on the 7x7 maps; the weight gradient where the map is 14 or a '
          'multiple of 28 pixels wide); the launch-mix ceiling below is the bench\'s own: dense peak x algorithmic / executed flops of the launches it timed.\n\n')
        w('* %s\n* train steps alone: %.1f TFLOP/s algorithmic (`algorithmic_tflops_train_steps`)\n\n' % (roof(d), d['algorithmic_tflops_train_steps']))
        w(fam_table(d) + '\n\nphases: `%s`\n\n' % json.dumps(d['phases']))
        w('rocprofv3 --kernel-trace --stats of `bench.py --arch %s --steps 20 --warmup 5 --no-cpu-baseline` (bench line under the profiler: '
          '%.1f img/s):\n\n' % (a, under[a]['value']))
        w(read(os.path.join(G, 'summary_%s_%s.md' % (tag, a))))
        w('\n')
print('wrote profiles/%s_final.md, profiles/%s_other_nets.md' % (out, out))
